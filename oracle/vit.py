"""ORACLE -- test infrastructure only.

Plain torch (fp32 unless told otherwise) restatement of the reference's image -> visual-token path.
Functional style: every function takes a state-dict with the *reference's parameter names*.

Follows (paths relative to /root/reference/VisionLLMv2/):
  * InternViT: visionllmv2/model/internvit/modeling_intern_vit.py
      - InternRMSNorm            :33-44   (fp32 mean-square, rsqrt(var+eps), cast back, then * weight)
      - InternVisionEmbeddings   :82-90   (Conv2d k=s=patch WITH bias, flatten(2).T, cat CLS, + pos)
      - InternAttention._naive_attn :126-143 (qkv Linear, layout (three h d), QK-RMSNorm over the
                                   flattened H*D=C, (q*scale)@k^T, softmax, @v, proj)
      - InternMLP                :167-179 (fc1, act, fc2)
      - InternVisionEncoderLayer :198-210 (h + attn(norm1(h))*ls1 ; h + mlp(norm2(h))*ls2)
      - InternVisionEncoder      :253-270 (hidden_states tuple: L inputs + final output)
  * CLIP ViT-L/14 (third-party: transformers.CLIPVisionModel, pinned transformers==4.34.0 in
    requirements.txt:22; call sites visionllmv2/model/modeling_visionllmv2.py:135, 565-568):
      CLIPVisionEmbeddings (Conv2d no bias, CLS, pos) -> pre_layrnorm -> L x [LN1, q/k/v proj (+bias),
      q*scale, softmax, out_proj, residual, LN2, fc1, quick_gelu, fc2, residual]; hidden_states[0] is
      the *pre-layernormed* embedding output... NOTE: in HF CLIP hidden_states[0] is the output of
      pre_layrnorm (encoder input).  Pinned by tests/golden/clip_tiny.npz (generated with the HF class).
  * select + pixel-shuffle + bridge: visionllmv2/model/modeling_visionllmv2.py:569-579, 381-392, 162-182.

Parity pinning: tests/test_oracle_vit.py checks every function here against fixtures produced by the
reference classes imported in the build container (oracle/gen_golden.py).  The reference has no tests of
its own for this part of the path, so those fixtures are the pin.
"""
import math

import torch
import torch.nn.functional as F


def rms_norm(x, weight, eps):
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return weight * xf.to(dt)


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


_ACT = {"gelu": gelu_erf, "quick_gelu": quick_gelu}


def patch_embed(pixel_values, weight, bias, class_embedding, position_embedding, patch):
    x = F.conv2d(pixel_values, weight, bias, stride=patch)  # [N, C, g, g]
    x = x.flatten(2).transpose(1, 2)  # [N, g*g, C]
    cls = class_embedding.reshape(1, 1, -1).expand(x.shape[0], 1, -1).to(x.dtype)
    x = torch.cat([cls, x], dim=1)
    return x + position_embedding.reshape(1, -1, x.shape[-1]).to(x.dtype)


def attention_core(q, k, v, scale):
    """q,k,v [B,H,S,D] -> [B,S,H*D]; (q*scale)@k^T, softmax, @v  (modeling_intern_vit.py:136-140)."""
    attn = (q * scale) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    B, H, S, D = q.shape
    return (attn @ v).transpose(1, 2).reshape(B, S, H * D)


def intern_attention(x, sd, prefix, num_heads, qk_norm, eps):
    B, N, C = x.shape
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, num_heads, C // num_heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    if qk_norm:
        B_, H_, N_, D_ = q.shape
        q = rms_norm(q.transpose(1, 2).flatten(-2, -1), sd[prefix + "q_norm.weight"], eps)
        q = q.view(B_, N_, H_, D_).transpose(1, 2)
        k = rms_norm(k.transpose(1, 2).flatten(-2, -1), sd[prefix + "k_norm.weight"], eps)
        k = k.view(B_, N_, H_, D_).transpose(1, 2)
    scale = (C // num_heads) ** -0.5
    o = attention_core(q, k, v, scale)
    return F.linear(o, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


def intern_vit_forward(sd, cfg, pixel_values):
    """cfg: dict(hidden_size, num_attention_heads, num_hidden_layers, patch_size, layer_norm_eps,
    qk_normalization, hidden_act).  Returns list of L+1 hidden states [N,S,C]."""
    eps = cfg.get("layer_norm_eps", 1e-6)
    act = _ACT[cfg.get("hidden_act", "gelu")]
    h = patch_embed(pixel_values, sd["embeddings.patch_embedding.weight"], sd.get("embeddings.patch_embedding.bias"),
                    sd["embeddings.class_embedding"], sd["embeddings.position_embedding"], cfg["patch_size"])
    hs = []
    for i in range(cfg["num_hidden_layers"]):
        hs.append(h)
        p = f"encoder.layers.{i}."
        a = intern_attention(rms_norm(h, sd[p + "norm1.weight"], eps), sd, p + "attn.",
                             cfg["num_attention_heads"], cfg.get("qk_normalization", True), eps)
        h = h + a * sd[p + "ls1"]
        m = rms_norm(h, sd[p + "norm2.weight"], eps)
        m = F.linear(m, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        m = F.linear(act(m), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        h = h + m * sd[p + "ls2"]
    hs.append(h)
    return hs


def clip_vit_forward(sd, cfg, pixel_values, prefix=""):
    """HF CLIPVisionModel restatement.  ``prefix`` is '' (transformers>=5 layout) or 'vision_model.'."""
    eps = cfg.get("layer_norm_eps", 1e-5)
    act = _ACT[cfg.get("hidden_act", "quick_gelu")]
    C = cfg["hidden_size"]
    H = cfg["num_attention_heads"]
    g = lambda k: sd[prefix + k]  # noqa: E731
    h = patch_embed(pixel_values, g("embeddings.patch_embedding.weight"), None,
                    g("embeddings.class_embedding"), g("embeddings.position_embedding.weight"), cfg["patch_size"])
    h = F.layer_norm(h, (C,), g("pre_layrnorm.weight"), g("pre_layrnorm.bias"), eps)
    hs = []
    for i in range(cfg["num_hidden_layers"]):
        hs.append(h)
        p = f"encoder.layers.{i}."
        x = F.layer_norm(h, (C,), g(p + "layer_norm1.weight"), g(p + "layer_norm1.bias"), eps)
        B, S, _ = x.shape
        q = F.linear(x, g(p + "self_attn.q_proj.weight"), g(p + "self_attn.q_proj.bias"))
        k = F.linear(x, g(p + "self_attn.k_proj.weight"), g(p + "self_attn.k_proj.bias"))
        v = F.linear(x, g(p + "self_attn.v_proj.weight"), g(p + "self_attn.v_proj.bias"))
        sp = lambda t: t.view(B, S, H, C // H).transpose(1, 2)  # noqa: E731
        o = attention_core(sp(q), sp(k), sp(v), (C // H) ** -0.5)
        h = h + F.linear(o, g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
        x = F.layer_norm(h, (C,), g(p + "layer_norm2.weight"), g(p + "layer_norm2.bias"), eps)
        x = F.linear(x, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"))
        h = h + F.linear(act(x), g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
    hs.append(h)
    return hs


def pixel_shuffle(x, scale_factor=0.5):
    """modeling_visionllmv2.py:381-392 (InternVL (w,h) permute order)."""
    n, w, h, c = x.size()
    x = x.view(n, w, int(h * scale_factor), int(c / scale_factor))
    x = x.permute(0, 2, 1, 3).contiguous()
    x = x.view(n, int(h * scale_factor), int(w * scale_factor), int(c / (scale_factor * scale_factor)))
    return x.permute(0, 2, 1, 3).contiguous()


def select_features(hidden_states, layer=-2, use_pixelshuffle=False):
    """modeling_visionllmv2.py:569-578: hs[layer][:,1:] (+ pixel-shuffle)."""
    f = hidden_states[layer][:, 1:]
    if use_pixelshuffle:
        hw = int(f.shape[1] ** 0.5)
        f = pixel_shuffle(f.reshape(f.shape[0], hw, hw, -1), 0.5)
        f = f.reshape(f.shape[0], -1, f.shape[-1])
    return f


def bridge_forward(sd, kind, x, prefix=""):
    """vl_bridge (modeling_visionllmv2.py:162-182).  kinds: 'linear', 'internvl_mlp', 'mlp{N}x_gelu'."""
    g = lambda k: sd[prefix + k]  # noqa: E731
    if kind == "linear":
        return F.linear(x, g("weight"), g("bias"))
    if kind in ("internvl_mlp", "internvl"):
        x = F.layer_norm(x, (x.shape[-1],), g("0.weight"), g("0.bias"), 1e-5)
        x = F.linear(x, g("1.weight"), g("1.bias"))
        return F.linear(gelu_erf(x), g("3.weight"), g("3.bias"))
    import re
    m = re.match(r"^mlp(\d+)x_gelu*", kind)
    if not m:
        raise NotImplementedError(kind)
    depth = int(m.group(1))
    x = F.linear(x, g("0.weight"), g("0.bias"))
    for i in range(1, depth):
        x = F.linear(gelu_erf(x), g(f"{2 * i}.weight"), g(f"{2 * i}.bias"))
    return x


# ---- any-res tiling grid (host logic feeding the path; mm_utils.py:23-77) -------------------------
def find_closest_aspect_ratio(aspect_ratio, target_ratios, width, height, image_size):
    best_diff, best = float("inf"), (1, 1)
    area = width * height
    for r in target_ratios:
        d = abs(aspect_ratio - r[0] / r[1])
        if d < best_diff:
            best_diff, best = d, r
        elif d == best_diff and area > 0.5 * image_size * image_size * r[0] * r[1]:
            best = r
    return best


def tile_grid(width, height, min_num=1, max_num=6, image_size=448, use_thumbnail=True):
    """(cols, rows, n_tiles incl. thumbnail) chosen by dynamic_preprocess (mm_utils.py:39-77)."""
    ratios = sorted({(i, j) for n in range(min_num, max_num + 1) for i in range(1, n + 1)
                     for j in range(1, n + 1) if min_num <= i * j <= max_num}, key=lambda x: x[0] * x[1])
    c, r = find_closest_aspect_ratio(width / height, ratios, width, height, image_size)
    n = c * r
    if use_thumbnail and n != 1:
        n += 1
    return c, r, n
