#!/usr/bin/env python3
"""Hot-path benchmark: images/sec of the image -> visual-token path + MSDA forward on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N>1: the driver launches `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
     bench.py --gpus N ...`; started WITHOUT a launcher, `python bench.py --gpus N` re-executes itself under that launcher.)

One "step" = one pass of the hot path over one batch of synthetic input PER RANK (weak scaling):
  workload "vitl14_336_5tiles+mlp2x_gelu+msda_cfg4" (BASELINE.json metric; configs[1] arch + configs[3] shapes)
    * 8 images of 1336x1336 -> dynamic_preprocess gives 5 tiles of 336x336 each (mm_utils.py:39-77) = 40 tiles
    * CLIP ViT-L/14-336 (24 layers, bf16, all 25 hidden states materialised, as the reference requests them)
    * hidden_states[-2][:, 1:] -> mlp2x_gelu vl_bridge (1024 -> 4096 -> 4096)   => visual tokens [40, 576, 4096]
    * N>1: RCCL all-gather of the visual tokens over xGMI
    * MSDA forward, Grounding-DINO det head at the 1344x1344 padded image: 4 levels 168^2..21^2, M=8, D=32, P=4,
      B=8: 6 encoder-shaped calls (Lq = S = 37485) + 6 decoder-shaped calls (Lq = 900), fp32 (the reference upcasts).
Weights are random-init (no network), inputs synthetic and resident in HBM before the timed region.

Everything of a step is enqueued on ONE stream (--msda-stream 1 puts the det-head MSDA calls on a side stream: valid only
as cross-batch pipelining).
Prints ONE JSON line (rank 0): metric/value/unit + "roofline" (the kernel with the largest measured share of a step) +
"rooflines" (every hot kernel, each timed alone with HIP events on the stream the C ABI launches on) + "phases_ms"
(ViT+projector / token all-gather / MSDA, max over ranks) + "cpu_baseline" (reference CPU path on the host cores: 1 warm-up
+ 3 repetitions, median, bounded sample; N=1 only).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak

IMAGES_PER_RANK = 8
TILES_PER_IMAGE = 5
VIT = dict(hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=24, image_size=336,
           patch_size=14, hidden_act="quick_gelu", layer_norm_eps=1e-5)
# BASELINE.json configs[2]: InternViT-6B + pixel-shuffle + internvl_mlp projector on 448x448 tiles (--workload internvit6b)
IVIT = dict(hidden_size=3200, num_attention_heads=25, intermediate_size=12800, num_hidden_layers=48, image_size=448,
            patch_size=14, qk_normalization=True, qkv_bias=False, hidden_act="gelu", layer_norm_eps=1e-6)
LLM_HIDDEN = 4096
MSDA = dict(M=8, D=32, P=4, shapes=[(168, 168), (84, 84), (42, 42), (21, 21)], dec_queries=900, enc_layers=6,
            dec_layers=6)


def build_intern_model(dev):
    from visionllm_amd.bridge import build_vl_bridge
    from visionllm_amd.intern_vit import InternVisionConfig, InternVisionModel
    torch.manual_seed(0)
    with torch.device(dev):
        enc = InternVisionModel(InternVisionConfig(**IVIT))
        with torch.no_grad():
            for name, p in enc.named_parameters():
                if p.dim() >= 2:
                    p.normal_(0, 0.02)
        bridge = build_vl_bridge("internvl_mlp", IVIT["hidden_size"], LLM_HIDDEN, use_pixelshuffle=True)
    return enc.to(torch.bfloat16).eval().requires_grad_(False), bridge.to(torch.bfloat16).eval().requires_grad_(False)


def build_model(dev):
    from transformers import CLIPVisionConfig
    from visionllm_amd.bridge import build_vl_bridge
    from visionllm_amd.clip_vit import CLIPVisionModel
    torch.manual_seed(0)
    enc = CLIPVisionModel(CLIPVisionConfig(**VIT))
    with torch.no_grad():
        for name, p in enc.named_parameters():
            if p.dim() >= 2:
                p.normal_(0, 0.02)
    torch.manual_seed(1)
    bridge = build_vl_bridge("mlp2x_gelu", VIT["hidden_size"], LLM_HIDDEN, use_pixelshuffle=False)
    return enc.to(dev).to(torch.bfloat16).eval().requires_grad_(False), bridge.to(dev).to(torch.bfloat16).eval().requires_grad_(False)


def build_msda_inputs(dev, B, seed):
    from msda_inputs import make_inputs
    g_enc = make_inputs(1, MSDA["M"], MSDA["D"], MSDA["shapes"], MSDA["P"], mode="encoder_like", seed=seed)
    g_dec = make_inputs(1, MSDA["M"], MSDA["D"], MSDA["shapes"], MSDA["P"], Lq=MSDA["dec_queries"], mode="encoder_like",
                        seed=seed + 1)
    out = {}
    for tag, g in (("enc", g_enc), ("dec", g_dec)):
        t = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
        gen = torch.Generator(device=dev).manual_seed(seed)
        for k in ("value", "loc", "attw"):
            t[k] = t[k].repeat(B, *([1] * (t[k].dim() - 1))).contiguous()
        t["value"] = t["value"] + 0.05 * torch.randn(t["value"].shape, device=dev, generator=gen)
        t["loc"] = (t["loc"] + 0.002 * torch.randn(t["loc"].shape, device=dev, generator=gen)).contiguous()
        # every det-head layer has its own value_proj, i.e. its own value tensor: the decoder-shaped calls rotate over 3
        # buffers (3 x 307 MB > the 256 MB Infinity Cache) instead of re-reading one tensor out of the caches; the
        # encoder-shaped call moves 1.07 GB per call anyway
        n_buf = 3 if tag == "dec" else 1
        t["values"] = [t["value"]] + [t["value"] + 0.01 * torch.randn(t["value"].shape, device=dev, generator=gen) for _ in range(n_buf - 1)]
        out[tag] = t
    return out


def build_msda_layer(dev, B):
    """The deformable-attention LAYER the det head actually calls (value_proj + offsets / weights linears + softmax + location
    arithmetic + operator + output_proj; ...mask_dn.py:706-784) at the encoder shape of cfg 4, bf16 module: a reported entry
    (`rooflines.msda_layer`), not part of the headline step.  Returns (callable, algorithmic bytes)."""
    from visionllm_amd import ms_deform_attn as A
    C, M, L, P = 256, MSDA["M"], len(MSDA["shapes"]), MSDA["P"]
    S = sum(h * w for h, w in MSDA["shapes"])
    torch.manual_seed(7)
    mod = A.MSDeformAttn(C, L, M, P).to(dev).to(torch.bfloat16).eval().requires_grad_(False)
    with torch.no_grad():
        mod.sampling_offsets.weight.normal_(0, 0.01)
    ss = torch.tensor(MSDA["shapes"], device=dev)
    lsi = torch.cat([ss.new_zeros(1), (ss[:, 0] * ss[:, 1]).cumsum(0)[:-1]])
    gen = torch.Generator(device=dev).manual_seed(17)
    src = torch.randn(B, S, C, device=dev, generator=gen).to(torch.bfloat16)
    q = torch.randn(B, S, C, device=dev, generator=gen).to(torch.bfloat16)
    pts = []
    for h, w in MSDA["shapes"]:   # a query's reference point is its own pixel centre (...mask_dn.py:1579-1606)
        ys, xs = torch.meshgrid(torch.arange(h, device=dev) + 0.5, torch.arange(w, device=dev) + 0.5, indexing="ij")
        pts.append(torch.stack([xs.reshape(-1) / w, ys.reshape(-1) / h], -1))
    ref = torch.cat(pts)[None, :, None, :].expand(B, S, L, 2).contiguous()

    def call():
        with torch.no_grad():
            return mod(q, ref, src, ss, lsi, None)
    # algorithmic bytes: the layer's bf16 inputs and output (input_flatten, query, out) + the reference points; its four linears
    # are 137.6 GFLOP (= 55 us at the dense bf16 peak): HBM is the tighter bound
    ab = 3 * B * S * C * 2 + ref.numel() * 4
    return call, ab


def msda_bytes(t):
    B, S, M, D = t["value"].shape
    Lq, L, P = t["loc"].shape[1], t["loc"].shape[3], t["loc"].shape[4]
    return 4 * (B * S * M * D + B * Lq * M * L * P * 2 + B * Lq * M * L * P + B * Lq * M * D)


def event_time(fn, iters, stream=None, rounds=3):
    """Average duration (s) of fn() launched `iters` times back to back on the CURRENT torch stream, HIP events; the median of
    `rounds` such loops (the first loop after an idle gap reads 5-10 % slow while the clocks ramp)."""
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3 / iters)
    return float(np.median(ts))


class ClockSampler:
    """Shader clock + package power of the GPU this rank runs on, sampled WHILE the timed steps run (the same
    binary reads +-10 % on different boxes of the pool; without the clock next to `ms_per_step` a box effect cannot be told from a
    code effect).  Source: the amdgpu hwmon files (freq1_input = sclk in Hz, power1_average / power1_input in uW) read by a
    background thread every 20 ms -- a file read, nothing is spawned inside the timed region.  Where they are missing,
    `rocm-smi --showclocks --showpower` is sampled over a SECOND, untimed run of the same steps (a subprocess per sample)."""

    def __init__(self, dev):
        import glob
        self.rows, self._stop, self._th, self.src = [], False, None, None
        want = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
        except Exception:
            pass
        cands = []
        for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            f = os.path.join(h, "freq1_input")
            pw = next((os.path.join(h, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, n))), None)
            if os.path.exists(f):
                cands.append((os.path.realpath(os.path.join(h, "..", "..")), f, pw))
        hit = [c for c in cands if want and want in c[0]]
        self.files = (hit or cands)[:8 if not hit else 1]

    def _read(self):
        best = None
        for _, f, pw in self.files:
            try:
                mhz = int(open(f).read()) / 1e6
                w = int(open(pw).read()) / 1e6 if pw else float("nan")
            except Exception:
                continue
            if best is None or (w if w == w else 0.0) > (best[1] if best[1] == best[1] else 0.0):   # several candidates: the busy one
                best = (mhz, w)
        return best

    def _loop_sysfs(self):
        while not self._stop:
            r = self._read()
            if r:
                self.rows.append(r)
            time.sleep(0.02)

    def _loop_smi(self):
        import re
        import subprocess
        while not self._stop:
            try:
                o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
                sclk = re.search(r"sclk clock level.*?\((\d+)Mhz\)", o)
                pw = re.search(r"Power \(W\):\s*([\d.]+)", o)
                if sclk:
                    self.rows.append((float(sclk.group(1)), float(pw.group(1)) if pw else float("nan")))
            except Exception:
                return
            time.sleep(0.05)

    def has_sysfs(self):
        return bool(self.files) and self._read() is not None

    def start(self, sysfs=True):
        import threading
        self.rows, self._stop = [], False
        self.src = "amdgpu hwmon (freq1_input / power1_average), 20 ms period" if sysfs else "rocm-smi --showclocks --showpower"
        self._th = threading.Thread(target=self._loop_sysfs if sysfs else self._loop_smi, daemon=True)
        self._th.start()

    def stop(self):
        self._stop = True
        if self._th is not None:
            self._th.join(timeout=10)
        rows = self.rows[1:] if len(self.rows) > 2 else self.rows     # (the first sample predates the load)
        if not rows:
            return None
        mhz = [r[0] for r in rows]
        pw = [r[1] for r in rows if r[1] == r[1]]
        return {"sclk_mhz_median": float(np.median(mhz)), "sclk_mhz_min": float(min(mhz)), "sclk_mhz_max": float(max(mhz)),
                "power_w_median": float(np.median(pw)) if pw else None, "power_w_max": float(max(pw)) if pw else None,
                "samples": len(rows), "source": self.src}


NONPYR_SHAPES = [(100, 167), (50, 84), (25, 42), (13, 21)]   # ceil-divided levels of an 800 x 1333 detection input


def kernel_rooflines(dev, msda_in, n_tiles, cfg, bridge_dims, iters=10, workload="vitl", in_step=None, layer=None):
    """Achieved vs peak of every hot kernel of the workload.  `us_per_launch` (and achieved / frac) are the IN-STEP figures
    when `in_step` has them ({prof tag: (us per launch, launches per step)}, from vllm_prof_read over instrumented steps:
    event-to-event time inside step(), queueing and launch gaps included); `us_per_launch_isolated` is the same launch
    alone, back to back (HIP events on torch's current stream = the stream the C ABI launches on), which reads 3-6 % fast.
    cfg: VIT or IVIT; bridge_dims: [(M, N, K), ...]."""
    from visionllm_amd import _lib
    from visionllm_amd import ms_deform_attn as A
    L = _lib.lib()
    st = _lib.current_stream(torch.device(dev))
    in_step = in_step or {}
    C, H, I, NL = cfg["hidden_size"], cfg["num_attention_heads"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    S = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
    D = C // H
    M = n_tiles * S
    act = 2 if cfg["hidden_act"] == "quick_gelu" else 1
    out = {}

    def entry(tag, kernel, bound, work, sec_iso, n_launch, peak, unit, scale, **extra):
        us_iso = sec_iso * 1e6
        us = in_step[tag][0] if tag in in_step else us_iso
        e = dict(kernel=kernel, bound=bound, achieved=work / (us * 1e-6) / scale, peak=peak, unit=unit,
                 frac=(work / (us * 1e-6) / scale / peak) if peak else None, traffic=None, us_per_launch=us,
                 us_per_launch_isolated=us_iso, timing="in-step (event to event inside step())" if tag in in_step else "isolated launches",
                 launches_per_step=in_step[tag][1] if tag in in_step else n_launch)
        e.update(extra)
        return e

    # MSDA: HBM bound, algorithmic bytes = value + loc + attw + out
    for tag, nm, n_launch, ptag in (("enc", "msda", MSDA["enc_layers"], "msda_encoder_shape"), ("dec", "msda_dec", MSDA["dec_layers"], "msda_other")):
        t = msda_in[tag]
        vals = t["values"]
        k = [0]

        def f(t=t, vals=vals, k=k):
            k[0] += 1
            return A.ms_deform_attn_forward(vals[k[0] % len(vals)], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
        f(); torch.cuda.synchronize()
        sec = event_time(f, iters)
        ab = msda_bytes(t)
        geo = A.known_geometry(t["shapes"], t["loc"].shape[1])
        if tag == "enc":
            kern = {A.GEO_PYRAMID: "msda_fwd_tiled9_kernel<12 waves in two teams, 1 block per CU> (pyramid items; ONE launch: geometry known on the host)",
                    A.GEO_NESTED: "msda_fwd_tiled9_kernel<nested maps> (pyramid items; one launch)",
                    A.GEO_GENERAL: "msda_fwd_tiled4_kernel (any-geometry LDS-tiled kernel; one launch)",
                    A.GEO_UNKNOWN: "msda_fwd_tiled9_kernel + empty msda_fwd_tiled4_kernel launch (geometry decided on the device)"}[geo]
            kern += " fp32, D32, encoder shape Lq=S=37485, B=8"
            out[nm] = entry(ptag, kern, "hbm", ab, sec, n_launch, HBM_PEAK_GBS, "GB/s", 1e9, algorithmic_bytes=ab)
        else:
            # 325 MB of value + 37 MB of the rest per call; the step rotates over len(vals) value tensors (one per decoder layer,
            # as in the det head), so the calls stream from HBM instead of re-reading one tensor from L2 / Infinity Cache
            out[nm] = entry(ptag, "msda_fwd_vec_kernel (fp32, D32, decoder shape Lq=900, B=8)", "hbm", ab, sec, n_launch,
                            HBM_PEAK_GBS, "GB/s", 1e9, algorithmic_bytes=ab,
                            note=f"value tensors rotated over {len(vals)} buffers ({len(vals) * vals[0].numel() * 4 / 1e6:.0f} MB > the 256 MB Infinity Cache)")
    if layer is not None:
        try:
            call, ab = layer
            call(); torch.cuda.synchronize()
            sec = event_time(call, 5)
            out["msda_layer"] = entry("msda_layer", "vllm_msda_layer_forward (bf16 MSDeformAttn module, encoder shape Lq=S=37485, B=8: gemm_skinny_kernel value GEMM "
                                      "fp32 -> gemm_skinny_kernel query GEMM with softmax + location epilogue -> msda_fwd_tiled9_kernel writing bf16 -> "
                                      "gemm_skinny_kernel output GEMM)", "hbm", ab, sec, 1, HBM_PEAK_GBS,
                                      "GB/s", 1e9, algorithmic_bytes=ab,
                                      note="the layer every det-head call site runs around the operator; ONE call per instrumented step, behind the 12 "
                                           "operator calls; NOT part of the timed step / headline")
        except Exception as e:   # measurement extra: never fail the bench line for it
            out["msda_layer"] = {"error": repr(e)}
    # the same operator on a NON-pyramid geometry (ceil-divided levels, what detection inputs usually give): isolated only
    try:
        from msda_inputs import make_inputs
        g = make_inputs(1, MSDA["M"], MSDA["D"], NONPYR_SHAPES, MSDA["P"], mode="encoder_like", seed=3)
        tn = {k_: torch.from_numpy(v).to(dev) for k_, v in g.items()}
        for k_ in ("value", "loc", "attw"):
            tn[k_] = tn[k_].repeat(IMAGES_PER_RANK, *([1] * (tn[k_].dim() - 1))).contiguous()
        A.remember_geometry(tn["shapes"])
        f = lambda: A.ms_deform_attn_forward(tn["value"], tn["shapes"], tn["lsi"], tn["loc"], tn["attw"], 64)  # noqa: E731
        f(); torch.cuda.synchronize()
        sec = event_time(f, iters)
        tn["values"] = [tn["value"]]
        ab = msda_bytes(tn)
        geo_n = A.known_geometry(tn["shapes"], tn["loc"].shape[1])
        kern_n = ("msda_fwd_tiled9_kernel (pyramid items on NESTED maps: ceil-divided levels" if geo_n in (A.GEO_PYRAMID, A.GEO_NESTED)
                  else "msda_fwd_tiled4_kernel (any-geometry kernel: levels")
        out["msda_nonpyramid"] = entry("-", kern_n + " 100x167 / 50x84 / 25x42 / 13x21, B=8, Lq=S)", "hbm",
                                       ab, sec, 0, HBM_PEAK_GBS, "GB/s", 1e9, algorithmic_bytes=ab,
                                       note="not part of the step: the level maps of an 800 x 1333 detection input (not exact halves)")
        del tn
    except Exception as e:   # measurement extra: never fail the bench line for it
        out["msda_nonpyramid"] = {"error": repr(e)}
    # MSDA backward at the encoder shape (the det / pose heads are trained): not part of the forward step, isolated.
    # Algorithmic bytes: everything the forward reads + grad_out, and the three gradients written once (grad_value is accumulated
    # with atomics; its zero fill is the caller's and not counted).
    try:
        t = msda_in["enc"]
        go = torch.randn(t["loc"].shape[0], t["loc"].shape[1], t["value"].shape[2] * t["value"].shape[3], device=dev)
        f = lambda: A.ms_deform_attn_backward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], go, 64)  # noqa: E731
        f(); torch.cuda.synchronize()
        sec = event_time(f, 3)
        ab = 2 * msda_bytes(t)
        out["msda_bwd"] = entry("-", "msda_bwd_mfma_kernel + zero fill of grad_value (fp32, D32, encoder shape Lq=S=37485, B=8; grad_value = S^T x grad_out on "
                                "v_mfma_f32_32x32x2_f32)", "hbm", ab, sec, 0, HBM_PEAK_GBS, "GB/s", 1e9, algorithmic_bytes=ab,
                                note="not part of the step (training only); time includes the allocation of the three gradients and the zero fill of grad_value (the kernel writes every element of the two per-point gradients: no memset for them since round 5)")
        del go
    except Exception as e:   # measurement extra: never fail the bench line for it
        out["msda_bwd"] = {"error": repr(e)}
    # SURVEY 8(f) rows 3 / 4 (round-4 review, item 6): DCNv3 forward at two InternImage stages, region point sampling, visual-token
    # splice -- none is part of the timed step (the north-star path has no InternImage backbone / region prompts in it); isolated
    # launches on this box so that every row of 8(f) has a driver-run fraction.  Algorithmic bytes = every operand once.
    try:
        out.update(extra_rooflines(dev, entry))
    except Exception as e:   # measurement extra: never fail the bench line for it
        out["dcnv3"] = {"error": repr(e)}
    # attention: MFMA bound, flops = 4*H*S^2*d per tile
    qkv = torch.randn(n_tiles, S, 3, H, D, device=dev).to(torch.bfloat16)
    ao = torch.empty(n_tiles, S, H, D, dtype=torch.bfloat16, device=dev)
    f = lambda: _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(ao), n_tiles, S, H, D, D ** -0.5, st))  # noqa: E731
    f(); torch.cuda.synchronize()
    sec = event_time(f, iters)
    fl = 4.0 * H * S * S * D * n_tiles
    out["attn"] = entry("attn", f"attn_fwd_kernel<D{D}> ({n_tiles} tiles x {H} heads x S{S})", "mfma", fl, sec, NL, MFMA_BF16_PEAK_TF,
                        "TFLOP/s", 1e12, algorithmic_flops=fl)
    del qkv, ao
    # the four encoder GEMMs + the projector GEMMs; epilogues as the encoder uses them: 3 = bias (+ LayerScale for InternViT) + residual
    gemms = [("gemm_qkv", M, 3 * C, C, 0, NL, "gemm_qkv"), ("gemm_proj", M, C, C, 3, NL, "gemm_proj"), ("gemm", M, I, C, act, NL, "gemm_fc1"),
             ("gemm_fc2", M, C, I, 3, NL, "gemm_fc2")]
    gemms += [(f"gemm_bridge{i}", m, n, k, 0, 1, "-") for i, (m, n, k) in enumerate(bridge_dims)]
    has_ls = "qk_normalization" in cfg   # InternViT: LayerScale in the residual epilogue; CLIP: plain residual
    names = {"gemm": "MLP fc1 + activation", "gemm_qkv": "QKV", "gemm_proj": "attention out-proj + " + ("LayerScale + " if has_ls else "") + "residual",
             "gemm_fc2": "MLP fc2 + " + ("LayerScale + " if has_ls else "") + "residual"}
    for nm, m, n, k, epi, n_launch, ptag in gemms:
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(n, device=dev).to(torch.bfloat16)
        y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        res = torch.randn(m, n, device=dev).to(torch.bfloat16) if epi == 3 else None
        ls = torch.full((n,), 0.1, device=dev).to(torch.bfloat16) if (epi == 3 and has_ls) else None
        f = lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), m, n, k, k, k, n, epi,  # noqa: E731
                                                 _lib.ptr(ls), _lib.ptr(res), n if epi == 3 else 0, 0, st))
        pl0 = L.vllm_gemm_persistent_launches()
        f(); torch.cuda.synchronize()
        kname = ("gemm256p_kernel [persistent 8-phase schedule]" if L.vllm_gemm_persistent_launches() > pl0 else
                 "gemm256_bf16_kernel [8-phase schedule, one workgroup per tile]")
        sec = event_time(f, iters)
        fl = 2.0 * m * n * k
        out[nm] = entry(ptag, f"{kname} ({names.get(nm, 'projector linear')}: M{m} N{n} K{k})", "mfma", fl, sec, n_launch,
                        MFMA_BF16_PEAK_TF, "TFLOP/s", 1e12, algorithmic_flops=fl)
        del x, w, b, y, res, ls
    # norms (LayerNorm / RMSNorm, incl. InternViT's q / k norms): HBM bound, one read + one write of [M, C] bf16
    if "norm" in in_step:
        ab = 2.0 * M * C * 2
        us, n = in_step["norm"]
        out["norm"] = dict(kernel=f"norm_bf16_kernel ([{M}, {C}] bf16)", bound="hbm", achieved=ab / (us * 1e-6) / 1e9, peak=HBM_PEAK_GBS,
                           unit="GB/s", frac=ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, traffic=None, us_per_launch=us,
                           timing="in-step (event to event inside step())", launches_per_step=n, algorithmic_bytes=ab)
    if "qk_norm" in in_step:
        ab = 2.0 * 2 * M * C * 2
        us, n = in_step["qk_norm"]
        out["qk_norm"] = dict(kernel=f"norm_bf16_kernel (q and k RMSNorm, one launch over the [{M}, 2 x {C}] slab of qkv)", bound="hbm",
                              achieved=ab / (us * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s", frac=ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                              traffic=None, us_per_launch=us, timing="in-step (event to event inside step())", launches_per_step=n,
                              algorithmic_bytes=ab)
    # optional: HBM traffic per launch from a separate rocprofv3 --pmc pass (profiles/pmc_traffic.json)
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            tr = json.load(open(pmc)).get(workload, {})
            for k in out:
                if k in tr:
                    out[k]["traffic"] = tr[k]
        except Exception:
            pass
    return out


def extra_rooflines(dev, entry):
    """rooflines.dcnv3 / dcnv3_84 / point_sample / point_sample_mean / splice (isolated; reference: ops_dcnv3/test.py:19-96 shapes scaled to
    InternImage stages, region_encoder.py:95-145, modeling_visionllmv2.py:582-605)."""
    from visionllm_amd import dcnv3 as DC
    from visionllm_amd import region_encoder as RE
    from visionllm_amd import splice as SP
    out = {}
    torch.manual_seed(0)
    k = 3
    for nm, (N, H, W, G, Cg) in (("dcnv3", (8, 168, 168, 20, 32)), ("dcnv3_84", (8, 84, 84, 40, 32))):
        x = torch.randn(N, H, W, G * Cg, device=dev)
        off = torch.randn(N, H, W, G * k * k * 2, device=dev)
        m = torch.softmax(torch.randn(N, H, W, G, k * k, device=dev), -1).reshape(N, H, W, -1)
        f = lambda: DC.dcnv3_forward(x, off, m, k, k, 1, 1, 1, 1, 1, 1, G, Cg, 1.0)  # noqa: E731
        f(); torch.cuda.synchronize()
        sec = event_time(f, 10)
        ab = (x.numel() * 2 + off.numel() + m.numel()) * 4.0
        out[nm] = entry("-", f"dcnv3_fwd_pipe_kernel (fp32, N{N} {H}x{W}, {G} groups x {Cg} channels, 3x3 points: an InternImage stage)", "hbm", ab, sec, 0,
                        HBM_PEAK_GBS, "GB/s", 1e9, algorithmic_bytes=ab, note="not part of the step (SURVEY 8f row 3); input + output + offsets + mask, fp32")
        del x, off, m
    # region point sampling: 16 regions x 2304 points on the three ViT feature maps concatenated (region_encoder.py:135: C = 3 x 1024 at 24 x 24)
    N, C, H, W, P = 16, 3072, 24, 24, 2304
    x = torch.randn(N, C, H, W, device=dev)
    pts = torch.rand(N, P, 2, device=dev)
    valid = torch.rand(N, P, device=dev) < 0.8
    with torch.no_grad():
        f = lambda: RE.point_sample(x, pts)  # noqa: E731
        f(); torch.cuda.synchronize()
        sec = event_time(f, 10)
        ab = (x.numel() + pts.numel() + N * C * P) * 4.0
        out["point_sample"] = entry("-", f"point_sample_kernel (fp32, {N} regions x {P} points, C{C} on {H}x{W})", "hbm", ab, sec, 0, HBM_PEAK_GBS, "GB/s", 1e9,
                                    algorithmic_bytes=ab, note="not part of the step (SURVEY 8f row 4); feature map + points read, (N, C, P) written")
        f = lambda: RE.point_sample_masked_mean(x, pts, valid)  # noqa: E731
        f(); torch.cuda.synchronize()
        sec = event_time(f, 10)
        ab = (x.numel() + pts.numel() + N * C) * 4.0 + valid.numel()
        out["point_sample_mean"] = entry("-", f"point_sample_mean_pix_kernel (the fused masked mean of region_encoder.py:135-140 as a pixel-weight product, same shapes)", "hbm", ab, sec, 0,
                                         HBM_PEAK_GBS, "GB/s", 1e9, algorithmic_bytes=ab,
                                         note="113 MB of feature maps read once, (N, C) out; round 5's point walk: 103 us = 0.13")
    del x, pts, valid
    # visual-token splice: 8 samples x 4096 positions x 4096 channels bf16, 40 tiles x 576 tokens scattered into the <im_patch> slots
    B, Lt, Cc, T = 8, 4096, 4096, 576
    emb = torch.randn(B, Lt, Cc, device=dev).to(torch.bfloat16)
    ids = torch.zeros(B, Lt, dtype=torch.int64, device=dev)
    ids[:, 100:100 + 5 * T] = 7
    feats = torch.randn(B * 5, T, Cc, device=dev).to(torch.bfloat16)
    f = lambda: SP.splice_visual_tokens(emb, ids, 7, feats, split_sizes=[5] * B)  # noqa: E731
    f(); torch.cuda.synchronize()
    sec = event_time(f, 10)
    ab = 2.0 * feats.numel() * 2
    out["splice"] = entry("-", f"splice_visual_tokens = vllm_splice_visual_tokens_bf16: slot scan + row mover, checked ({B * 5 * T} rows x {Cc} bf16 into [{B}, {Lt}, {Cc}])", "hbm", ab, sec, 0,
                          HBM_PEAK_GBS, "GB/s", 1e9, algorithmic_bytes=ab,
                          note="not part of the step (SURVEY 8 row a12 / 8f row 4); one native call incl. the slot scan (one block) and the status read-back "
                               "(the only host synchronisation; rounds 1-4: torch mask / nonzero / has_image gather + a scatter kernel: 225-236 us); "
                               "bytes = rows read + rows written")
    # the native row mover alone (the same rows, indices already on the device): what the kernel does without torch's mask / nonzero
    from visionllm_amd import _lib
    idx = torch.nonzero((ids == 7).reshape(-1), as_tuple=False).reshape(-1).contiguous()
    rows = feats.reshape(-1, Cc).contiguous()
    L_ = _lib.lib()
    f = lambda: _lib.check(L_.vllm_scatter_rows_bf16(_lib.ptr(rows), _lib.ptr(idx), _lib.ptr(emb), idx.numel(), Cc, B * Lt,  # noqa: E731
                                                      _lib.current_stream(torch.device(dev))))
    f(); torch.cuda.synchronize()
    sec = event_time(f, 10)
    out["splice_rows"] = entry("-", f"vllm_scatter_rows_bf16 alone ({idx.numel()} rows x {Cc} bf16, indices resident)", "hbm", ab, sec, 0, HBM_PEAK_GBS, "GB/s", 1e9,
                               algorithmic_bytes=ab, note="the kernel of `splice` without the index bookkeeping")
    return out


def _median_time(fn, reps=3, warmup=1):
    """1 warm-up + >= 3 timed repetitions, median."""
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


ALL_CORES_BUDGET_S = 20.0   # per repetition of the all-cores leg; beyond it the figure is reported as null, with the time seen

_ALL_CORES_CHILD = r"""
import sys, time, torch
from transformers import CLIPVisionConfig, CLIPVisionModel
cfg, th_ref, th_all = eval(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
with torch.no_grad():
    model = CLIPVisionModel(CLIPVisionConfig(**cfg, attn_implementation="eager")).eval()
    x = torch.randn(1, 3, 336, 336)
    def med(th, warm):
        torch.set_num_threads(th)
        ts = []
        for i in range(3 + warm):
            t0 = time.perf_counter(); model(pixel_values=x); ts.append(time.perf_counter() - t0)
            if i == 0 and warm:
                print("W", th, ts[0], flush=True)
        return sorted(ts[warm:])[1]
    print("R", th_ref, med(th_ref, 1), flush=True)
    print("R", th_all, med(th_all, 1), flush=True)
"""


def _all_cores_leg(th_ref, th_all):
    """ONE 336^2 tile through the full-depth ViT-L at th_ref and at th_all threads, 1 warm-up + 3 repetitions each, median -- in a
    child process, so that an oversubscribed all-cores repetition (73.7 s seen on a 256-thread host) can be abandoned at the budget
    instead of being waited for.  Returns (all_cores dict or None, note or None)."""
    import queue
    import subprocess
    import threading
    p = subprocess.Popen([sys.executable, "-c", _ALL_CORES_CHILD, repr(VIT), str(th_ref), str(th_all)], stdout=subprocess.PIPE,
                         stderr=subprocess.DEVNULL, text=True)
    q = queue.Queue()
    threading.Thread(target=lambda: [q.put(ln.split()) for ln in p.stdout] + [q.put(None)], daemon=True).start()
    got, t_ref, t_all = {}, None, None
    deadline = time.perf_counter() + 90.0          # start-up (imports) + the th_ref leg
    try:
        while True:
            ln = q.get(timeout=max(0.1, deadline - time.perf_counter()))
            if ln is None:
                break
            got[(ln[0], int(ln[1]))] = float(ln[2])
            if ln[0] == "R" and int(ln[1]) == th_ref:
                t_ref = float(ln[2])
                deadline = time.perf_counter() + ALL_CORES_BUDGET_S          # the all-cores warm-up must report within the budget
            elif ln[0] == "W" and int(ln[1]) == th_all:
                deadline = time.perf_counter() + 3 * ALL_CORES_BUDGET_S + 5
            elif ln[0] == "R" and int(ln[1]) == th_all:
                t_all = float(ln[2])
    except queue.Empty:
        pass
    finally:
        if p.poll() is None:
            p.kill()             # (the exact child we started)
        p.wait()
    if t_all is not None and t_ref is not None:
        return dict(value=1.0 / t_all, unit="tiles/sec (ViT-L only, one 336^2 tile, full depth)", cores=th_all,
                    same_sample_at_value_cores=1.0 / t_ref,
                    sample=f"1 warm-up + 3 repetitions, median: {t_all:.2f}s at {th_all} threads, {t_ref:.2f}s at {th_ref}"), None
    seen = got.get(("W", th_all))
    return None, (f"not measured: ONE 336^2 tile through ViT-L at torch.set_num_threads({th_all}) " +
                  (f"took {seen:.1f}s for the warm-up repetition and the three timed ones did not finish in {3 * ALL_CORES_BUDGET_S + 5:.0f}s" if seen is not None
                   else f"did not finish its warm-up repetition within the {ALL_CORES_BUDGET_S:.0f}s budget (73.7s seen on a 256-thread host of this pool)") +
                  (f"; the same sample at {th_ref} threads: {t_ref:.2f}s (median of 3)" if t_ref is not None else ""))


def cpu_baseline(ivit=False):
    """`value`: the reference CPU path at the thread count we measured fastest (32), 1 warm-up + 3 repetitions, median (BASELINE.md
    section 2).  `all_cores`: the SAME protocol at os.cpu_count() threads on a sample small enough to finish -- ONE 336^2 tile through
    the full-depth ViT-L (no bridge, no MSDA) -- reported in its own unit (tiles/sec) next to the same sample at `cores` threads; no
    scaling of one figure by another.  If the all-cores warm-up alone exceeds ALL_CORES_BUDGET_S (on the pool's 256-thread hosts
    torch.set_num_threads(256) oversubscribes these matrix sizes), `all_cores` is null and `all_cores_note` says what was seen."""
    main_leg = _cpu_leg(ivit, min(32, os.cpu_count() or 1), reps=3)
    n_all = os.cpu_count() or 1
    if ivit:
        return main_leg          # (the all-cores leg is measured once, on the headline workload)
    if n_all <= main_leg["cores"]:
        main_leg["all_cores"] = None
        main_leg["all_cores_note"] = f"`value` already uses every host thread ({n_all})"
        return main_leg
    try:
        main_leg["all_cores"], main_leg["all_cores_note"] = _all_cores_leg(main_leg["cores"], n_all)
    except Exception as e:   # never fail the bench line for the second leg
        main_leg["all_cores"] = None
        main_leg["all_cores_note"] = "error: " + repr(e)[:160]
    return main_leg


def _cpu_leg(ivit, threads, reps):
    """The reference's CPU path on the host cores of this box, bounded sample, extrapolated to images/sec.

    vitl: transformers.CLIPVisionModel (the class the reference instantiates, modeling_visionllmv2.py:135; eager attention,
    fp32) on the 5 tiles of ONE image + the mlp2x_gelu bridge (oracle restatement of :174-182) + the reference's pure-torch MSDA
    twin (oracle restatement of multi_scale_deform_attn.py:100-159, pinned to reference-run fixtures) at B=1 for the
    encoder and decoder shapes.  internvit6b: the InternViT restatement (oracle/vit.py, pinned to the reference class's
    fixtures) at 2 of 48 layers on the 5 tiles, extrapolated x24 (stated in `sample`).  ~10-25 s of CPU work."""
    from msda_inputs import make_inputs
    from oracle import msda as OM
    from oracle import vit as OV
    # torch's CPU kernels stop scaling (and thrash) far below the 256 hardware threads of the GPU box for these
    # single-tile problem sizes: 32 threads (or all, if fewer) is the fastest setting we measured
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    _mt = lambda fn: _median_time(fn, reps=reps)  # noqa: E731
    with torch.no_grad():
        if not ivit:
            from transformers import CLIPVisionConfig, CLIPVisionModel
            model = CLIPVisionModel(CLIPVisionConfig(**VIT, attn_implementation="eager")).eval()
            x = torch.randn(TILES_PER_IMAGE, 3, 336, 336)   # the 5 tiles of ONE 1336^2 image
            hs = [None]

            def run_vit():
                hs[0] = model(pixel_values=x, output_hidden_states=True).hidden_states
            t_vit = _mt(run_vit)
            C = VIT["hidden_size"]
            bsd = {"0.weight": torch.randn(LLM_HIDDEN, C) * 0.02, "0.bias": torch.zeros(LLM_HIDDEN),
                   "2.weight": torch.randn(LLM_HIDDEN, LLM_HIDDEN) * 0.02, "2.bias": torch.zeros(LLM_HIDDEN)}
            t_bridge = _mt(lambda: OV.bridge_forward(bsd, "mlp2x_gelu", hs[0][-2][:, 1:]))
            vit_note = (f"transformers CLIPVisionModel ViT-L/14-336 fp32 eager, 5 tiles x 24 layers ({t_vit:.2f}s) + mlp2x_gelu bridge ({t_bridge:.2f}s)")
        else:
            cfg = dict(IVIT, num_hidden_layers=2)
            C, I = IVIT["hidden_size"], IVIT["intermediate_size"]
            S = (IVIT["image_size"] // IVIT["patch_size"]) ** 2 + 1
            sd = {"embeddings.patch_embedding.weight": torch.randn(C, 3, 14, 14) * 0.02, "embeddings.patch_embedding.bias": torch.zeros(C),
                  "embeddings.class_embedding": torch.randn(1, 1, C) * 0.02, "embeddings.position_embedding": torch.randn(1, S, C) * 0.02}
            for i in range(2):
                pfx = f"encoder.layers.{i}."
                sd.update({pfx + "attn.qkv.weight": torch.randn(3 * C, C) * 0.02, pfx + "attn.q_norm.weight": torch.ones(C),
                           pfx + "attn.k_norm.weight": torch.ones(C), pfx + "attn.proj.weight": torch.randn(C, C) * 0.02,
                           pfx + "attn.proj.bias": torch.zeros(C), pfx + "mlp.fc1.weight": torch.randn(I, C) * 0.02,
                           pfx + "mlp.fc1.bias": torch.zeros(I), pfx + "mlp.fc2.weight": torch.randn(C, I) * 0.02,
                           pfx + "mlp.fc2.bias": torch.zeros(C), pfx + "norm1.weight": torch.ones(C), pfx + "norm2.weight": torch.ones(C),
                           pfx + "ls1": torch.full((C,), 0.1), pfx + "ls2": torch.full((C,), 0.1)})
            x = torch.randn(TILES_PER_IMAGE, 3, 448, 448)   # the 5 tiles of ONE image
            t2 = _mt(lambda: OV.intern_vit_forward(sd, cfg, x))
            t_vit = t2 * (IVIT["num_hidden_layers"] / 2)
            bsd = {"0.weight": torch.ones(4 * C), "0.bias": torch.zeros(4 * C), "1.weight": torch.randn(LLM_HIDDEN, 4 * C) * 0.02,
                   "1.bias": torch.zeros(LLM_HIDDEN), "3.weight": torch.randn(LLM_HIDDEN, LLM_HIDDEN) * 0.02, "3.bias": torch.zeros(LLM_HIDDEN)}
            feats = OV.select_features([torch.randn(TILES_PER_IMAGE, S, C)] * 2, -2, True)
            t_bridge = _mt(lambda: OV.bridge_forward(bsd, "internvl_mlp", feats))
            vit_note = (f"InternViT-6B restatement (oracle/vit.py) fp32, 5 tiles (448^2) x 2 of 48 layers ({t2:.2f}s) EXTRAPOLATED x24 = {t_vit:.1f}s "
                        f"+ pixel-shuffle + internvl_mlp bridge ({t_bridge:.2f}s)")
        g = make_inputs(1, MSDA["M"], MSDA["D"], MSDA["shapes"], MSDA["P"], mode="encoder_like", seed=0)
        tv, tl, tw = torch.from_numpy(g["value"]), torch.from_numpy(g["loc"]), torch.from_numpy(g["attw"])
        t_msda_enc = _mt(lambda: OM.grid_sample_twin(tv, g["shapes"].tolist(), tl, tw))
        gd = make_inputs(1, MSDA["M"], MSDA["D"], MSDA["shapes"], MSDA["P"], Lq=MSDA["dec_queries"], mode="encoder_like", seed=1)
        dv, dl, dw = torch.from_numpy(gd["value"]), torch.from_numpy(gd["loc"]), torch.from_numpy(gd["attw"])
        t_msda_dec = _mt(lambda: OM.grid_sample_twin(dv, gd["shapes"].tolist(), dl, dw))
    t_image = t_vit + t_bridge + MSDA["enc_layers"] * t_msda_enc + MSDA["dec_layers"] * t_msda_dec
    return dict(value=1.0 / t_image, unit="images/sec", cores=threads, kind="port",
                sample=(f"ONE image ({threads} of {os.cpu_count()} host threads, 1 warm-up + {reps} reps, median): {vit_note} + grid_sample MSDA twin B=1 "
                        f"Lq=37485 ({t_msda_enc:.2f}s) x6 + Lq=900 ({t_msda_dec:.2f}s) x6"))


def _respawn_under_launcher(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(sys.executable, cmd, env)


def read_in_step_profile(lib, n_steps):
    """{tag name: (us per launch, launches per step)} from the library's in-step recorder (vllm_prof_read)."""
    import ctypes
    n = 32
    us = (ctypes.c_double * n)()
    cnt = (ctypes.c_long * n)()
    got = lib.vllm_prof_read(us, cnt, n)
    out = {}
    for i in range(max(got, 0)):
        if cnt[i] > 0:
            out[lib.vllm_prof_tag_name(i).decode()] = (us[i] / cnt[i], cnt[i] / n_steps)
    return out


def run_workload(args, workload, dev, rank, world, dist, dry):
    """One workload end to end: build, warm up, time EXACTLY args.steps steps (barrier + synchronize on both sides, max over
    ranks), phases, in-step kernel times, rooflines.  Returns the record (rank 0) or None."""
    from visionllm_amd.dist import all_gather_visual_tokens
    from visionllm_amd.dist import shard_images
    ivit = workload == "internvit6b"
    n_tiles = IMAGES_PER_RANK * TILES_PER_IMAGE
    counts = [n_tiles] * world
    images_total = world * IMAGES_PER_RANK
    if args.ragged_tiles:
        # any-res batches are ragged: 1-7 tiles per image (mm_utils.py:39-77).  The GLOBAL batch of world x 8 images gets a fixed
        # 1..7 pattern, whole images are dealt to the ranks by tile count (dist.shard_images: every image's tiles stay on one
        # rank, modeling_visionllmv2.py:563), and every rank passes the tile counts of all ranks to the all-gather (no count
        # exchange, no host sync).  Not the headline workload: the line says so in config.workload.
        tiles_per_image = [1 + (i * i + 2 * i + 1) % 7 for i in range(images_total)]
        shards = shard_images(tiles_per_image, world)
        counts = [sum(tiles_per_image[i] for i in sh) for sh in shards]
        n_tiles = counts[rank]
    cfg = IVIT if ivit else VIT
    T = (cfg["image_size"] // cfg["patch_size"]) ** 2
    T_out = T // 4 if ivit else T
    if dry:
        # plumbing mode (CPU, gloo): the kernels are stubbed out IN THE BENCH ONLY -- a step is "produce a token tensor";
        # everything around it (respawn, rank / device mapping, process group, lagged collective + drain, phases, max over
        # ranks, JSON) is the code the GPU run executes
        tok_shape = (n_tiles, 8, 64)
        enc = bridge = None
        pixels = msda_in = None
        A = None
    else:
        from visionllm_amd import _lib
        from visionllm_amd import ms_deform_attn as A
        if args.encoder_chunks:
            from visionllm_amd import vit_common
            vit_common.set_encoder_chunks(args.encoder_chunks)
        enc, bridge = build_intern_model(dev) if ivit else build_model(dev)
        if ivit:
            enc.keep_hidden_states = (-1, -2, -3)   # 49 x 262 MB otherwise; the reference reads only these
        img = cfg["image_size"]
        gen = torch.Generator(device=dev).manual_seed(100 + rank)
        pixels = torch.randn(n_tiles, 3, img, img, device=dev, generator=gen).to(torch.bfloat16)
        msda_in = build_msda_inputs(dev, IMAGES_PER_RANK, 200 + rank)
        for t in msda_in.values():
            A.remember_geometry(t["shapes"])     # one read-back now (a det head's own shape check does it), one launch per call
        layer = None
        if rank == 0 and not ivit:
            try:
                layer = build_msda_layer(dev, IMAGES_PER_RANK)
                layer[0](); torch.cuda.synchronize()
            except Exception as e:   # a reported extra: never fail the bench for it
                print(f"bench: msda_layer entry skipped: {e!r}", file=sys.stderr)
                layer = None
    side = torch.cuda.Stream(device=dev) if (args.msda_stream and not dry) else None
    rot = [0]

    def msda_calls(res):
        if dry:
            return
        for tag, n in (("enc", MSDA["enc_layers"]), ("dec", MSDA["dec_layers"])):
            t = msda_in[tag]
            for _ in range(n):
                rot[0] += 1
                res.append(A.ms_deform_attn_forward(t["values"][rot[0] % len(t["values"])], t["shapes"], t["lsi"], t["loc"], t["attw"], 64))

    def sync():
        if not dry:
            torch.cuda.synchronize()

    class _Mark:
        def __init__(self):
            self.t = None
            self.e = None if dry else torch.cuda.Event(enable_timing=True)

        def record(self):
            if dry:
                self.t = time.perf_counter()
            else:
                self.e.record()

        def ms_to(self, other):
            return (other.t - self.t) * 1e3 if dry else self.e.elapsed_time(other.e)

    pending = [None]
    lag = [1]
    step_no = [0 if dry else -1]
    checked = [0]

    def check_gathered(g, k):
        """dry run: the tensor a step's collective delivered is [tiles of rank 0 | rank 1 | ...] of THAT step (ragged counts, lag 0 / 1)"""
        if not dry or g is None:
            return
        assert g.shape[0] == sum(counts), (g.shape, counts)
        o = 0
        for r, c in enumerate(counts):
            assert bool((g[o:o + c].float() == float(r) + float(k % 64)).all()), (r, k, g[o:o + c].flatten()[:4])
            o += c
        checked[0] += 1

    def step(marks=None):
        """marks: optional list receiving 4 marks (start, ViT+projector done, all-gather done, MSDA done); with marks the
        collective is waited for before the MSDA calls so that the three phases can be told apart."""
        res = []
        ev = lambda: (marks.append(_Mark()), marks[-1].record())  # noqa: E731
        if marks is not None:
            ev()
        main = None if dry else torch.cuda.current_stream(dev)
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                msda_calls(res)
        if dry:
            tokens = torch.full(tok_shape, float(rank), dtype=torch.bfloat16)
            if step_no[0] >= 0:
                tokens = tokens + float(step_no[0] % 64)    # (dry run: every step's tokens are recognisable, see `checked`)
                step_no[0] += 1
        else:
            out = enc(pixels, output_hidden_states=True)
            tokens = bridge.project_hidden_state(out.hidden_states[-2], ivit)
        if marks is not None:
            ev()
        # the token all-gather (RCCL over xGMI, its own stream) overlaps the det-head MSDA kernels of this step
        handle = all_gather_visual_tokens(tokens, counts=counts, async_op=True, algo=args.allgather)
        if marks is not None:
            gathered, _ = handle.wait()
            check_gathered(gathered, step_no[0] - 1)
            ev()
        if side is None:
            msda_calls(res)
        if marks is None:
            if lag[0] and world > 1:
                # software pipeline across steps: this step's collective is waited for one step later (before the next one is
                # launched), so it overlaps the NEXT step's encoder as well; the last one is drained before the timed region ends
                prev, pending[0] = pending[0], (handle, step_no[0] - 1)
                gathered = prev[0].wait()[0] if prev is not None else None
                if prev is not None:
                    check_gathered(gathered, prev[1])
            else:
                gathered, _ = handle.wait()
                check_gathered(gathered, step_no[0] - 1)
        if side is not None:
            main.wait_stream(side)
        if marks is not None:
            ev()
        res.append(gathered)
        return res

    def drain():
        if pending[0] is not None:
            check_gathered(pending[0][0].wait()[0], pending[0][1])
            pending[0] = None

    def timed(n_steps, lag_steps):
        """EXACTLY n_steps steps between (barrier + synchronize) pairs; the last lagged collective is drained inside."""
        lag[0] = lag_steps
        sync()
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        drain()                      # (the last step's collective belongs to the timed region)
        sync()
        if world > 1:
            dist.barrier()
        sync()
        return time.perf_counter() - t0

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        drain()
        # headline: the collective of a step is waited for INSIDE the step (lag 0: its cost is fully visible); for N > 1 the
        # pipelined schedule (lag 1: waited for one step later, overlapping the next encoder) is timed as well, same steps
        headline_lag = args.allgather_lag if world > 1 else 0
        sampler = ClockSampler(dev) if (not dry and rank == 0) else None
        sysfs = sampler is not None and sampler.has_sysfs()
        if sysfs:
            sampler.start(True)
        dt = timed(args.steps, headline_lag)
        clocks = sampler.stop() if sysfs else None
        if sampler is not None and clocks is None:
            # no hwmon files on this box: rocm-smi over a SECOND, untimed run of the same steps (>= 1.5 s of them)
            sampler.start(False)
            t_end = time.perf_counter() + 1.5
            while time.perf_counter() < t_end:
                step()
                sync()
            drain()
            clocks = sampler.stop()
            if clocks is not None:
                clocks["note"] = "sampled over an untimed repetition of the same steps right after the timed region"
        dt_alt = timed(args.steps, 1 - headline_lag) if world > 1 else None
        # per-phase times (outside the timed region): 3 instrumented steps, median per rank, max over ranks
        ph = []
        for _ in range(3):
            marks = []
            step(marks)
            sync()
            ph.append([marks[i].ms_to(marks[i + 1]) for i in range(3)])
        ph = np.median(np.array(ph), axis=0)
        # in-step kernel times: 3 more steps with the library's recorder on (an event in front of every operator)
        in_step = {}
        if not dry and rank == 0:
            from visionllm_amd import _lib
            L = _lib.lib()
            L.vllm_prof_enable(1)
            for _ in range(3):
                step()
                if layer is not None:
                    layer[0]()     # (behind the step, inside the recorder's window: the `msda_layer` tag)
            in_step = read_in_step_profile(L, 3)
            L.vllm_prof_enable(0)
        elif not dry:
            for _ in range(3):
                step()       # (all ranks run the same number of collectives)
        drain()
    red = torch.tensor([dt, ph[0], ph[1], ph[2], dt_alt if dt_alt is not None else 0.0], device=None if dry else dev, dtype=torch.float64)
    per_rank = [red.clone()]
    if world > 1:
        per_rank = [torch.empty_like(red) for _ in range(world)]
        dist.all_gather(per_rank, red)            # every rank's own clock and phases: the first SCALE run explains itself
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
    dt = float(red[0].item())
    if rank != 0:
        return None
    rec = {
        "value": world * IMAGES_PER_RANK * args.steps / dt,
        "ms_per_step": dt / args.steps * 1e3,
        "config": {"workload": ("internvit6b_448_5tiles+pixelshuffle+internvl_mlp+msda_cfg4" if ivit else
                                "vitl14_336_5tiles+mlp2x_gelu+msda_cfg4") + ("  [DRY RUN: kernels stubbed, CPU / gloo plumbing only]" if dry else ""),
                   "images_per_gpu": IMAGES_PER_RANK, "tiles_per_image": TILES_PER_IMAGE if not args.ragged_tiles else "1-7 (any-res pattern; NOT the headline workload)", "image": "1336x1336",
                   "vit": "InternViT-6B 48L bf16 (448^2 tiles)" if ivit else "ViT-L/14-336 24L bf16",
                   "bridge": "pixel_shuffle + internvl_mlp 12800->4096->4096" if ivit else "mlp2x_gelu 1024->4096->4096",
                   "msda": "B8 M8 D32 L4 P4 168^2..21^2 fp32, 6x Lq=37485 + 6x Lq=900 (3 rotating value buffers)",
                   "parallelism": f"dp{world}" + ("+allgather(tokens)" if world > 1 else ""),
                   "rccl_ranks": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
                   "backend": (dist.get_backend() if (world > 1 and dist.is_initialized()) else "none"),
                   "allgather": args.allgather, "allgather_lag_steps": headline_lag,
                   "tiles_per_rank": counts, "ragged_tiles": bool(args.ragged_tiles),
                   "streams": ("vit+projector | msda (side stream; cross-batch pipelining)" if args.msda_stream else "single") +
                              ("" if args.encoder_chunks <= 1 else f"; vit tiles as {args.encoder_chunks} chunks on separate streams")},
        "phases_ms": {"vit_projector": float(red[1].item()), "token_allgather": float(red[2].item()), "msda_12_calls": float(red[3].item()),
                      "note": "3 instrumented steps after the timed region (collective waited for before the MSDA calls), median per rank, max over ranks"},
        "per_rank": [{"rank": r, "timed_s": float(v[0].item()), "vit_projector_ms": float(v[1].item()), "token_allgather_ms": float(v[2].item()),
                      "msda_12_calls_ms": float(v[3].item()), "tiles": counts[r]} for r, v in enumerate(per_rank)],
    }
    if dry:
        rec["dry_run_collectives_checked"] = checked[0]
    if not dry:
        rec["clocks"] = clocks
    if world > 1:
        alt = float(red[4].item())
        rec["allgather_lag_alt"] = {"allgather_lag_steps": 1 - headline_lag, "value": world * IMAGES_PER_RANK * args.steps / alt,
                                    "ms_per_step": alt / args.steps * 1e3,
                                    "note": "the same steps with the other collective schedule (lag 1 = a step's tokens are waited for one step "
                                            "later and overlap the next encoder; lag 0 = inside the step)"}
    if not dry:
        bdims = ([(n_tiles * T // 4, LLM_HIDDEN, 4 * cfg["hidden_size"]), (n_tiles * T // 4, LLM_HIDDEN, LLM_HIDDEN)] if ivit else
                 [(n_tiles * T, LLM_HIDDEN, cfg["hidden_size"]), (n_tiles * T, LLM_HIDDEN, LLM_HIDDEN)])
        rl = kernel_rooflines(dev, msda_in, n_tiles, cfg, bdims, workload=workload, in_step=in_step, layer=layer)
        step_us = dt / args.steps * 1e6
        for v in rl.values():
            if "us_per_launch" in v:
                v["share_of_step"] = v["launches_per_step"] * v["us_per_launch"] / step_us
        dom = max((k for k in rl if "share_of_step" in rl[k]), key=lambda k: rl[k]["share_of_step"])
        rec["roofline"] = {k: rl[dom][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} | \
                          {"kernel": rl[dom]["kernel"], "share_of_step": rl[dom]["share_of_step"], "us_per_launch": rl[dom]["us_per_launch"],
                           "timing": rl[dom]["timing"]}
        rec["rooflines"] = rl
        rec["in_step_us_per_launch"] = {k: round(v[0], 2) for k, v in in_step.items()}
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(ivit)
    del enc, bridge, pixels, msda_in
    layer = None
    if not dry:
        torch.cuda.empty_cache()
    return rec


LINE_LIMIT = 6144   # bytes: the driver keeps an 8 KB stdout tail; the round-5 line (26 KB) did not parse


def _sig(x, n=5):
    """floats to n significant digits (the detail file keeps full precision); NaN / inf -> None (strict JSON)"""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{n}g}")
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _compact_rooflines(rl):
    """{name: {frac, us_per_launch, launches_per_step, traffic}} for the kernels that run inside the step (+ the fused MSDA layer,
    which runs behind it in the instrumented steps) and {name: frac} for the isolated SURVEY 8(f) rows."""
    keep, iso = {}, {}
    for k, v in rl.items():
        if "frac" not in v:
            continue
        if v.get("launches_per_step", 0) or k == "msda_layer":
            keep[k] = {"frac": v["frac"], "us_per_launch": v["us_per_launch"], "launches_per_step": v["launches_per_step"],
                       "traffic": v.get("traffic")}
        else:
            iso[k] = v["frac"]
    return keep, iso


def _short_kernel(name):
    return name if len(name) <= 120 else name[:117] + "..."


def compact_record(rec):
    """The part of a workload's record that goes on the (<= 6 KB) final line; the full record goes to the detail file."""
    out = {k: rec[k] for k in ("value", "ms_per_step", "config", "clocks", "dry_run_collectives_checked") if k in rec}
    if "phases_ms" in rec:
        out["phases_ms"] = {k: v for k, v in rec["phases_ms"].items() if k != "note"}
    if len(rec.get("per_rank", [])) > 1:
        out["per_rank"] = rec["per_rank"]
    if "allgather_lag_alt" in rec:
        out["allgather_lag_alt"] = {k: v for k, v in rec["allgather_lag_alt"].items() if k != "note"}
    if "clocks" in out and out["clocks"]:
        out["clocks"] = {k: v for k, v in out["clocks"].items() if k in ("sclk_mhz_median", "sclk_mhz_min", "power_w_median", "samples")}
    if "roofline" in rec:
        out["roofline"] = dict(rec["roofline"], kernel=_short_kernel(rec["roofline"]["kernel"]))
        out["rooflines"], out["isolated_frac"] = _compact_rooflines(rec["rooflines"])
    if "cpu_baseline" in rec:
        cb = dict(rec["cpu_baseline"])
        cb["sample"] = cb["sample"][:300]
        if isinstance(cb.get("all_cores"), dict) and "sample" in cb["all_cores"]:
            cb["all_cores"] = dict(cb["all_cores"], sample=cb["all_cores"]["sample"][:200])
        if "all_cores_note" in cb:
            cb["all_cores_note"] = cb["all_cores_note"][:260]
        out["cpu_baseline"] = cb
    return out


def emit(line_full, head, rec, extra):
    """Full record -> gpurun_out/bench_detail.json (+ stderr); ONE compact JSON line (<= LINE_LIMIT bytes, strict JSON) -> stdout, last."""
    line = dict(head)
    line.update({k: v for k, v in compact_record(rec).items() if k not in ("value", "ms_per_step")})
    if extra is not None:
        ce = compact_record(extra)
        ce["config"] = {"workload": ce["config"]["workload"], "vit": ce["config"]["vit"], "bridge": ce["config"]["bridge"]}
        ce.pop("clocks", None)
        ce.pop("phases_ms", None)
        ce.pop("isolated_frac", None)
        if "cpu_baseline" in ce:
            ce["cpu_baseline"].pop("all_cores", None)
            ce["cpu_baseline"].pop("all_cores_note", None)
        line["internvit6b"] = ce
    line = _sig(line)
    detail = None
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        detail = os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        with open(detail, "w") as f:
            json.dump(_sig(line_full, 9), f, allow_nan=False)
        line["detail"] = "gpurun_out/bench_detail.json (every kernel's name, notes, isolated 8(f) rows; also on stderr)"
    except Exception:
        pass
    txt = json.dumps(line, allow_nan=False, separators=(",", ":"))
    # belt and braces: shed the optional parts, largest first, until the line fits
    for k in ("isolated_frac", "per_rank", "phases_ms", "clocks"):
        if len(txt) <= LINE_LIMIT:
            break
        line.pop(k, None)
        if "internvit6b" in line:
            line["internvit6b"].pop(k, None)
        txt = json.dumps(line, allow_nan=False, separators=(",", ":"))
    assert len(txt) <= LINE_LIMIT, len(txt)
    sys.stderr.write("bench detail: " + json.dumps(_sig(line_full, 9)) + "\n")
    sys.stderr.flush()
    sys.stdout.write(txt + "\n")
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="auto", choices=["auto", "vitl", "internvit6b", "both"],
                    help="vitl: the metric's ViT-L config; internvit6b: BASELINE configs[2] (5 tiles of 448^2 per image through "
                         "InternViT-6B + pixel-shuffle + internvl_mlp projector); both: the vitl line with the InternViT-6B record under "
                         "the key \"internvit6b\"; auto (default): both on one GPU, vitl on several")
    ap.add_argument("--msda-stream", type=int, default=0, choices=[0, 1],
                    help="0 (default): one stream; 1: the MSDA calls on a side stream next to the ViT (cross-batch pipelining: in "
                         "the reference the det head of a batch depends on that batch's LLM output)")
    ap.add_argument("--allgather-lag", type=int, default=0, choices=[0, 1],
                    help="N > 1, the schedule of the HEADLINE number: 0 (default) = a step's token all-gather is waited for inside the "
                         "step; 1 = one step later (overlaps the next step's encoder; drained inside the timed region).  The other "
                         "schedule is timed too and reported under \"allgather_lag_alt\"")
    ap.add_argument("--allgather", default="collective", choices=["collective", "direct"],
                    help="token all-gather: RCCL all_gather_into_tensor (default) or batched point-to-point to all peers at once")
    ap.add_argument("--encoder-chunks", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="0 / 1 (default): one launch sequence; k: the tiles as k chunks on k streams (measured slower)")
    ap.add_argument("--ragged-tiles", action="store_true",
                    help="any-res batch: the world x 8 images get 1-7 tiles each and are dealt to the ranks by tile count "
                         "(dist.shard_images); exercises ragged shards in the token all-gather.  Not the headline workload")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU plumbing check (no GPU, gloo): launcher respawn, rank mapping, process group, lagged collective + drain, "
                         "phases, JSON -- with the kernels stubbed out in the bench only; the line says so and is NOT a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn_under_launcher(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    import torch.distributed as dist
    if args.dry_run:
        dev = "cpu"
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
        torch.cuda.set_device(local_rank)
        dev = f"cuda:{local_rank}"
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    workload = args.workload
    if workload == "auto":
        workload = "both" if (world == 1 and not args.dry_run) else "vitl"
    first = "internvit6b" if workload == "internvit6b" else "vitl"
    rec = run_workload(args, first, dev, rank, world, dist, args.dry_run)
    extra = run_workload(args, "internvit6b", dev, rank, world, dist, args.dry_run) if workload == "both" else None
    if rank == 0:
        head = {
            "metric": "images/sec (ViT-L+projector+MSDeformAttn fwd, 1336px)",
            "value": rec["value"],
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
        }
        full = dict(head)
        full.update({k: v for k, v in rec.items() if k not in ("value", "ms_per_step")})
        if extra is not None:
            full["internvit6b"] = dict(extra, metric="images/sec (InternViT-6B+pixel-shuffle+internvl_mlp+MSDeformAttn fwd, 1336px; BASELINE configs[2])",
                                       unit="images/sec", n_gpus=world, steps=args.steps, warmup=args.warmup)
        emit(full, head, rec, extra)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
