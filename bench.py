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
        out[tag] = t
    return out


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


def kernel_rooflines(dev, msda_in, n_tiles, cfg, bridge_dims, iters=10, workload="vitl"):
    """Achieved vs peak of every hot kernel of the workload, each launched alone (HIP events on torch's current stream =
    the stream the C ABI launches on) + its launches per step.  cfg: VIT or IVIT; bridge_dims: [(M, N, K), ...]."""
    from visionllm_amd import _lib
    from visionllm_amd import ms_deform_attn as A
    L = _lib.lib()
    st = _lib.current_stream(torch.device(dev))
    C, H, I, NL = cfg["hidden_size"], cfg["num_attention_heads"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    S = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
    D = C // H
    M = n_tiles * S
    act = 2 if cfg["hidden_act"] == "quick_gelu" else 1
    out = {}
    # MSDA: HBM bound, algorithmic bytes = value + loc + attw + out
    for tag, nm, n_launch in (("enc", "msda", MSDA["enc_layers"]), ("dec", "msda_dec", MSDA["dec_layers"])):
        t = msda_in[tag]
        f = lambda: A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)  # noqa: E731
        f(); torch.cuda.synchronize()
        sec = event_time(f, iters)
        ab = msda_bytes(t)
        kern = ("msda_fwd_tiled7_kernel<12 waves, 1 block per CU> (fp32, D32, pyramid items, encoder shape Lq=S=37485, B=8)"
                if tag == "enc" else "msda_fwd_vec_kernel (fp32, D32, decoder shape Lq=900, B=8)")
        out[nm] = dict(kernel=kern, bound="hbm", achieved=ab / sec / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                       frac=ab / sec / 1e9 / HBM_PEAK_GBS, traffic=None, us_per_launch=sec * 1e6, algorithmic_bytes=ab,
                       launches_per_step=n_launch)
    # attention: MFMA bound, flops = 4*H*S^2*d per tile
    qkv = torch.randn(n_tiles, S, 3, H, D, device=dev).to(torch.bfloat16)
    ao = torch.empty(n_tiles, S, H, D, dtype=torch.bfloat16, device=dev)
    f = lambda: _lib.check(L.vllm_attn_fwd_qkvpacked_bf16(_lib.ptr(qkv), _lib.ptr(ao), n_tiles, S, H, D, D ** -0.5, st))  # noqa: E731
    f(); torch.cuda.synchronize()
    sec = event_time(f, iters)
    fl = 4.0 * H * S * S * D * n_tiles
    out["attn"] = dict(kernel=f"attn_fwd_kernel<D{D}> ({n_tiles} tiles x {H} heads x S{S})", bound="mfma", achieved=fl / sec / 1e12,
                       peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s", frac=fl / sec / 1e12 / MFMA_BF16_PEAK_TF, traffic=None,
                       us_per_launch=sec * 1e6, algorithmic_flops=fl, launches_per_step=NL)
    del qkv, ao
    # the four encoder GEMMs + the projector GEMMs
    # epilogues as the encoder uses them: 3 = bias (+ LayerScale for InternViT) + residual
    gemms = [("gemm_qkv", M, 3 * C, C, 0, NL), ("gemm_proj", M, C, C, 3, NL), ("gemm", M, I, C, act, NL), ("gemm_fc2", M, C, I, 3, NL)]
    gemms += [(f"gemm_bridge{i}", m, n, k, 0, 1) for i, (m, n, k) in enumerate(bridge_dims)]
    has_ls = "qk_normalization" in cfg   # InternViT: LayerScale in the residual epilogue; CLIP: plain residual
    names = {"gemm": "MLP fc1 + activation", "gemm_qkv": "QKV", "gemm_proj": "attention out-proj + " + ("LayerScale + " if has_ls else "") + "residual",
             "gemm_fc2": "MLP fc2 + " + ("LayerScale + " if has_ls else "") + "residual"}
    for nm, m, n, k, epi, n_launch in gemms:
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
        b = torch.zeros(n, device=dev).to(torch.bfloat16)
        y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        res = torch.randn(m, n, device=dev).to(torch.bfloat16) if epi == 3 else None
        ls = torch.full((n,), 0.1, device=dev).to(torch.bfloat16) if (epi == 3 and has_ls) else None
        f = lambda: _lib.check(L.vllm_gemm_bf16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(b), _lib.ptr(y), m, n, k, k, k, n, epi,  # noqa: E731
                                                 _lib.ptr(ls), _lib.ptr(res), n if epi == 3 else 0, 0, st))
        f(); torch.cuda.synchronize()
        sec = event_time(f, iters)
        fl = 2.0 * m * n * k
        out[nm] = dict(kernel=f"gemm256_bf16_kernel ({names.get(nm, 'projector linear')}: M{m} N{n} K{k})", bound="mfma",
                       achieved=fl / sec / 1e12, peak=MFMA_BF16_PEAK_TF, unit="TFLOP/s", frac=fl / sec / 1e12 / MFMA_BF16_PEAK_TF,
                       traffic=None, us_per_launch=sec * 1e6, algorithmic_flops=fl, launches_per_step=n_launch)
        del x, w, b, y, res, ls
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


def cpu_baseline(ivit=False):
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
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    with torch.no_grad():
        if not ivit:
            from transformers import CLIPVisionConfig, CLIPVisionModel
            model = CLIPVisionModel(CLIPVisionConfig(**VIT, attn_implementation="eager")).eval()
            x = torch.randn(TILES_PER_IMAGE, 3, 336, 336)   # the 5 tiles of ONE 1336^2 image
            hs = [None]

            def run_vit():
                hs[0] = model(pixel_values=x, output_hidden_states=True).hidden_states
            t_vit = _median_time(run_vit)
            C = VIT["hidden_size"]
            bsd = {"0.weight": torch.randn(LLM_HIDDEN, C) * 0.02, "0.bias": torch.zeros(LLM_HIDDEN),
                   "2.weight": torch.randn(LLM_HIDDEN, LLM_HIDDEN) * 0.02, "2.bias": torch.zeros(LLM_HIDDEN)}
            t_bridge = _median_time(lambda: OV.bridge_forward(bsd, "mlp2x_gelu", hs[0][-2][:, 1:]))
            vit_note = (f"transformers {__import__('transformers').__version__} CLIPVisionModel ViT-L/14-336 fp32 eager, the 5 tiles of one "
                        f"image x 24 layers ({t_vit:.2f}s) + mlp2x_gelu bridge on them ({t_bridge:.2f}s)")
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
            t2 = _median_time(lambda: OV.intern_vit_forward(sd, cfg, x))
            t_vit = t2 * (IVIT["num_hidden_layers"] / 2)
            bsd = {"0.weight": torch.ones(4 * C), "0.bias": torch.zeros(4 * C), "1.weight": torch.randn(LLM_HIDDEN, 4 * C) * 0.02,
                   "1.bias": torch.zeros(LLM_HIDDEN), "3.weight": torch.randn(LLM_HIDDEN, LLM_HIDDEN) * 0.02, "3.bias": torch.zeros(LLM_HIDDEN)}
            feats = OV.select_features([torch.randn(TILES_PER_IMAGE, S, C)] * 2, -2, True)
            t_bridge = _median_time(lambda: OV.bridge_forward(bsd, "internvl_mlp", feats))
            vit_note = (f"InternViT-6B restatement (oracle/vit.py) fp32, the 5 tiles (448^2) of one image x 2 of 48 layers ({t2:.2f}s), "
                        f"EXTRAPOLATED x24 = {t_vit:.1f}s + pixel-shuffle + internvl_mlp bridge on them ({t_bridge:.2f}s)")
        g = make_inputs(1, MSDA["M"], MSDA["D"], MSDA["shapes"], MSDA["P"], mode="encoder_like", seed=0)
        tv, tl, tw = torch.from_numpy(g["value"]), torch.from_numpy(g["loc"]), torch.from_numpy(g["attw"])
        t_msda_enc = _median_time(lambda: OM.grid_sample_twin(tv, g["shapes"].tolist(), tl, tw))
        gd = make_inputs(1, MSDA["M"], MSDA["D"], MSDA["shapes"], MSDA["P"], Lq=MSDA["dec_queries"], mode="encoder_like", seed=1)
        dv, dl, dw = torch.from_numpy(gd["value"]), torch.from_numpy(gd["loc"]), torch.from_numpy(gd["attw"])
        t_msda_dec = _median_time(lambda: OM.grid_sample_twin(dv, gd["shapes"].tolist(), dl, dw))
    t_image = t_vit + t_bridge + MSDA["enc_layers"] * t_msda_enc + MSDA["dec_layers"] * t_msda_dec
    return dict(value=1.0 / t_image, unit="images/sec", cores=threads, kind="port",
                sample=(f"{threads} host threads, 1 warm-up + 3 repetitions each, median: {vit_note} + reference grid_sample MSDA twin B=1 "
                        f"Lq=37485 ({t_msda_enc:.2f}s) and Lq=900 ({t_msda_dec:.2f}s); sample = ONE image: its 5 tiles in one batch + 6 encoder-shaped + 6 "
                        f"decoder-shaped MSDA calls at B=1 (one call of each shape timed, x6)"))


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="vitl", choices=["vitl", "internvit6b"],
                    help="vitl (default; the metric's ViT-L config) or internvit6b (BASELINE configs[2]: 5 tiles of 448^2 per "
                         "image through InternViT-6B + pixel-shuffle + internvl_mlp projector)")
    ap.add_argument("--msda-stream", type=int, default=0, choices=[0, 1],
                    help="0 (default): one stream; 1: the MSDA calls on a side stream next to the ViT (cross-batch pipelining: in "
                         "the reference the det head of a batch depends on that batch's LLM output)")
    ap.add_argument("--allgather-lag", type=int, default=1, choices=[0, 1],
                    help="N > 1: 1 (default) = a step's token all-gather is waited for one step later, so it overlaps the next "
                         "step's encoder too (steady-state pipeline; everything is drained inside the timed region); 0 = waited "
                         "for at the end of its own step")
    ap.add_argument("--allgather", default="collective", choices=["collective", "direct"],
                    help="token all-gather: RCCL all_gather_into_tensor (default) or batched point-to-point to all peers at once")
    ap.add_argument("--encoder-chunks", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="0 / 1 (default): one launch sequence; k: the tiles as k chunks on k streams (measured slower)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn_under_launcher(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the product path)"
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    from visionllm_amd import ms_deform_attn as A
    from visionllm_amd.dist import all_gather_visual_tokens

    ivit = args.workload == "internvit6b"
    if args.encoder_chunks:
        from visionllm_amd import vit_common
        vit_common.set_encoder_chunks(args.encoder_chunks)
    enc, bridge = build_intern_model(dev) if ivit else build_model(dev)
    if ivit:
        enc.keep_hidden_states = (-1, -2, -3)   # 49 x 262 MB otherwise; the reference reads only these
    n_tiles = IMAGES_PER_RANK * TILES_PER_IMAGE
    img = IVIT["image_size"] if ivit else VIT["image_size"]
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    pixels = torch.randn(n_tiles, 3, img, img, device=dev, generator=gen).to(torch.bfloat16)
    msda_in = build_msda_inputs(dev, IMAGES_PER_RANK, 200 + rank)
    side = torch.cuda.Stream(device=dev) if args.msda_stream else None

    def msda_calls(res):
        for tag, n in (("enc", MSDA["enc_layers"]), ("dec", MSDA["dec_layers"])):
            t = msda_in[tag]
            for _ in range(n):
                res.append(A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64))

    def step(marks=None):
        """marks: optional list receiving 4 HIP events (start, ViT+projector done, all-gather done, MSDA done); with marks the
        collective is waited for before the MSDA calls so that the three phases can be told apart."""
        res = []
        main = torch.cuda.current_stream(dev)
        ev = lambda: (marks.append(torch.cuda.Event(enable_timing=True)), marks[-1].record())  # noqa: E731
        if marks is not None:
            ev()
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                msda_calls(res)
        out = enc(pixels, output_hidden_states=True)
        tokens = bridge.project_hidden_state(out.hidden_states[-2], ivit)
        if marks is not None:
            ev()
        # the token all-gather (RCCL over xGMI, its own stream) overlaps the det-head MSDA kernels of this step
        handle = all_gather_visual_tokens(tokens, counts=[tokens.shape[0]] * world, async_op=True, algo=args.allgather)
        if marks is not None:
            gathered, _ = handle.wait()
            ev()
        if side is None:
            msda_calls(res)
        if marks is None:
            if args.allgather_lag and world > 1:
                # software pipeline across steps: this step's collective is waited for one step later (before the next one is
                # launched), so it overlaps the NEXT step's encoder as well; the last one is drained before the timed region ends
                prev, pending[0] = pending[0], handle
                gathered = prev.wait()[0] if prev is not None else None
            else:
                gathered, _ = handle.wait()
        if side is not None:
            main.wait_stream(side)
        if marks is not None:
            ev()
        res.append(gathered)
        return res

    pending = [None]

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    with torch.no_grad():
        for _ in range(args.warmup):
            step()
        drain()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()                      # (the last step's collective belongs to the timed region)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        # per-phase times (outside the timed region): 3 instrumented steps, median per rank, max over ranks
        ph = []
        for _ in range(3):
            marks = []
            step(marks)
            torch.cuda.synchronize()
            ph.append([marks[i].elapsed_time(marks[i + 1]) for i in range(3)])
        ph = np.median(np.array(ph), axis=0)
    red = torch.tensor([dt, ph[0], ph[1], ph[2]], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(red, op=dist.ReduceOp.MAX)
    dt = float(red[0].item())

    if rank == 0:
        cfg = IVIT if ivit else VIT
        T = (cfg["image_size"] // cfg["patch_size"]) ** 2
        bdims = ([(n_tiles * T // 4, LLM_HIDDEN, 4 * cfg["hidden_size"]), (n_tiles * T // 4, LLM_HIDDEN, LLM_HIDDEN)] if ivit else
                 [(n_tiles * T, LLM_HIDDEN, cfg["hidden_size"]), (n_tiles * T, LLM_HIDDEN, LLM_HIDDEN)])
        rl = kernel_rooflines(dev, msda_in, n_tiles, cfg, bdims, workload=args.workload)
        step_us = dt / args.steps * 1e6
        for v in rl.values():
            v["share_of_step"] = v["launches_per_step"] * v["us_per_launch"] / step_us
        dom = max(rl, key=lambda k: rl[k]["share_of_step"])
        line = {
            "metric": "images/sec (ViT-L+projector+MSDeformAttn fwd, 1336px)",
            "value": world * IMAGES_PER_RANK * args.steps / dt,
            "unit": "images/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": ("internvit6b_448_5tiles+pixelshuffle+internvl_mlp+msda_cfg4" if ivit else
                                    "vitl14_336_5tiles+mlp2x_gelu+msda_cfg4"), "images_per_gpu": IMAGES_PER_RANK,
                       "tiles_per_image": TILES_PER_IMAGE, "image": "1336x1336",
                       "vit": "InternViT-6B 48L bf16 (448^2 tiles)" if ivit else "ViT-L/14-336 24L bf16",
                       "bridge": "pixel_shuffle + internvl_mlp 12800->4096->4096" if ivit else "mlp2x_gelu 1024->4096->4096",
                       "msda": "B8 M8 D32 L4 P4 168^2..21^2 fp32, 6x Lq=37485 + 6x Lq=900",
                       "parallelism": f"dp{world}" + ("+allgather(tokens)" if world > 1 else ""),
                       "rccl_ranks": world, "allgather": args.allgather, "allgather_lag_steps": args.allgather_lag if world > 1 else 0,
                       "streams": ("vit+projector | msda (side stream; cross-batch pipelining)" if args.msda_stream else "single") +
                                  ("" if args.encoder_chunks <= 1 else f"; vit tiles as {args.encoder_chunks} chunks on separate streams")},
            "phases_ms": {"vit_projector": float(red[1].item()), "token_allgather": float(red[2].item()), "msda_12_calls": float(red[3].item()),
                          "note": "3 instrumented steps after the timed region (collective waited for before the MSDA calls), median per rank, max over ranks"},
            "roofline": {k: rl[dom][k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")} |
                        {"kernel": rl[dom]["kernel"], "share_of_step": rl[dom]["share_of_step"]},
            "rooflines": rl,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(ivit)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
