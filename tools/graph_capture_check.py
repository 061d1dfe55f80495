"""hipGraph capture of one bench step (ViT-L + bridge on the main stream, MSDA on the side stream) replayed against the eager
step: measured 23.05 vs 23.02 ms -- the native runtime enqueues a step in 1.7 ms of CPU time, the GPU is never starved, so
bench.py stays eager."""
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from visionllm_amd import ms_deform_attn as A
dev="cuda:0"; torch.cuda.set_device(0)
enc, bridge = bench.build_model(dev)
n_tiles = bench.IMAGES_PER_RANK * bench.TILES_PER_IMAGE
pixels = torch.randn(n_tiles, 3, 336, 336, device=dev).to(torch.bfloat16)
msda_in = bench.build_msda_inputs(dev, bench.IMAGES_PER_RANK, 200)
side = torch.cuda.Stream(device=dev)
def msda_calls(res):
    for tag, n in (("enc", bench.MSDA["enc_layers"]), ("dec", bench.MSDA["dec_layers"])):
        t = msda_in[tag]
        for _ in range(n):
            res.append(A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64))
def step():
    res = []
    main = torch.cuda.current_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        msda_calls(res)
    out = enc(pixels, output_hidden_states=True)
    tokens = bridge.project_hidden_state(out.hidden_states[-2], False)
    main.wait_stream(side)
    res.append(tokens)
    return res
def timeit(fn, k=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/k*1e3
print("eager ms/step", timeit(step))
# CPU-side enqueue time
torch.cuda.synchronize(); t0=time.perf_counter(); step(); t1=time.perf_counter(); torch.cuda.synchronize()
print("cpu enqueue ms", (t1-t0)*1e3)
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = step()
    print("graph ms/step", timeit(g.replay))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
