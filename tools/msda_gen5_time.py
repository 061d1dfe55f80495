import json, os, sys, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from visionllm_amd import _lib, ms_deform_attn as A
t = bench.build_msda_inputs("cuda:0", 8, 200)["enc"]
def timeit(iters=10):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
out = {}
for name, v in (("gen4", 1), ("gen5_4prod", 6), ("gen5_8prod", 7)):
    _lib.set_option("msda_tiled", v); timeit(3)
    out[name] = round(min(timeit() for _ in range(3)), 1)
print(os.environ.get("VLLM_HIP_LIB", "in-tree")[-28:], json.dumps(out))
