import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from visionllm_amd import _lib, ms_deform_attn as A
dev = "cuda:0"
t = bench.build_msda_inputs(dev, 8, 200)["enc"]
for v in [int(x) for x in os.environ.get("MSDA_MODES", "9,1,15").split(",")]:   # generation 4, 6, 7
    _lib.set_option("msda_tiled", v)
    for _ in range(3):
        A.ms_deform_attn_forward(t["value"], t["shapes"], t["lsi"], t["loc"], t["attw"], 64)
torch.cuda.synchronize()
