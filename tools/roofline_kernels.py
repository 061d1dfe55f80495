#!/usr/bin/env python3
"""Launch the three roofline kernels of bench.py a few times each (target of the rocprofv3 --pmc passes)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

dev = "cuda:0"
torch.cuda.set_device(0)
msda_in = bench.build_msda_inputs(dev, bench.IMAGES_PER_RANK, 200)
rl = bench.kernel_rooflines(dev, None, msda_in, bench.IMAGES_PER_RANK * bench.TILES_PER_IMAGE, iters=5)
print(json.dumps(rl))
