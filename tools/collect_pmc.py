#!/usr/bin/env python3
"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes -> profiles/pmc_traffic.json (HBM bytes per launch, per bench workload).

    python tools/collect_pmc.py <fetch_dir> <write_dir> <out.json> [vitl|internvit6b]

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports half of the bytes of a
wide coalesced read stream, so reads are doubled; both counters are in KiB.  Gather-heavy kernels (MSDA) are not a wide
coalesced stream, so their figure is an upper-bound style estimate and is flagged as such."""
import collections
import csv
import glob
import json
import os
import sys

# kernel-name fragments -> bench.py roofline keys ("gemm" = MLP fc1: the quick_gelu (ViT-L) / gelu (InternViT) epilogue instance)
# (fc1 runs on the persistent schedule, gemm256p_kernel<EPI, MT, folded-norm consumer, statistics>; the one-workgroup-per-tile
#  names are kept for A/B passes with VLLM_GEMM_PERSIST=0)
KEYS = {"vitl": {"msda_fwd_tiled7": "msda", "msda_fwd_tiled8_kernel<1200, false, true>": "msda", "msda_fwd_tiled9_kernel<1200, false, true>": "msda", "attn_fwd_kernel": "attn", "gemm256p_kernel<2,": "gemm",
                 "gemm256p_kernelILi2": "gemm", "gemm256_bf16_kernel<2,": "gemm", "gemm256_bf16_kernelILi2": "gemm"},
        "internvit6b": {"msda_fwd_tiled7": "msda", "msda_fwd_tiled8_kernel<1200, false, true>": "msda", "msda_fwd_tiled9_kernel<1200, false, true>": "msda", "attn_fwd_kernel": "attn", "gemm256p_kernel<1,": "gemm",
                        "gemm256p_kernelILi1": "gemm", "gemm256_bf16_kernel<1,": "gemm", "gemm256_bf16_kernelILi1": "gemm"}}


# round 6: the other in-step GEMMs.  qkv = the bias epilogue that CONSUMES a folded norm (template flag LNC = true); proj and fc2 are
# the SAME instantiation (residual epilogue + statistics): told apart by the kernel dispatched in front of them.
EXTRA = {"gemm256p_kernel<0, 4, true": "gemm_qkv", "gemm256p_kernel<0, 3, true": "gemm_qkv"}
RESIDUAL = ("gemm256p_kernel<3, 4, false, 1", "gemm256p_kernel<3, 3, false, 1", "gemm256p_kernel<3, 4, false, 2", "gemm256p_kernel<3, 3, false, 2")


def load(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for path in f:
        rows = list(csv.DictReader(open(path)))
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0) or 0))
        prev = ""
        for r in rows:
            kn = r["Kernel_Name"]
            if any(p in kn for p in RESIDUAL):
                # a layer is qkv, attention, proj, fc1, fc2: the residual GEMM behind the attention kernel is proj, the one behind
                # an activation-epilogue GEMM is fc2
                if "attn_fwd_kernel" in prev:
                    kn = "RESIDUAL_proj " + kn
                elif "gemm256p_kernel<1," in prev or "gemm256p_kernel<2," in prev:
                    kn = "RESIDUAL_fc2 " + kn
            if r["Kernel_Name"] != prev and r.get("Counter_Name"):
                pass
            prev_name = r["Kernel_Name"]
            agg[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
            prev = prev_name
    return agg


def main(fetch_dir, write_dir, out, workload="vitl"):
    KEY = dict(KEYS[workload], **EXTRA)
    KEY["RESIDUAL_proj "] = "gemm_proj"
    KEY["RESIDUAL_fc2 "] = "gemm_fc2"
    res = {}
    raw = {}
    for d in (fetch_dir, write_dir):
        for (kn, cn), vals in load(d).items():
            for pat, key in KEY.items():
                if pat in kn:
                    vals = sorted(vals)
                    raw.setdefault(key, {})[cn] = vals[len(vals) // 2]   # median launch
    for key, c in raw.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            res[key] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    def merge(path, val):   # one file, one entry per workload
        try:
            cur = json.load(open(path))
        except Exception:
            cur = {}
        cur[workload] = val
        json.dump(cur, open(path, "w"), indent=1)
    merge(out, res)
    merge(out.replace(".json", "_raw.json"), raw)
    print(json.dumps({"traffic_bytes_per_launch": res, "raw_KiB": raw}, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:5])
