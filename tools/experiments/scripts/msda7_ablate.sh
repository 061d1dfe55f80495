#!/bin/bash
# Timing-only ablation builds of msda_tiled7.hip (one small shared library per mask) -> visionllm_amd/_build_abl/
# Usage: tools/msda7_ablate.sh [mask ...]   (default: 0 1 2 3 4 8 16 20 28 64)
set -e
cd "$(dirname "$0")/.."
mkdir -p visionllm_amd/_build_abl
MASKS="${@:-0 1 2 3 4 8 16 20 28 64}"
for m in $MASKS; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -fno-slp-vectorize -DT7_ABL=$((m % 1024)) -DT7_OLD_PLACEMENT=$((m / 1024)) -DT7_ABL_ENTRY \
      -o visionllm_amd/_build_abl/libmsda7_abl$m.so visionllm_amd/csrc/msda_tiled7.hip 2>&1 | grep -E "error|spill" || true ) &
done
wait
ls visionllm_amd/_build_abl/ | grep msda7
