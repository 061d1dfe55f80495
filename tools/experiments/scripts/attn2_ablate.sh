#!/bin/bash
# Timing-only ablation builds of attn2.hip (one small shared library per mask) -> visionllm_amd/_build_abl/ (git-ignored,
# travels to the GPU box).  Usage: tools/attn2_ablate.sh [mask ...]   (default: 0 1 2 4 8 12 16 32 64)
set -e
cd "$(dirname "$0")/.."
mkdir -p visionllm_amd/_build_abl
MASKS="${@:-0 1 2 4 8 12 16 32 64}"
for m in $MASKS; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -DATTN2_ABL=$m -DATTN2_ABL_ENTRY \
      -o visionllm_amd/_build_abl/libattn2_abl$m.so visionllm_amd/csrc/attn2.hip 2>&1 | grep -v "warning\|^ *[0-9]* |\|^ *|\|generated" || true ) &
done
wait
ls -la visionllm_amd/_build_abl/
