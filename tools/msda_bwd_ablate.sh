#!/bin/bash
# Timing-only ablation builds of msda_bwd_tiled.hip -> visionllm_amd/_build_abl/libmsdabwd_abl<mask>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p visionllm_amd/_build_abl
MASKS="${@:-0 1 2 3 4 7 8 15 31}"
for m in $MASKS; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -DBT_ABL=$m -DBT_ABL_ENTRY \
      -o visionllm_amd/_build_abl/libmsdabwd_abl$m.so tools/experiments/msda_bwd_tiled.hip 2>&1 | grep -E "error" || true ) &
done
wait
ls visionllm_amd/_build_abl/ | grep msdabwd
