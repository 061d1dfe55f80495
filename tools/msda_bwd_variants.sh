#!/bin/bash
# Builds msda_bwd_mfma.hip variants: each argument is "name:-Dflag -Dflag ..." -> visionllm_amd/_build_abl/libbwdv_<name>.so
cd "$(dirname "$0")/.."
mkdir -p visionllm_amd/_build_abl
rm -f visionllm_amd/_build_abl/libbwdv_*.so
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -DBT_ABL_ENTRY $flags \
      -o visionllm_amd/_build_abl/libbwdv_$name.so visionllm_amd/csrc/msda_bwd_mfma.hip 2>&1 | grep -E "error" || true ) &
done
wait
ls visionllm_amd/_build_abl/ | grep bwdv
