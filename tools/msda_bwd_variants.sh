#!/bin/bash
# Builds msda_bwd_mfma.hip variants: each argument is "name:-Dflag -Dflag ..." -> visionllm_amd/_build_abl/libbwdv_<name>.so
# (the special name 0_head builds the committed kernel, git show HEAD:..., as the reference)
cd "$(dirname "$0")/.."
mkdir -p visionllm_amd/_build_abl
rm -f visionllm_amd/_build_abl/libbwdv_*.so
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  src=visionllm_amd/csrc/msda_bwd_mfma.hip
  if [ "$name" = "0_head" ]; then git show HEAD:$src > visionllm_amd/csrc/_head_msda_bwd_mfma.hip; src=visionllm_amd/csrc/_head_msda_bwd_mfma.hip; fi
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -DBT_ABL_ENTRY $flags \
      -o visionllm_amd/_build_abl/libbwdv_$name.so $src 2>&1 | grep -E "error" || true ) &
done
wait
rm -f visionllm_amd/csrc/_head_msda_bwd_mfma.hip
ls visionllm_amd/_build_abl/ | grep bwdv
