#!/bin/bash
# Side builds of msda_tiled9.hip (one small shared library each) -> visionllm_amd/_build_abl/libmsda9_<name>.so
# Usage: tools/msda9_variants.sh name=flags ...      e.g.  base=  prio0=-DT9_GPRIO=0  nofma=-DT9_ABL=1
set -e
cd "$(dirname "$0")/.."
mkdir -p visionllm_amd/_build_abl
rm -f visionllm_amd/_build_abl/libmsda9_*.so
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude -Ivisionllm_amd/csrc -fno-slp-vectorize $flags -DT9_ABL_ENTRY \
      -o visionllm_amd/_build_abl/libmsda9_$name.so visionllm_amd/csrc/msda_tiled9.hip 2>&1 | grep -E "error|spill" || true ) &
done
wait
ls visionllm_amd/_build_abl/ | grep msda9
