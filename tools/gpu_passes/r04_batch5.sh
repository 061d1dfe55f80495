set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout 900 python tools/gemm_tile_order_ab.py 0,2,4,8,16 > $O/gemm_tile_order.txt 2>&1; grep -v amdgpu.ids $O/gemm_tile_order.txt
