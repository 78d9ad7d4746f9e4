# dev tool: same-box A/B of the wave-coalesced dwordx3 phase 1 of k_binary (var_x3*.so) against the build in the tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
echo "== parity x3nt"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_x3nt.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "binary or full_size or c2 or c5 or padding or geometry" 2>&1 | tail -5
for rep in 1 2; do
for g in 2 3 4; do
echo "== base groups $g"; python tools/k1_bench.py $g
for v in ${VARS:-x3nt x3 x3ntu2}; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
done
done
echo "== base 1920"; python tools/k1_bench.py 2 2 1920 1200
echo "== x3nt 1920"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_x3nt.so python tools/k1_bench.py 2 2 1920 1200
} > gpurun_out/abk1/out_k.txt 2>&1
grep -E "^==|k_binary image|k_binary no-image|rror|fault|passed|failed" gpurun_out/abk1/out_k.txt
