# dev tool: same-box A/B of the double-buffered phase 1 of k_binary (var_pipe<U>.so) against the build in the tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
for g in 2 3; do
echo "== base groups $g"; python tools/k1_bench.py $g
for v in pipe2 pipe3 pipe4; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
done
done
echo "== parity pipe2"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_pipe2.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "binary or full_size or c2 or c5 or padding or geometry" 2>&1 | tail -3
} > gpurun_out/abk1/out_j.txt 2>&1
grep -E "^==|k_binary image|k_binary no-image|rror|fault|passed|failed" gpurun_out/abk1/out_j.txt
