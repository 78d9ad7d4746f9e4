cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
echo "== parity sr64"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_sr64.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "binary or full_size or c2 or padding or geometr or coalesced" 2>&1 | tail -3
for rep in 1 2; do
for g in 2 3; do
echo "== tree (32 rows) groups $g"; python tools/k1_bench.py $g
for v in sr64 sr16; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
done
done
} > gpurun_out/abk1/out_p.txt 2>&1
grep -E "^==|k_binary image|rror|fault|passed|failed" gpurun_out/abk1/out_p.txt
