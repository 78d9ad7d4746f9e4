cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
for g in 2 3; do
echo "== base(U4) groups $g"; python tools/k1_bench.py $g
for v in u5 u6 u8; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
done
done
} > gpurun_out/abk1/out_n.txt 2>&1
grep -E "^==|k_binary image|rror|fault|passed|failed" gpurun_out/abk1/out_n.txt
