cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do for g in 2 3; do
echo "== prev groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_prev.so python tools/k1_bench.py $g
echo "== new groups $g"; python tools/k1_bench.py $g
done; done
echo "== prev 1920 groups 2"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_prev.so python tools/k1_bench.py 2 2 1920 1200
echo "== new 1920 groups 2"; python tools/k1_bench.py 2 2 1920 1200
} > gpurun_out/abk1/out_g.txt 2>&1
grep -E "^==|k_binary|rror|fault" gpurun_out/abk1/out_g.txt
