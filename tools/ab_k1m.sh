cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
for g in 2 3 4; do
echo "== base groups $g"; python tools/k1_bench.py $g
for v in x3nte x3nteu3 x3nteu2; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
done
done
for g in 2 3; do
echo "== base 1920 groups $g"; python tools/k1_bench.py $g 2 1920 1200
echo "== x3nte 1920 groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_x3nte.so python tools/k1_bench.py $g 2 1920 1200
done
for m in 0 1; do
echo "== base morph $m"; python tools/k1_bench.py 3 $m
echo "== x3nte morph $m"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_x3nte.so python tools/k1_bench.py 3 $m
done
} > gpurun_out/abk1/out_m.txt 2>&1
grep -E "^==|k_binary image|k_binary no-image|rror|fault|passed|failed" gpurun_out/abk1/out_m.txt
