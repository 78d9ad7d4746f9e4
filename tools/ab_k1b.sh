cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for g in 2 4; do
for m in 0 1 2; do echo "== base groups $g morph $m"; python tools/k1_bench.py $g $m; done
for v in sr16 sr64 nold nost; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
done
echo "== base groups 6"; python tools/k1_bench.py 6
echo "== base groups 4 1920x1200"; python tools/k1_bench.py 4 2 1920 1200
} > gpurun_out/abk1/out_b.txt 2>&1
grep -E "^==|k_binary" gpurun_out/abk1/out_b.txt
