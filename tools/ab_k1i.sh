cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
echo "== base"; python tools/k1_bench.py 2
for v in st0 st1 st3 ld1 ld5; do echo "== $v"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py 2; done
done
} > gpurun_out/abk1/out_i.txt 2>&1
grep -E "^==|k_binary image|rror|fault" gpurun_out/abk1/out_i.txt
