# dev tool: cache-policy bits of the coalesced pixel kernel's loads and image stores (aux: 1 = sc0, 2 = nt, 3 = sc0 + nt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
echo "== tree (loads nt, stores nt) groups 3"; python tools/k1_bench.py 3
for v in st0 st1 st3 ld3 ld1; do echo "== $v groups 3"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py 3; done
done
} > gpurun_out/abk1/out_q.txt 2>&1
grep -E "^==|k_binary image|rror|fault" gpurun_out/abk1/out_q.txt
