cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for wpc in 4 8; do
for dbg in 0 1 2 4 6 7 3; do
echo "== stream wpc $wpc dbg $dbg"; RMCV_K1_WPC=$wpc RMCV_KS_DBG=$dbg python tools/k1_bench.py 2
done; done
} > gpurun_out/abk1/out_d.txt 2>&1
grep -E "^==|k_binary image|rror|fault" gpurun_out/abk1/out_d.txt
