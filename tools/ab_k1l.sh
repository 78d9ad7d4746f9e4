# dev tool: same-box A/B, many alternating repetitions (box clocks drift within a run)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2 3 4; do
for g in 3; do
echo "== base groups $g"; python tools/k1_bench.py $g
for v in ${VARS:-x3nt x3nte x3ntu3}; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
done
done
} > gpurun_out/abk1/out_l.txt 2>&1
grep -E "^==|k_binary image|rror|fault|passed|failed" gpurun_out/abk1/out_l.txt
