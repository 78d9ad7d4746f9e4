cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
echo "== v1 (k_binary) groups 2"; RMCV_K1_V1=1 python tools/k1_bench.py 2
echo "== v1 (k_binary) groups 4"; RMCV_K1_V1=1 python tools/k1_bench.py 4
for wpc in 4 8 12; do for band in 32 64 128; do
echo "== stream wpc $wpc band $band"; RMCV_K1_WPC=$wpc RMCV_K1_BAND=$band python tools/k1_bench.py 2
done; done
echo "== stream default groups 2 1920x1200"; python tools/k1_bench.py 2 2 1920 1200
echo "== v1 groups 4 1920x1200"; RMCV_K1_V1=1 python tools/k1_bench.py 4 2 1920 1200
echo "== stream morph 1"; python tools/k1_bench.py 2 1
echo "== stream morph 0"; python tools/k1_bench.py 2 0
} > gpurun_out/abk1/out_c.txt 2>&1
grep -E "^==|k_binary|checksum|rror" gpurun_out/abk1/out_c.txt
