#!/bin/bash
# dev tool: the same-box A/B experiments of rounds 1-2, one script instead of one file per experiment.
#   bash tools/ab_k1_all.sh <experiment>      (through gpurun; results under gpurun_out/)
# Each experiment keeps the comment that said what it was for.  They are records of measurements that were taken
# (DESIGN.md 6c cites them by name); variant libraries come from tools/build_variant.sh.
exp="$1"; [ -n "$exp" ] || { echo "usage: $0 <experiment>; experiments: a b c d e f g h i j k l m n o p q"; exit 2; }
case "$exp" in
a)
# dev tool: same-box A/B of k_binary builds/knobs with tools/k1_bench.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
for g in 2 4; do
echo "== base groups $g"; python tools/k1_bench.py $g
echo "== LOADV=1 (coalesced through LDS) groups $g"; RMCV_K1_LOADV=1 python tools/k1_bench.py $g
echo "== nt loads, strided, groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so python tools/k1_bench.py $g
echo "== nt loads + LOADV=1 groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so RMCV_K1_LOADV=1 python tools/k1_bench.py $g
done
done
echo "== nt loads + LOADV=1 groups 1"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so RMCV_K1_LOADV=1 python tools/k1_bench.py 1
echo "== nt loads + LOADV=1 groups 3"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so RMCV_K1_LOADV=1 python tools/k1_bench.py 3
echo "== nt loads + LOADV=1 groups 6"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_ntl.so RMCV_K1_LOADV=1 python tools/k1_bench.py 6
} > gpurun_out/abk1/out.txt 2>&1
grep -E "^==|k_binary|checksum" gpurun_out/abk1/out.txt
;;
b)
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
;;
c)
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
;;
d)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for wpc in 4 8; do
for dbg in 0 1 2 4 6 7 3; do
echo "== stream wpc $wpc dbg $dbg"; RMCV_K1_WPC=$wpc RMCV_KS_DBG=$dbg python tools/k1_bench.py 2
done; done
} > gpurun_out/abk1/out_d.txt 2>&1
grep -E "^==|k_binary image|rror|fault" gpurun_out/abk1/out_d.txt
;;
e)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do for g in 2 3 4 6; do
echo "== new groups $g"; python tools/k1_bench.py $g
done; done
echo "== new groups 4 1920x1200"; python tools/k1_bench.py 4 2 1920 1200
echo "== new groups 2 1920x1200"; python tools/k1_bench.py 2 2 1920 1200
echo "== new groups 4 morph 1"; python tools/k1_bench.py 4 1
echo "== new groups 4 morph 0"; python tools/k1_bench.py 4 0
} > gpurun_out/abk1/out_e.txt 2>&1
grep -E "^==|k_binary|rror|fault" gpurun_out/abk1/out_e.txt
;;
f)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for g in 2 3; do
echo "== base groups $g"; python tools/k1_bench.py $g
for v in u2 u6 sr64 ldnt ldsc1; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
echo "== base notaper groups $g"; RMCV_K1_TAPER=0 python tools/k1_bench.py $g
done
} > gpurun_out/abk1/out_f.txt 2>&1
grep -E "^==|k_binary|rror|fault" gpurun_out/abk1/out_f.txt
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/abk1/bench_f.json 2> gpurun_out/abk1/bench_f.err
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/abk1/bench_f.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["timed_region"]["ms_per_step_each"], j["roofline"], j["lone_batch_ms"], j["stage_ms"], j["c2_binary_only"], j["detect_only_no_image"])
PY
;;
g)
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
;;
h)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do for g in 2 3 4; do
echo "== per-lane groups $g"; RMCV_K1_COAL=0 python tools/k1_bench.py $g
echo "== coalesced groups $g"; python tools/k1_bench.py $g
done; done
echo "== per-lane 1920 groups 2"; RMCV_K1_COAL=0 python tools/k1_bench.py 2 2 1920 1200
echo "== coalesced 1920 groups 2"; python tools/k1_bench.py 2 2 1920 1200
echo "== coalesced morph 1"; python tools/k1_bench.py 2 1
} > gpurun_out/abk1/out_h.txt 2>&1
grep -E "^==|k_binary|rror|fault" gpurun_out/abk1/out_h.txt
;;
i)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
echo "== base"; python tools/k1_bench.py 2
for v in st0 st1 st3 ld1 ld5; do echo "== $v"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py 2; done
done
} > gpurun_out/abk1/out_i.txt 2>&1
grep -E "^==|k_binary image|rror|fault" gpurun_out/abk1/out_i.txt
;;
j)
# dev tool: same-box A/B of the double-buffered phase 1 of k_binary (var_pipe<U>.so) against the build in the tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
for g in 2 3; do
echo "== base groups $g"; python tools/k1_bench.py $g
for v in pipe2 pipe3 pipe4; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
done
done
echo "== parity pipe2"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_pipe2.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "binary or full_size or c2 or c5 or padding or geometry" 2>&1 | tail -3
} > gpurun_out/abk1/out_j.txt 2>&1
grep -E "^==|k_binary image|k_binary no-image|rror|fault|passed|failed" gpurun_out/abk1/out_j.txt
;;
k)
# dev tool: same-box A/B of the wave-coalesced dwordx3 phase 1 of k_binary (var_x3*.so) against the build in the tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
echo "== parity x3nt"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_x3nt.so timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "binary or full_size or c2 or c5 or padding or geometry" 2>&1 | tail -5
for rep in 1 2; do
for g in 2 3 4; do
echo "== base groups $g"; python tools/k1_bench.py $g
for v in ${VARS:-x3nt x3 x3ntu2}; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
done
done
echo "== base 1920"; python tools/k1_bench.py 2 2 1920 1200
echo "== x3nt 1920"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_x3nt.so python tools/k1_bench.py 2 2 1920 1200
} > gpurun_out/abk1/out_k.txt 2>&1
grep -E "^==|k_binary image|k_binary no-image|rror|fault|passed|failed" gpurun_out/abk1/out_k.txt
;;
l)
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
;;
m)
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
;;
n)
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
;;
o)
# dev tool: k_binary in the tree, alone (groups 2,3,4) + pixel-kernel parity tests + the driver command twice
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
echo "== parity"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "binary or full_size or c2 or c5 or padding or geometry" 2>&1 | tail -5
for rep in 1 2; do
for g in 2 3 4; do echo "== tree groups $g"; python tools/k1_bench.py $g; done
done
echo "== tree 1920 groups 3"; python tools/k1_bench.py 3 2 1920 1200
for rep in 1 2; do
echo "== bench driver command"; python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abk1/b.json 2>gpurun_out/abk1/b.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abk1/b.json').read().strip().splitlines()[-1]); print('   ', j['steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'k1', j['roofline']['avg_launch_ms'])"
done
} > gpurun_out/abk1/out_o.txt 2>&1
grep -E "^==|k_binary image|k_binary no-image|rror|fault|passed|failed|^    " gpurun_out/abk1/out_o.txt
;;
p)
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
;;
q)
# dev tool: cache-policy bits of the coalesced pixel kernel's loads and image stores (aux: 1 = sc0, 2 = nt, 3 = sc0 + nt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for rep in 1 2; do
echo "== tree (loads nt, stores nt) groups 3"; python tools/k1_bench.py 3
for v in st0 st1 st3 ld3 ld1; do echo "== $v groups 3"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py 3; done
done
} > gpurun_out/abk1/out_q.txt 2>&1
grep -E "^==|k_binary image|rror|fault" gpurun_out/abk1/out_q.txt
;;
*) echo "unknown experiment $exp"; exit 2;;
esac
