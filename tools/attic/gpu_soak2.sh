# more soak on a fresh box: parity over further seed blocks + long fuzz with new seeds, in parallel where the oracle is the bottleneck
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/soak2
timeout -k 10 900 python tools/soak_parity.py 16 2000000 > gpurun_out/soak2/parity.txt 2>&1 &
timeout -k 10 900 python tools/fuzz_path.py 300 201 0 > gpurun_out/soak2/path0.txt 2>&1 &
timeout -k 10 900 python tools/fuzz_path.py 200 202 1 > gpurun_out/soak2/path1.txt 2>&1 &
timeout -k 10 900 python tools/fuzz_contours.py 12000 203 > gpurun_out/soak2/contours.txt 2>&1 &
timeout -k 10 900 python tools/fuzz_legacy.py 2500 204 > gpurun_out/soak2/legacy.txt 2>&1 &
while [ -n "$(jobs -r)" ]; do sleep 30; echo "still running: $(jobs -r | wc -l)"; done
wait
for f in gpurun_out/soak2/*.txt; do echo "== $f"; grep -v amdgpu.ids $f | tail -2; done
