# dev tool: the GPU's shader clock and socket power while the pipelined bench steps, and while only 1 GiB copies run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/clk
one() { rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo; }
python3 bench.py --steps 4000 --warmup 5 --cpu-frames 0 --no-extras --repeats 8 > gpurun_out/clk/b.json 2> gpurun_out/clk/b.err &
BP=$!
echo "== while bench.py starts, generates frames, then steps (one sample per second; idle = 94 MHz)"
for i in $(seq 1 45); do if ! kill -0 $BP 2>/dev/null; then break; fi; one; sleep 1; done | uniq -c
wait $BP
python3 - <<'PY' &
import torch, time
a = torch.empty(1 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(50): b.copy_(a, non_blocking=True)
    torch.cuda.synchronize()
PY
CP=$!
sleep 4
echo "== copies only"; for i in 1 2 3; do one; sleep 0.5; done
wait $CP
python3 -c "
import json
j=json.loads(open('gpurun_out/clk/b.json').read().strip().splitlines()[-1]); print('bench during the probe (4000-step regions):', j['value'], j['ms_per_step'])"
