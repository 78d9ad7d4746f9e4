# dev tool: the step time of pipelines created one after the other in ONE process (is the second one slower?  does closing the first help?)
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rmcv_amd import CAMP_BLUE, STAGE_ALL, Pipeline, default_params, synth
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
n, W, H = 256, 1280, 1024
sets = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).to(dev) for k in range(8)]
p = default_params()
i = [0]
def region(pl, k=200):
    for _ in range(16): pl.submit(sets[i[0] % 8].data_ptr(), n, H, W, p, STAGE_ALL); i[0] += 1
    pl.drain()
    t0 = time.perf_counter()
    for _ in range(k): pl.submit(sets[i[0] % 8].data_ptr(), n, H, W, p, STAGE_ALL); i[0] += 1
    pl.drain()
    return (time.perf_counter() - t0) / k * 1e3
mk = lambda: Pipeline(device=0, depth=8, armour_cap=n * 8, max_frames=n, max_width=W, max_height=H)
A = mk()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.8: region(A, 100)
print("A alone %.4f %.4f" % (region(A), region(A)), flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "keep"
if mode == "close":
    A.close()
    B = mk()
    print("A closed, B: %.4f %.4f" % (region(B), region(B)), flush=True)
    B.close()
    C_ = mk()
    print("B closed, C: %.4f %.4f" % (region(C_), region(C_)), flush=True)
else:
    B = mk()
    print("A kept, B: %.4f A: %.4f B: %.4f A: %.4f" % (region(B), region(A), region(B), region(A)), flush=True)
    C_ = mk()
    print("A, B kept, C: %.4f A: %.4f B %.4f" % (region(C_), region(A), region(B)), flush=True)
# ... and the pixel kernel ALONE on the contexts of each pipeline (one stream, no queue sharing in play): is it the memory placement?
from rmcv_amd import STAGE_BINARY
def kb_alone(pl, reps=40):
    s = torch.cuda.Stream(device=dev)
    for k, c in enumerate(pl.contexts):
        c.bind_device_frames(sets[k].data_ptr(), n, H, W)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        pl.contexts[0].run(p, STAGE_BINARY, s.cuda_stream)
        e0.record(s)
        for r in range(reps):
            pl.contexts[(r + 1) % 8].run(p, STAGE_BINARY, s.cuda_stream)
        e1.record(s)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
if mode != "close":
    for name, q in (("A", A), ("B", B), ("C", C_), ("A", A), ("B", B), ("C", C_)):
        print("k_binary alone on %s's contexts: %.4f" % (name, kb_alone(q)), flush=True)
