"""dev tool: k_binary alone and COLD (4 contexts on 4 frame sets in turn), by workgroups per CU, for the build RMCV_LIB_PATH names --
the attribution table of profiles/r04b_k_binary_ablation.txt (loads compiled out / stores compiled out / both: tools/build_variant.sh
with -DRMCV_K1_NOLOAD, -DRMCV_K1_NOSTORE).  One stream; `two` = the launches alternating over two streams (what the pipeline does)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from rmcv_amd import CAMP_BLUE, OPT_PIXEL_GROUPS, STAGE_BINARY, Context, default_params, synth  # noqa: E402

n, W, H = 256, 1280, 1024
torch.cuda.init()
dev = torch.device("cuda", 0)
sets = [torch.from_numpy(synth.batch(k * 1000003, n, W, H, CAMP_BLUE, 0, threads=16)).to(dev) for k in range(4)]
ctxs = []
for k in range(4):
    c = Context(device=0, max_frames=n, max_width=W, max_height=H)
    c.bind_device_frames(sets[k].data_ptr(), n, H, W, keepalive=sets[k])
    ctxs.append(c)
p = default_params()
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
name = os.path.basename(os.environ.get("RMCV_LIB_PATH", "librmcv_hip.so"))
for groups in (1, 2, 3, 4):
    for c in ctxs:
        c.set_option(OPT_PIXEL_GROUPS, groups)
    out = []
    for nstreams in (1, 2):
        for i in range(8):
            ctxs[i % 4].run(p, STAGE_BINARY, ss[i % nstreams].cuda_stream)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            import time
            t0 = time.perf_counter()
            for i in range(40):
                ctxs[i % 4].run(p, STAGE_BINARY, ss[i % nstreams].cuda_stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 40 * 1e3)
        ts.sort()
        out.append(ts[2])
    print("%-22s groups/CU %d: one stream %.4f ms (%.0f GB/s)   two streams %.4f ms (%.0f GB/s)" %
          (name, groups, out[0], n * W * H * 4 / out[0] / 1e6, out[1], n * W * H * 4 / out[1] / 1e6), flush=True)
