"""dev tool: from a rocprofv3 --kernel-trace CSV of tools/pipeline_bench.c (regions of `steps` batches between drains), the pixel kernels' timeline
of the fastest and of the slowest region: start, end, duration, and which sparse kernels ran beside them.
    python tools/trace_regions.py <kernel_trace.csv> [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ks = sorted(((r["Kernel_Name"].split("(")[0].replace("void rmcv::", "").replace("rmcv::", "")[:16], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows), key=lambda k: k[1])
pix = [k for k in ks if k[0].startswith("k_binary")]
# regions: a gap of > 40 us between the end of everything and the next pixel start
regs, cur = [], [pix[0]]
for a, b in zip(pix, pix[1:]):
    busy_end = max(e for n, s, e, q in ks if s < b[1])
    if b[1] - busy_end > 20000:
        regs.append(cur)
        cur = []
    cur.append(b)
regs.append(cur)
regs = [r for r in regs if len(r) == steps]
def span(r):
    last_end = max(e for n, s, e, q in ks if s >= r[0][1] and s <= r[-1][2] + 400000 and e <= r[-1][2] + 1500000 and (n.startswith("k_compact") or n.startswith("k_binary")))
    return (last_end - r[0][1]) / 1e3
regs.sort(key=span)
print("%d regions of %d steps; us per step: fastest %.1f, median %.1f, slowest %.1f" % (len(regs), steps, span(regs[0]) / steps, span(regs[len(regs) // 2]) / steps, span(regs[-1]) / steps))
for label, r in (("FASTEST", regs[0]), ("SLOWEST", regs[-1])):
    t0 = r[0][1]
    print("\n== %s region: %.1f us per step" % (label, span(r) / steps))
    w1 = r[-1][2] + 600000
    for n, s, e, q in ks:
        if s >= t0 - 1000 and s <= w1 and (n.startswith("k_binary") or n.startswith("k_contours") or n.startswith("k_compact")):
            print("   %-16s q%-3s %8.1f .. %8.1f  (%6.1f us)" % (n, q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
