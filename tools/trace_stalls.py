"""dev tool: from a rocprofv3 --kernel-trace CSV of a pipelined run, the places where the pixel stream STALLED: the largest
end-to-end pitches between consecutive pixel kernels, and what ran around them.   python tools/trace_stalls.py <kernel_trace.csv> [top N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ks = sorted(((r["Kernel_Name"].split("(")[0].replace("void rmcv::", "").replace("rmcv::", "")[:28], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", "?"), r.get("Queue_Id", "?")) for r in rows), key=lambda k: k[1])
pix = [k for k in ks if k[0].startswith("k_binary")]
t0 = pix[0][1]
import statistics as st
pitch = [(pix[i][2] - pix[i - 1][2]) / 1e3 for i in range(1, len(pix))]
med = st.median(pitch)
print("pixel kernels %d, end-to-end pitch median %.1f us" % (len(pix), med))
cand = sorted(range(1, len(pix)), key=lambda i: -pitch[i - 1])
shown = 0
for i in cand:
    p = pitch[i - 1]
    gap = (pix[i][1] - pix[i - 1][2]) / 1e3
    if gap > 150:          # a region boundary (drain + host): not a stall
        continue
    print("\n== pitch %.1f us (x%.2f) before pixel kernel #%d at %.3f ms; it started %.1f us %s the previous one ended" %
          (p, p / med, i, (pix[i][1] - t0) / 1e6, abs(gap), "after" if gap > 0 else "before"))
    w0, w1 = pix[max(0, i - 3)][1], pix[min(len(pix) - 1, i + 1)][2]
    for name, s, e, sid, qid in ks:
        if e >= w0 and s <= w1:
            print("   %-28s q%-3s %9.1f .. %9.1f  (%7.1f us)" % (name, qid, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
    shown += 1
    if shown >= top:
        break
