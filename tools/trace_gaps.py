"""dev tool: from a rocprofv3 --kernel-trace CSV, the pixel kernels' timeline: duration, start-to-start pitch, how long two of them
overlap, and the sparse / compaction kernels' durations over the same window.   python tools/trace_gaps.py <kernel_trace.csv> [last N]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ks = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
ks.sort(key=lambda k: k[1])
pix = [k for k in ks if "k_binary" in k[0]][-last:]
t0 = pix[0][1]
print("pixel kernel: %s" % pix[0][0][:60])
prev = None
durs, pitches, ovl = [], [], []
for name, s, e in pix:
    d = (e - s) / 1e3
    durs.append(d)
    if prev:
        pitches.append((s - prev[1]) / 1e3)
        ovl.append((prev[2] - s) / 1e3)
    prev = (name, s, e)
import statistics as st
print("n %d  duration us: median %.1f min %.1f max %.1f | start-to-start pitch: median %.1f | overlap with the previous one: median %.1f (negative = gap)" %
      (len(pix), st.median(durs), min(durs), max(durs), st.median(pitches), st.median(ovl)))
print("end-to-end pitch (median of end[i] - end[i-1]): %.1f us" % st.median([(pix[i][2] - pix[i - 1][2]) / 1e3 for i in range(1, len(pix))]))
w0, w1 = pix[0][1], pix[-1][2]
for pat in ("k_contours_w4", "k_contours_w8", "k_compact"):
    d = [(e - s) / 1e3 for n, s, e in ks if pat in n and s >= w0 and e <= w1]
    if d:
        print("%-16s n %d  median %.1f us  min %.1f  max %.1f" % (pat, len(d), st.median(d), min(d), max(d)))
if len(sys.argv) > 3:
    for name, s, e in pix[:16]:
        print("%8.1f .. %8.1f  (%.1f)" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
