# dev tool: from a rocprofv3 kernel trace of bench.py, how busy is the GPU with k_binary in the timed region?
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if "rmcv" in r["Kernel_Name"]]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1][:12], r["Queue_Id"]) for r in rows]
ev.sort()
# timed region = the longest run of k_binary<0,2> launches whose starts are < 1 ms apart (the steps of the timed loop)
kb = [e for e in ev if e[2].startswith("k_binary")]
lo, hi = kb[3][0], kb[3 + 30][0] if len(kb) > 40 else kb[-1][0]
sel = [e for e in ev if lo <= e[0] < hi]
span = hi - lo
def union(iv):
    iv = sorted(iv); tot = 0; cur_s, cur_e = iv[0]
    for s, e in iv[1:]:
        if s > cur_e: tot += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    return tot + cur_e - cur_s
kbs = [(s, min(e, hi)) for s, e, n, q in sel if n.startswith("k_binary")]
allk = [(s, min(e, hi)) for s, e, n, q in sel]
print("steps in window", len(kbs), "span per step %.1f us" % (span / len(kbs) / 1e3))
print("k_binary busy %.3f  any kernel busy %.3f" % (union(kbs) / span, union(allk) / span))
two = 0
pts = sorted([(s, 1) for s, e in kbs] + [(e, -1) for s, e in kbs]); c = 0; last = pts[0][0]
for t, d in pts:
    if c >= 2: two += t - last
    c += d; last = t
print("two or more k_binary at once %.3f" % (two / span))
for name in ("k_binary", "k_contours", "k_fit", "k_pairs", "k_compact"):
    d = [e - s for s, e, n, q in sel if n.startswith(name)]
    if d: print(name.ljust(12), "avg %.1f us" % (sum(d) / len(d) / 1e3), "queues", sorted({q for s, e, n, q in sel if n.startswith(name)}))
