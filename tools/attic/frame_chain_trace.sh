#!/bin/bash
# dev tool: device-side timeline of the per-frame chain (kernel + memory-copy trace of tools/frame_chain.py), the last chains printed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/fc_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -- python3 tools/frame_chain.py > $out/run.txt 2> $out/err.txt
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
ev = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-30:]))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", "?"))[-24:]))
ev.sort()
# the last 3 chains of the first mode (pageable): find k_binary launches
kb = [i for i, e in enumerate(ev) if "k_binary" in e[2]]
for i0 in kb[150:153]:
    # back up to the copy before it
    j = i0 - 1
    t0 = ev[j][0]
    print("---- chain")
    k = j
    while k < len(ev) and (k == j or "k_binary" not in ev[k][2] or k == i0):
        s, e, n = ev[k]
        print("  %8.1f .. %8.1f  (%6.1f us)  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
        k += 1
        if k - j > 14: break
PY
