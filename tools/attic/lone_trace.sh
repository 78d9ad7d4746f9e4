# dev tool: kernel timeline of lone batches with and without the frame-level hand-over (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/lone_trace; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 tools/lone_bench.py 0 one > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rmcv" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 12 kernels of each configuration: print relative to the k_binary before them
last = None
out = []
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void rmcv::", "")[:22]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if "k_binary" in name:
        last = s
    out.append((name, (s - last) / 1e3 if last else 0, (e - last) / 1e3 if last else 0))
for name, s, e in out[-24:]:
    print("%-24s start %+9.1f us  end %+9.1f us" % (name, s, e))
PY
