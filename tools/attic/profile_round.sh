# usage: bash tools/profile_round.sh <tag>   (run on the GPU box through gpurun; writes gpurun_out/prof_<tag>/)
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
# (1) kernel stats of the roofline command: serial steps, so every k_binary launch runs alone on the GPU
rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -- python3 bench.py --steps 20 --warmup 3 --streams 1 --cpu-frames 0 > $out/bench_serial.json 2> $out/serial.err
# (2) kernel stats of the default (double-buffered) command
rocprofv3 --kernel-trace --stats --output-format csv -d $out/default -- python3 bench.py --steps 20 --warmup 3 --cpu-frames 0 > $out/bench_default.json 2> $out/default.err
# (3) HBM traffic counters, one pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass), no tracing domains besides kernel-trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --streams 1 --cpu-frames 0 --no-extras > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --streams 1 --cpu-frames 0 --no-extras > /dev/null 2> $out/pmc_write.err
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
def avg_counter(d, name):
    f = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_binary" in r["Kernel_Name"] and r["Counter_Name"] == name]
    return sum(v) / len(v), len(v)
fetch, n1 = avg_counter("pmc_fetch", "FETCH_SIZE")
write, n2 = avg_counter("pmc_write", "WRITE_SIZE")
# MI355X_MICROARCH.md (HBM): counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane)
# coalesced streaming read -> doubled; WRITE_SIZE is exact for 16 B/lane streaming stores
rec = {"kernel": "k_binary", "frames": 256, "width": 1280, "height": 1024, "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB_raw": write,
       "launches_averaged": [n1, n2], "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 wide-load under-count), write bytes = WRITE_SIZE x 1024",
       "hbm_bytes_per_launch": int((2 * fetch + write) * 1024)}
json.dump(rec, open(out + "/k_binary_traffic.json", "w"), indent=1)
print(rec)
for d in ("serial", "default"):
    f = glob.glob(out + "/" + d + "/**/*kernel_stats.csv", recursive=True)[0]
    print(d, [ (r["Name"][:30], r["Calls"], r["AverageNs"]) for r in csv.DictReader(open(f)) if "rmcv" in r["Name"]][:7])
PY
tail -1 $out/bench_serial.json | cut -c1-400; tail -1 $out/bench_default.json | cut -c1-400
