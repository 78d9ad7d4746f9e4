# usage: bash tools/profile_round2.sh <tag>   (on the GPU box through gpurun; writes gpurun_out/prof_<tag>/)
# Round-2 evidence: kernel stats (serial + default command), HBM traffic of k_binary, and SQ / TCC counters of the pixel and the
# sparse kernel.  Every rocprofv3 call has the program directly after `--`; PMC passes carry --kernel-trace only.
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag; mkdir -p $out
Q="--cpu-frames 0 --no-extras --repeats 1 --warmup-seconds 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/serial -- python3 bench.py --steps 20 --warmup 3 --streams 1 --cpu-frames 0 --no-extras > $out/bench_serial.json 2> $out/serial.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/default -- python3 bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras > $out/bench_default.json 2> $out/default.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --streams 1 $Q > /dev/null 2> $out/pmc_fetch.err
# calibration of FETCH_SIZE for THIS kernel's loads (12 B per lane, wave-coalesced; the guide calibrates 16 B per lane only): with
# morph = none no row is read twice, so every launch reads exactly 256 x 1280 x 1024 x 3 bytes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_cal -- python3 tools/k1_bench.py 3 0 > $out/cal.txt 2> $out/pmc_cal.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 bench.py --steps 3 --warmup 1 --streams 1 $Q > /dev/null 2> $out/pmc_write.err
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU --output-format csv -d $out/pmc_sq1 -- python3 bench.py --steps 3 --warmup 1 $Q > /dev/null 2> $out/pmc_sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $out/pmc_sq2 -- python3 bench.py --steps 3 --warmup 1 $Q > /dev/null 2> $out/pmc_sq2.err
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_tcc -- python3 bench.py --steps 3 --warmup 1 $Q > /dev/null 2> $out/pmc_tcc.err
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
def table(d):
    fs = glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        return {}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
res = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_tcc"):
    for k, cs in table(d).items():
        if "rmcv" in k:
            res.setdefault(k, {}).update(cs)
json.dump(res, open(out + "/counters_per_kernel.json", "w"), indent=1, sort_keys=True)
kb = [v for k, v in res.items() if "k_binary" in k]
cal = [v for k, v in table("pmc_cal").items() if "k_binary" in k]
if kb and "FETCH_SIZE" in kb[0] and "WRITE_SIZE" in kb[0]:
    known = 256 * 1280 * 1024 * 3
    factor = known / (cal[0]["FETCH_SIZE"] * 1024) if cal and cal[0].get("FETCH_SIZE") else None
    rec = {"kernel": "k_binary", "frames": 256, "width": 1280, "height": 1024, "FETCH_SIZE_KiB_raw": kb[0]["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": kb[0]["WRITE_SIZE"],
           "correction": "read bytes = 2 x FETCH_SIZE x 1024 (gfx950 under-count of wide coalesced reads, MI355X_MICROARCH.md HBM), write bytes = WRITE_SIZE x 1024",
           "calibration": {"what": "the same kernel with morph = none reads every input byte exactly once (no halo rows): known bytes / (FETCH_SIZE x 1024) for its 12-B-per-lane wave-coalesced loads",
                           "known_read_bytes": known, "FETCH_SIZE_KiB_raw": cal[0]["FETCH_SIZE"] if cal else None, "factor": factor},
           "hbm_bytes_per_launch": int((2 * kb[0]["FETCH_SIZE"] + kb[0]["WRITE_SIZE"]) * 1024),
           "hbm_bytes_per_launch_calibrated": int((factor * kb[0]["FETCH_SIZE"] + kb[0]["WRITE_SIZE"]) * 1024) if factor else None}
    json.dump(rec, open(out + "/k_binary_traffic.json", "w"), indent=1)
    print(rec)
for k, cs in res.items():
    print(k, {c: round(v, 1) for c, v in sorted(cs.items())})
for d in ("serial", "default"):
    f = glob.glob(out + "/" + d + "/**/*kernel_stats.csv", recursive=True)
    if f:
        print(d, [(r["Name"].split("(")[0][-28:], r["Calls"], r["AverageNs"]) for r in csv.DictReader(open(f[0])) if "rmcv" in r["Name"]][:8])
PY
tail -c 600 $out/pmc_sq1.err; tail -c 300 $out/pmc_sq2.err; tail -c 300 $out/pmc_tcc.err
