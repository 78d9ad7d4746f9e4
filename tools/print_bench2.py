# dev tool: the figures of a bench line this round looks at
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], d["timed_region"]["ms_per_step_each"], "max submit", d["timed_region"].get("max_submit_host_ms_each"), "blocking", d["timed_region"].get("host_blocking_calls"))
print("steady", d["steady_state"]["ms_per_step"], "collect", (d.get("steps_with_collect") or {}).get("ms_per_step"), "roofline", d["roofline"]["frac"], d["roofline"]["in_schedule_frac"], "traffic", d["roofline"]["traffic"])
for lv in d.get("density_sweep", {}).get("levels", []):
    print({k: lv.get(k) for k in ("stream", "ms_per_step", "ms_per_step_each", "max_submit_host_ms", "frames_mid_tier", "x_plain")})
print("c5", d.get("c5", {}).get("ms_per_step"), d.get("c5", {}).get("hbm_frac"), "lone", d["lone_batch_ms"]["median"], "fused", d["stage_ms"]["fused_sparse"], "c2", d.get("c2_binary_only"))
sf = d.get("single_frame_ms", {})
for k in ("c_host", "c_host_beside_this_process", "c_host_beside_this_process_runtime_copies_only", "c_host_beside_this_process_own_paths_only"):
    v = sf.get(k) or {}
    print(k, {m: (v[m]["median_ms"], v[m]["p90_ms"]) for m in v if isinstance(v[m], dict) and "median_ms" in v[m]} or v)
print("ctypes", {k: sf[k]["median_ms"] for k in sf if isinstance(sf[k], dict) and "median_ms" in sf[k] and "extract_color_ms" in sf[k]})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_16_threads", {}))
