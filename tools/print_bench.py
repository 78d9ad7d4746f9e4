"""dev tool: the interesting fields of bench.py JSON lines"""
import json
import sys
for f in sys.argv[1:]:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e)
        continue
    tr = j.get("timed_region", {})
    print(f.split("/")[-1], "value", j["value"], "ms/step", j["ms_per_step"], "min", tr.get("ms_per_step_min"), "| k_binary", j["roofline"]["avg_launch_ms"],
          "frac", j["roofline"]["frac"], "| lone", j["lone_batch_ms"]["median"], "| path frac", j["path_hbm_frac"], "| fused", j["stage_ms"].get("fused_sparse"),
          "| no-image", j.get("detect_only_no_image", {}).get("fps"), "| c2", j.get("c2_binary_only"), "| cpu", (j.get("cpu_baseline") or {}).get("value"),
          (j.get("cpu_baseline_all_cores") or {}).get("value"), "| steady", (j.get("steady_state") or {}).get("ms_per_step"), "| sf", {k: v.get("median_ms") for k, v in (j.get("single_frame_ms") or {}).items() if isinstance(v, dict)})
