"""Developer tool: print a bench.py JSON line as a table."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], "img/s", d["ms_per_step"], "ms", d["step_ms"], "cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None)
    r = d["roofline"]
    print(" roofline", {k: v for k, v in r.items() if k not in ("note", "traffic_source", "hbm_kernel")})
    print(" hbm_kernel", r.get("hbm_kernel"))
    h = d["hot_path"]
    print(" hot sum", h["sum_kernel_us_per_img"], "nms+roi", h["nms_plus_roi_us_per_img"], "proposal stage", h["proposal_stage_us_per_img"])
    for k, v in sorted(h["kernels"].items(), key=lambda kv: -kv[1]["us_per_img"]):
        print("   %-28s %8.1f us/img  avg %7.1f med %7.1f p10 %7.1f p90 %7.1f n=%d %-7s hbm_frac=%s" % (k, v["us_per_img"], v["avg_us"], v["median_us"], v["p10_us"], v["p90_us"], v["launches"], v["bound"], v["hbm_frac"]))
