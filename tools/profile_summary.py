#!/usr/bin/env python
"""Per-step kernel summary of a rocprofv3 --kernel-trace run of bench.py / tools/hotpath_bench.py.

rocprofv3's own --stats table averages over the WHOLE process, including the MIOpen find phase of the first steps.  This tool cuts
the trace at the step boundaries (one `proposal_prologue_kernel` launch per step) and aggregates the LAST `--steps` full steps:
per kernel name calls per step, average / min / max launch duration, time per step, share of the step's GPU time.

    python tools/profile_summary.py gpurun_out/prof_x/*/*_kernel_trace.csv --steps 8 > profiles/r02_x_kernel_summary.csv
"""
import argparse
import collections
import csv
import sys


def short(name):
    n = name.replace("void ", "")
    cut = n.find("(")
    if cut > 0:
        n = n[:cut]
    return n[:96]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--marker", default="proposal_prologue_kernel")
    a = ap.parse_args()
    rows = sorted(csv.DictReader(open(a.trace)), key=lambda r: int(r["Start_Timestamp"]))
    marks = [int(r["Start_Timestamp"]) for r in rows if a.marker in r["Kernel_Name"]]
    if len(marks) < a.steps + 1:
        raise SystemExit("only %d step markers in the trace" % len(marks))
    t0, t1 = marks[-a.steps - 1], marks[-1]
    agg = collections.OrderedDict()
    busy = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < t0 or s >= t1:
            continue
        d = agg.setdefault(short(r["Kernel_Name"]), [0, 0, 10 ** 18, 0])
        d[0] += 1; d[1] += e - s; d[2] = min(d[2], e - s); d[3] = max(d[3], e - s)
        busy += e - s
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "calls_per_step", "avg_us", "min_us", "max_us", "us_per_step", "pct_of_gpu_busy_time"])
    for k, (n, tot, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, "%.2f" % (n / a.steps), "%.2f" % (tot / n / 1e3), "%.2f" % (mn / 1e3), "%.2f" % (mx / 1e3), "%.2f" % (tot / a.steps / 1e3),
                    "%.2f" % (100.0 * tot / busy)])
    w.writerow(["# window", "%d steps" % a.steps, "wall %.1f us/step" % ((t1 - t0) / a.steps / 1e3), "gpu busy %.1f us/step" % (busy / a.steps / 1e3), "", "", ""])


if __name__ == "__main__":
    main()
