"""Developer tool: where is the GPU idle inside an eager step?  From a rocprofv3 kernel_trace.csv: the idle time between consecutive kernels (end -> next start),
summed per (kernel before, kernel after) pair over the last N steps (a step = from one fused-SGD launch to the next), largest first.
    python tools/dev/gap_report.py TRACE.csv [steps]"""
import csv, sys, collections
f = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "")[:70]
marks = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r["Kernel_Name"] or "FusedSgd" in r["Kernel_Name"] or "fused_sgd" in r["Kernel_Name"].lower()]
# the optimizer launches several kernels per step: a step boundary = a mark that follows a non-mark
bounds = [i for k, i in enumerate(marks) if k == 0 or marks[k - 1] != i - 1]
if len(bounds) < nsteps + 1:
    print("only", len(bounds), "steps found"); nsteps = len(bounds) - 1
lo, hi = bounds[-nsteps - 1], bounds[-1]
gaps = collections.Counter(); cnt = collections.Counter()
busy = idle = 0
for a, b in zip(rows[lo:hi], rows[lo + 1:hi + 1]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    busy += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    if g > 0:
        idle += g
        gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))] += g; cnt[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))] += 1
span = int(rows[hi]["Start_Timestamp"]) - int(rows[lo]["Start_Timestamp"])
print("%d steps: %.1f us per step, busy %.1f, idle %.1f (%d kernels per step)" % (nsteps, span / nsteps / 1e3, busy / nsteps / 1e3, idle / nsteps / 1e3, (hi - lo) // nsteps))
for (a, b), g in gaps.most_common(25):
    print("%8.1f us/step  x%-4.1f  %-60s -> %s" % (g / nsteps / 1e3, cnt[(a, b)] / nsteps, a, b))
