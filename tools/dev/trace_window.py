"""Developer tool: print the kernels launched right before / after the n-th occurrence of a kernel in a rocprofv3 kernel_trace.csv."""
import csv, sys
f, pat, occ, before, after = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if pat in r["Kernel_Name"]]
i = idx[occ]
t0 = int(rows[i]["Start_Timestamp"])
for r in rows[max(0, i - before): i + after + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%+9.1f us  dur %8.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:110]))
