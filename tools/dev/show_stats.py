"""Developer tool: print a rocprofv3 kernel_stats.csv (optionally only rows whose name contains a pattern)."""
import csv, sys
f = sys.argv[1]; pat = sys.argv[2:] 
for r in csv.DictReader(open(f)):
    if pat and not any(p in r["Name"] for p in pat): continue
    print("%-64s calls %5s avg %8.1f us  min %7.1f max %7.1f total %9.1f us" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
