"""Compact a rocprofv3 --kernel-trace --stats kernel_stats.csv (kernel names truncated) into profiles/."""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
with open(dst, "w") as f:
    f.write("kernel,calls,total_ms,avg_us,pct,min_us,max_us\n")
    for r in rows:
        name = r["Name"].split("(")[0].replace("void ", "")[:70]
        f.write(f'"{name}",{r["Calls"]},{float(r["TotalDurationNs"])/1e6:.3f},{float(r["AverageNs"])/1e3:.2f},{r["Percentage"]},{float(r["MinNs"])/1e3:.2f},{float(r["MaxNs"])/1e3:.2f}\n')
print(open(dst).read())
