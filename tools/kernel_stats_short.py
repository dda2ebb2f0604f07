"""Print a rocprofv3 *kernel_stats.csv compactly: name (cut), calls, average us, total ms."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):5d}  avg {float(r["AverageNs"]) / 1e3:9.2f} us  total {float(r["TotalDurationNs"]) / 1e6:9.3f} ms')
