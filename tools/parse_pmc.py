"""Average rocprofv3 --pmc counters per dispatch of wf_step_kernel from gpurun_out/pmc_<tag>/*/counter_collection.csv."""
import csv, glob, json, sys, collections
tag = sys.argv[1]
kernel = sys.argv[2] if len(sys.argv) > 2 else "wf_step_kernel"  # substring of the kernel name to average over
res = {}
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
        res[k + "_n"] = len(v)
print(json.dumps(res, indent=1))
