"""Mean per-dispatch counter values of one kernel from rocprofv3 --pmc CSV output.
usage: pmc_summary.py <kernel substring> <dir or *_counter_collection.csv> [...]"""
import csv, glob, os, sys, collections, json
sub = sys.argv[1]
files = []
for a in sys.argv[2:]:
    files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
acc = collections.defaultdict(list)
dur = collections.defaultdict(list)
for f in files:
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            dur[row["Dispatch_Id"] + f].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
out = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
out["_dispatches"] = max((len(v) for v in acc.values()), default=0)
out["_kernel_ms_mean"] = sum(v[0] for v in dur.values()) / max(len(dur), 1)
print(json.dumps(out, indent=1))
