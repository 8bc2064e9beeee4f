"""Mean per-dispatch counter values of one kernel from rocprofv3 --pmc CSV output.
usage: pmc_summary.py <kernel substring> <dir or *_counter_collection.csv> [...]
Dispatches of the same name that are not the launch being measured are left out: a dispatch shorter than half the
longest one of that name (round 4: the seed model's calibration at index build is a scout-only launch of the filter
kernel -- for wide rows of the very same instantiation as the main launch); "_dispatches_left_out" counts them."""
import csv, glob, os, sys, collections, json
sub = sys.argv[1]
files = []
for a in sys.argv[2:]:
    files += glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True) if os.path.isdir(a) else [a]
rows = []
for f in files:
    for row in csv.DictReader(open(f)):
        if sub in row["Kernel_Name"]:
            rows.append((row["Dispatch_Id"] + f, row["Counter_Name"], float(row["Counter_Value"]),
                         (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6))
dur = {}
for d, _, _, ms in rows:
    dur.setdefault(d, ms)
longest = max(dur.values(), default=0.0)
keep = {d for d, ms in dur.items() if ms >= 0.5 * longest}
acc = collections.defaultdict(list)
for d, name, val, _ in rows:
    if d in keep:
        acc[name].append(val)
out = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
out["_dispatches"] = max((len(v) for v in acc.values()), default=0)
out["_dispatches_left_out"] = len(dur) - len(keep)
out["_kernel_ms_mean"] = sum(dur[d] for d in keep) / max(len(keep), 1)
print(json.dumps(out, indent=1))
