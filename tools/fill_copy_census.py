#!/usr/bin/env python3
"""Where the runtime's fill / copy dispatches of a bench run fall (525 fillBufferAligned + 264 copyBuffer
dispatches in a 23-step C2 profile -- inside the steps, or around them?).
usage: fill_copy_census.py <rocprofv3 *_kernel_trace.csv>   -> counts before the first step, inside the stepping region
(first to last launch of the bf16 tier's main kernel), and after it"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
main = [i for i, r in enumerate(rows) if ("bf16_filter_kernel<" in r["Kernel_Name"] and ", 2, " in r["Kernel_Name"]) or "bf16_filter8_kernel" in r["Kernel_Name"]]
first, last = main[0], main[-1]
def census(lo, hi):
    c = {}
    for r in rows[lo:hi]:
        n = r["Kernel_Name"]
        key = "fill" if "fillBuffer" in n else "copy" if "copyBuffer" in n else None
        if key:
            c[key] = c.get(key, 0) + 1
    return c
# the steps' own region starts with the query-pack kernel in front of the first main launch
packs = [i for i, r in enumerate(rows[:first]) if "pack_queries" in r["Kernel_Name"]]
start = packs[-1] if packs else first
print(json.dumps({"dispatches": len(rows), "main_launches": len(main),
                  "before_the_first_step": census(0, start), "inside_the_steps": census(start, last + 1),
                  "after_the_last_main_launch": census(last + 1, len(rows))}, indent=1))
