#!/usr/bin/env python3
"""Mean durations of the bf16 first tier's two launches per step (scout-only launch, main launch) from a rocprofv3
--kernel-trace --stats CSV -> the JSON record bench.py quotes as roofline.kernel_ms_main / kernel_ms_scout.
usage: tools/kernel_split.py <kernel_stats.csv> <config key> [committed csv name] > profiles/rNN_<cfg>_kernel_split.json
The scout-only launch is the instantiation with MODE = 1 (bf16_filter_kernel<KS, M, RAD, CI, 1, ...>, or the wide kernel's
shorter launch); everything else of the dominant kernel's name is the main launch (bf16_filter_kernel<..., 2, ...> or
bf16_filter8_kernel)."""
import csv, json, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
cfg = sys.argv[2]
src = sys.argv[3] if len(sys.argv) > 3 else sys.argv[1]
main, scout, other = [], [], []
for r in rows:
    n = r["Name"]
    avg, calls = float(r["AverageNs"]) / 1e6, int(r["Calls"])
    m = re.search(r"bf16_filter_kernel<(\d+), (\d+), (\w+), (\w+), (\d+)", n)
    if m:
        (scout if m.group(5) == "1" else main).append((avg, calls, n[:90]))
    elif "bf16_filter8_kernel" in n:
        main.append((avg, calls, n[:90]))
    elif "bf16_wide_kernel" in n:
        other.append((avg, calls, n[:90]))
if other and not main:  # wide rows: one kernel name, two launches per step -- the split is not in the stats file
    print(json.dumps({"config": cfg, "source": src, "note": "wide kernel: scout and main share one instantiation",
                      "mean_ms": other[0][0], "calls": other[0][1]}, indent=1))
    sys.exit(0)
w = lambda xs: sum(a * c for a, c, _ in xs) / max(sum(c for _, c, _ in xs), 1)
n_main, n_scout = sum(c for _, c, _ in main), sum(c for _, c, _ in scout)
rec = {"config": cfg, "source": src, "main_ms": round(w(main), 4), "scout_ms": round(w(scout), 4), "main_calls": n_main,
       "scout_calls": n_scout, "main_kernel": main[0][2] if main else None, "scout_kernel": scout[0][2] if scout else None}
if n_scout * 2 < n_main:  # round 4: thresholds from the index's seed model -- the only scout launch left is the model's
    rec["calibration_scout_ms"] = rec["scout_ms"]  # calibration at index build (once, outside the steps)
    rec["scout_ms"] = 0.0
    rec["note"] = "no scout launch inside the steps (seed model, DESIGN.md 4.12); calibration_scout_ms ran once at index build"
print(json.dumps(rec, indent=1))
