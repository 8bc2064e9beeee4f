#!/bin/bash
# usage (GPU box): tools/profile_cfg.sh <config> <tag> ["<more bench args>"] [steps]  -- rocprofv3 kernel-trace stats of one bench configuration
cd /tmp && export TMPDIR=/tmp
C=${1:-c2}; T=${2:-rXX}; X=${3:-}; K=${4:-20}
O=$GRAFT_REPO_ROOT/gpurun_out/prof_${T}_$C
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $C --steps $K --warmup 3 --no-cpu-baseline --no-verify $X > $O/bench.json 2> $O/bench.err
rm -f $O/*kernel_trace.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/p_kernel_stats.csv")))
for r in rows[:12]:
    print("%-90s calls %6s avg %10.1f us  %5s %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
tail -1 $O/bench.json | cut -c1-300
