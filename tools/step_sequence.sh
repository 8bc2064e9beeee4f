#!/bin/bash
# usage (GPU box): tools/step_sequence.sh <config>  -- the kernels one timed step dispatches, in order, with start offsets
cd /tmp && export TMPDIR=/tmp
C=${1:-c2}
O=$GRAFT_REPO_ROOT/gpurun_out/seq_$C
mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o p -- python3 $GRAFT_REPO_ROOT/bench.py --config $C --steps 3 --warmup 2 --no-cpu-baseline --no-verify > $O/bench.json 2> $O/bench.err
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/p_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last query-pack kernel on
packs = [i for i, r in enumerate(rows) if "pack_queries" in r["Kernel_Name"]]
i0 = packs[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
print("kernels in the trace:", len(rows), " in the last step:", len(rows) - i0)
PY
rm -f $O/p_kernel_trace.csv
