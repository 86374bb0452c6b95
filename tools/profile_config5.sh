#!/bin/bash
# On the GPU box: kernel trace of the native table driver (BASELINE configs[4]) -> gpurun_out/prof/<TAG>/
TAG=${1:-c5}
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o p -- python3 $R/tools/config5.py --lock-steps 2000 > $OUT/kt.log 2>&1 || exit 1
cat $OUT/kt/p_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/kt/p_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:len(rows) // 2 + 12]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(r["Kernel_Name"][:60], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          r.get("Grid_Size"), r.get("Workgroup_Size"))
PY
