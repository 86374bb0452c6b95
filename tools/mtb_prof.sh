#!/bin/bash
# On the GPU box: kernel trace of the parity mode by state blocks (tools/mtb_probe.py PLAYERS QUERIES) -> per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/mtb_prof; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o mtb -- python3 tools/mtb_probe.py ${1:-6} ${2:-1} > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
grep "per call" $OUT/log.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %5s  avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
