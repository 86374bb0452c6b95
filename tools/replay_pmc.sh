#!/bin/bash
# On the GPU box: where the parity mode's stream walk (mcq_mt_parse_kernel) spends its cycles -- issue and wait counters per
# launch of BASELINE configs[2] (4096 x 3 players x 50 000 runs)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/replaypmc_$$; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/rp.py <<PY
import sys
sys.path.insert(0, "$R")
import numpy as np
import neuron_poker_amd as npa
from bench import make_states
eng = npa.Engine(0)
hole, board = make_states(4096, 0)
q = npa.pack_queries(hole, board, 3, 50000)
for i in range(2):
    eng.eval_batch(q, seed=i, mode=npa.MODE_REPLAY_MT19937)
PY
pass() { rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$1 -o p -- python3 /tmp/rp.py > $OUT/log_$1 2>&1 || { tail -3 $OUT/log_$1; exit 1; }; }
pass SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES
pass SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE
pass SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VSKIPPED SQ_INSTS_FLAT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mcq_mt_parse_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
for k in sorted(m):
    print("%-24s %.4g" % (k, m[k]))
PY
