#!/bin/bash
# On the GPU box: instruction counters of the extended queries' stream walk (mcq_mt_parse_ext_kernel), 256 queries x 6 players
# x 20 000 runs, opponents restricted to the top quarter of the classes
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof/extreplay_$$; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
cat > /tmp/extreplay.py <<PY
import json, os, sys
import numpy as np
sys.path.insert(0, "$R")
import neuron_poker_amd as npa
from neuron_poker_amd import _lib
order = json.load(open(os.path.join("$R", "neuron_poker_amd", "preflop_classes.json")))
eng = npa.Engine(0)
g = np.random.default_rng(7)
cards = np.array([g.permutation(52)[:8] for _ in range(256)], np.uint8)
q = npa.pack_queries(cards[:, :2], np.full((256, 5), 255, np.uint8), 6, 20000)
ex = _lib.pack_query_ext(256, opp_range=_lib.range_bits(order[-int(169 * 0.25):]))
r = eng.eval_batch_ext(q, ex, 1, mode=npa.MODE_REPLAY_MT19937)
print("passes per iteration %.1f" % (r["passes"].sum() / (256 * 20000)))
PY
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -o p -- python3 /tmp/extreplay.py > $OUT/log 2>&1 || { tail -5 $OUT/log; exit 1; }
grep passes $OUT/log
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/p_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "mcq_mt_parse_ext_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
it = 256 * 20000
print({k: "%.3g" % v for k, v in agg.items()})
print("per iteration: VALU %.0f  SALU %.0f  LDS %.0f  wave cycles %.0f" % (agg["SQ_INSTS_VALU"] / it, agg["SQ_INSTS_SALU"] / it, agg["SQ_INSTS_LDS"] / it, agg["SQ_WAVE_CYCLES"] / it))
PY
