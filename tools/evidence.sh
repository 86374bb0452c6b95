#!/bin/bash
# On the GPU box: the round's evidence in one go -> gpurun_out/<TAG>_*  (copy what is to be judged into profiles/)
TAG=${1:-rXX}; O=gpurun_out; mkdir -p $O
timeout -k 10 300 python bench.py > $O/${TAG}_bench_full.json 2> $O/${TAG}_bench_full.err; echo "bench rc=$?"
timeout -k 10 200 python tools/small_probe.py > $O/${TAG}_small_probe.txt 2>&1; echo "small rc=$?"
timeout -k 10 200 python tools/ext_probe.py > $O/${TAG}_ext_speed.txt 2>&1; echo "ext rc=$?"
timeout -k 10 100 python tools/ext_latency_probe.py > $O/${TAG}_ext_latency.txt 2>&1; echo "extlat rc=$?"
timeout -k 10 100 python tools/showdown_probe.py > $O/${TAG}_showdown_probe.txt 2>&1; echo "showdown rc=$?"
timeout -k 10 100 python tools/replay_probe.py > $O/${TAG}_replay_probe.txt 2>&1; echo "replay rc=$?"
MCQ_MT_BLOCKS=0 timeout -k 10 100 python tools/replay_probe.py > $O/${TAG}_replay_probe_serial_walk.txt 2>&1; echo "replay (serial) rc=$?"
timeout -k 10 100 python tools/ext_replay_scaling.py > $O/${TAG}_ext_replay_scaling.txt 2>&1; echo "ext replay scaling rc=$?"
bash tools/mtb_prof.sh 6 1 > $O/${TAG}_mtb_kernels.txt 2>&1; echo "mtb trace rc=$?"
for t in 512 4096 32768; do timeout -k 10 200 python tools/config5.py --tables $t --lock-steps 2000; done > $O/${TAG}_config5.txt 2>&1; echo "config5 rc=$?"
timeout -k 10 400 python tests/fuzz_parity.py --seconds ${FUZZ_SECONDS:-240} --seed 2026 > $O/${TAG}_fuzz_parity.txt 2>&1; echo "fuzz rc=$?"
tail -2 $O/${TAG}_fuzz_parity.txt
