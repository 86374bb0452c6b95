#!/bin/bash
# On the GPU box: effect of the small-batch task cut (MCQ_SPLIT_MAX) on the lock-step table driver and on
# single-query latency -> gpurun_out/split_probe.txt
mkdir -p gpurun_out
for s in 0 1 2 3 4; do
  echo "== MCQ_SPLIT_MAX=$s"
  MCQ_SPLIT_MAX=$s timeout -k 10 120 python tools/config5.py --lock-steps 3000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('512 tables: ms/lock-step %.4f kernel %.4f env-steps/s %.4g' % (d['ms_per_lock_step'], d['kernel_ms_last_lock_step_avg'], d['env_steps_per_s']))"
  MCQ_SPLIT_MAX=$s timeout -k 10 120 python tools/latency_probe.py 2>/dev/null | tail -6
done
echo "== large table counts (threads automatic)"
for t in 4096 32768; do
  timeout -k 10 200 python tools/config5.py --lock-steps 2000 --tables $t 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$t tables: ms/lock-step %.4f kernel %.4f env-steps/s %.4g queries/s %.4g' % (d['ms_per_lock_step'], d['kernel_ms_last_lock_step_avg'], d['env_steps_per_s'], d['equity_queries_per_s']))"
done
