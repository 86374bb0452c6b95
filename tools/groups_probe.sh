#!/bin/bash
# On the GPU box: lock-step time of the native table driver by the number of stream groups (MCQ_TABLES_GROUPS)
for T in 512 4096; do for g in ${GROUPS_LIST:-1 2 3 4 5 6 8}; do
  MCQ_TABLES_GROUPS=$g timeout -k 10 200 python tools/config5.py --lock-steps 3000 --tables $T 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$T tables, $g groups: %.1f us per lock-step, %.3g env-steps/s' % (1e3*d['ms_per_lock_step'], d['env_steps_per_s']))"
done; done
