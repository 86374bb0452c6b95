#!/bin/bash
# usage (on the GPU box, via gpurun): tools/gpu_check.sh TAG  -> gpu tests + bench, outputs under gpurun_out/
set -o pipefail
TAG=${1:-x}
mkdir -p gpurun_out
timeout -k 10 ${GPU_TEST_TIMEOUT:-600} python -m pytest tests -m gpu -x -v --timeout 240 > gpurun_out/gpu_tests_$TAG.log 2>&1
echo "pytest rc=$?" >> gpurun_out/gpu_tests_$TAG.log
tail -3 gpurun_out/gpu_tests_$TAG.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
echo "bench rc=$?"
python - <<PY
import json
try:
    d = [json.loads(l) for l in open("gpurun_out/bench_$TAG.json") if l.startswith("{")][-1]
    print("value %.4g evals/s  ms/step %.3f  kernel_ms %.3f  frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
    for k, v in d.get("other_configs", {}).items():
        print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if not isinstance(b, dict)})
except Exception as e:
    print("bench parse failed", e)
    print(open("gpurun_out/bench_$TAG.err").read()[-2000:])
PY
