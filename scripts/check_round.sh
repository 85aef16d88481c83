#!/bin/bash
# GPU box: the gpu test suite, one bench line (no CPU baseline) and the per-rank wall times of a 1/8 slice
#   gpurun --timeout 1200 -- 'bash scripts/check_round.sh TAG'
tag=${1:-chk}
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1
tail -3 gpurun_out/${tag}_tests.log
python bench.py --no-cpu-baseline --pme-steps 0 > gpurun_out/${tag}_bench.json || exit 1
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench.json"))
print(d["value"], "ns/day", d["ms_per_step"], "ms/step  near", d["detail"]["near_kernel_us"], "boundary pass", d["detail"]["step_boundary_pass_us"])
PY
bash scripts/kstats_probe.sh ${tag}_w8 --cluster 1 --world 8 --rank 3 | grep world
