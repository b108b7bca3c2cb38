#!/bin/bash
# Round-3 GPU visit: the whole -m gpu suite, then the bench lines (headline with the CPU leg, the other configurations and
# the evaluation line without it).  Logs under gpurun_out/.
set -u
mkdir -p gpurun_out
export NUM_MELS=40
echo "== pytest -m gpu" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -12
echo "== bench c3" ; timeout 600 python bench.py 2>&1 | tee gpurun_out/bench_c3.log | tail -1 | cut -c1-3000
for c in c1 c2 c4 c5 eval; do
  echo "== bench $c" ; timeout 300 python bench.py --config $c --no-cpu-baseline 2>&1 | tee gpurun_out/bench_$c.log | tail -1 | cut -c1-700
done
echo "== bench B=16 (envs/res8.env batch)"; timeout 300 python bench.py --config c1 --batch-per-gpu 16 --no-cpu-baseline --no-roofline 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/bench_b16.log
