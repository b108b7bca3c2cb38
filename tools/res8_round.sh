#!/bin/bash
# GPU-box visit for res8: parity tests, bench lines at the small and the headline batch.
set -u
mkdir -p gpurun_out
export NUM_MELS=40
echo "== pytest res8" ; timeout 900 python -m pytest tests/test_gpu_res8.py tests/test_gpu_ddp.py -x -q 2>&1 | tee gpurun_out/pytest_res8.log | tail -6
for a in "--config c1" "--config c1 --batch-per-gpu 16" "--config c1 --batch-per-gpu 1" "--config c2" "--config c3" "--config eval"; do
  echo "== bench $a"; timeout 300 python bench.py $a --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d.get('roofline') or {}
o = r.get('other_kernels') or {}
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step | pair', r.get('avg_launch_ms'), r.get('frac'), '| fwd', [ (v.get('avg_launch_ms'), v.get('frac')) for k, v in o.items() if k.startswith('conv3x3')], '| logmel', (o.get('logmel') or {}).get('avg_launch_ms'), d.get('speedup_vs_one_window_per_launch'), d.get('agreement'))"
done
