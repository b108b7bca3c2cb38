#!/bin/bash
# round 6, visit D: head_fwd_kernel on twelve waves; full -m gpu suite; c1 / c2 / c3 bench lines
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6d; mkdir -p $O
for cfg in c1 c2 c3; do
  python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-unfused-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', d['ms_per_step'], d['repeats']['ms_per_step_median'], d['final_loss'])"
done | tee $O/bench.txt
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -8 $O/pytest_gpu.log
