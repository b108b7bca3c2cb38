#!/bin/bash
# round 6, visit M: the sliced weight gradient's extra tiles dealt over three SIMDs (A/B against the library of the evidence visit = build/variants6/libhowl_base.so)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6m; mkdir -p $O
for rep in 1 2; do for lib in new old; do for cfg in "c1" "c1 --batch-per-gpu 16" "c1 --batch-per-gpu 32" "c1 --batch-per-gpu 128"; do
  if [ $lib = old ]; then export HOWL_HIP_LIBRARY=$PWD/build/variants6/libhowl_base.so; else unset HOWL_HIP_LIBRARY; fi
  python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-unfused-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib $cfg', d['ms_per_step'], d['repeats']['ms_per_step_median'], d['final_loss'])"
done; done; done | tee $O/wgrad_ex_ab.txt
unset HOWL_HIP_LIBRARY
python -m pytest tests/test_gpu_res8.py -m gpu -q > $O/pytest_res8.log 2>&1; echo "res8 rc=$?"; tail -3 $O/pytest_res8.log
