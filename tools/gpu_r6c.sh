#!/bin/bash
# round 6, visit C: the one-utterance-per-workgroup (SOLO) form of the 3x3 kernels, A/B inside one library (HOWL_RES8_SOLO=0 = phased)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6c; mkdir -p $O
for rep in 1 2; do
for cfg in c1 c2; do
for solo in 1 0; do
  HOWL_RES8_SOLO=$solo python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-unfused-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg solo=$solo', d['ms_per_step'], d['repeats']['ms_per_step_median'], d['final_loss'])"
done; done; done | tee $O/solo_ab.txt
for b in 16 1 128; do for solo in 1 0; do
  HOWL_RES8_SOLO=$solo python bench.py --config c1 --batch-per-gpu $b --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-unfused-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c1 batch $b solo=$solo', d['ms_per_step'], d['repeats']['ms_per_step_median'], d['final_loss'])"
done; done | tee -a $O/solo_ab.txt
python -m pytest tests/test_gpu_res8.py -m gpu -x -q > $O/pytest_res8.log 2>&1; echo "res8 rc=$?"; tail -5 $O/pytest_res8.log
python -m pytest tests/test_gpu_lstm.py tests/test_gpu_threads.py tests/test_gpu_engine.py -m gpu -x -q > $O/pytest_rest.log 2>&1; echo "rest rc=$?"; tail -5 $O/pytest_rest.log
