#!/bin/bash
# round 6, visit O: the fused head launch's CTC on two waves per utterance (alpha / beta apart) against the one-wave form (build/ab/libhowl_prev.so)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6o; mkdir -p $O
for rep in 1 2; do for lib in new prev; do for la in "" "--no-lookahead"; do
  if [ $lib = prev ]; then export HOWL_HIP_LIBRARY=$PWD/build/ab/libhowl_prev.so; else unset HOWL_HIP_LIBRARY; fi
  python bench.py --config c4 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline --no-unfused-leg $la 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib c4 $la', d['ms_per_step'], d['repeats']['ms_per_step_median'], d['final_loss'])"
done; done; done | tee $O/ctc_pair_ab.txt
unset HOWL_HIP_LIBRARY
python -m pytest tests/test_gpu_lstm.py tests/test_gpu_engine.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -3 $O/pytest.log
