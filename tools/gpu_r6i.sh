#!/bin/bash
# round 6, visit I: CTC mean rides in the slab fold; c4 A/B again; lstm + mobilenet + engine tests
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6i; mkdir -p $O
python -m pytest tests/test_gpu_lstm.py tests/test_gpu_mobilenet.py tests/test_gpu_engine.py -m gpu -q > $O/pytest.log 2>&1; echo "tests rc=$?"; tail -4 $O/pytest.log
for rep in 1 2; do for fused in 1 0; do for la in "" "--no-lookahead"; do
  HOWL_SEQ_HEAD_FUSED=$fused python bench.py --config c4 --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-unfused-leg $la 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 fused=$fused $la', d['ms_per_step'], d['repeats']['ms_per_step_median'], d['final_loss'])"
done; done; done | tee $O/c4_ab.txt
