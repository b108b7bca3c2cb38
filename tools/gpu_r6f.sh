#!/bin/bash
# round 6, visit F: the one-pass 80-bin frontend (A/B against the two-launch form inside one library); frontend + lstm tests
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
python -m pytest tests/test_gpu_frontend.py tests/test_gpu_lstm.py -m gpu -q > $O/pytest_fe_lstm.log 2>&1; echo "fe+lstm rc=$?"; tail -3 $O/pytest_fe_lstm.log
python -m pytest tests/test_gpu_res8.py -m gpu -q -k "80_mel or stock_80" > $O/pytest_res8_80.log 2>&1; echo "res8 80 rc=$?"; tail -3 $O/pytest_res8_80.log
for two in 0 1; do
  if [ $two = 1 ]; then export HOWL_LOGMEL_TWO_LAUNCHES=1; else unset HOWL_LOGMEL_TWO_LAUNCHES; fi
  NUM_MELS=80 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-unfused-leg > $O/bench_c3_m80_two$two.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/bench_c3_m80_two$two.json").read().strip().splitlines()[-1])
print("c3 m80 two_launches=$two", d["ms_per_step"], d["repeats"]["ms_per_step_median"], d["final_loss"], d["roofline"]["other_kernels"].get("logmel"))
PY
done
unset HOWL_LOGMEL_TWO_LAUNCHES
NUM_MELS=80 python bench.py --config c4 --steps 40 --no-cpu-baseline --no-unfused-leg --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c4 m80', d['ms_per_step'], d['repeats']['ms_per_step_median'])"
