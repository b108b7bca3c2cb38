#!/bin/bash
OUT=gpurun_out/${1:-r5o}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_res8.py -q -m gpu > $OUT/pytest_res8.log 2>&1
tail -25 $OUT/pytest_res8.log
timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 10 > $OUT/bench_c3.jsonl 2> $OUT/bench_c3.err
python - <<'PY'
import json,sys
d=json.loads(open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r5o/bench_c3.jsonl").read().strip().splitlines()[-1])
print("c3", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["final_loss"])
PY
timeout 300 python bench.py --no-cpu-baseline --steps 30 --warmup 5 --seconds 2 --batch-per-gpu 256 > $OUT/bench_2s.jsonl 2> $OUT/bench_2s.err
tail -c 300 $OUT/bench_2s.err; python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/r5o/bench_2s.jsonl").read().strip().splitlines()[-1])
    print("2s x256", d["ms_per_step"], d["value"], d["final_loss"])
except Exception as e: print("2s bench failed", e)
PY
