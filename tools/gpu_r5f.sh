#!/bin/bash
set -u
OUT=gpurun_out/r5f
mkdir -p $OUT
export NUM_MELS=40
echo "== pytest res8 + engine"; timeout 1500 python -m pytest tests/test_gpu_res8.py tests/test_gpu_engine.py tests/test_gpu_ddp.py -m gpu -q 2>&1 | tee $OUT/pytest.log | tail -4
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
: > $OUT/bench_lines.jsonl
ab() { echo "== $1"; shift; cfg=$1; shift; env "$@" timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-roofline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('repeats',{}); print(d['ms_per_step'], d['value'], r.get('ms_per_step_median'), r.get('ms_per_step_min'), r.get('ms_per_step_max'), d['final_loss'])"; }
ab "c3 default" c3 A=1
ab "c3 separate adamw" c3 HOWL_NO_FOLD_ADAMW=1
ab "c3 default" c3 A=1
ab "c1 default" c1 A=1
ab "c1 separate adamw" c1 HOWL_NO_FOLD_ADAMW=1
ab "c2 default" c2 A=1
