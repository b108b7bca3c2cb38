#!/bin/bash
set -u
export NUM_MELS=40
timeout 1500 python -m pytest tests/test_gpu_res8.py tests/test_gpu_lstm.py -m gpu -q -k "optimiser or fused_step or fused_sequence" 2>&1 | tail -12
ab() { echo "== $1"; shift; cfg=$1; shift; env "$@" timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-roofline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('repeats',{}); print(d['ms_per_step'], d['value'], r.get('ms_per_step_median'), d['final_loss'])"; }
ab "c3 default" c3 A=1
ab "c3 separate adamw" c3 HOWL_NO_FOLD_ADAMW=1
