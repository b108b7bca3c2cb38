#!/bin/bash
set -u
OUT=gpurun_out/r5e
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
echo "== pytest"; timeout 600 python -m pytest tests/test_gpu_lstm.py -m gpu -q 2>&1 | tail -3
ab() { echo "== c4 $1"; shift; env "$@" timeout 600 python bench.py --config c4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d.get('repeats',{}); print(d['ms_per_step'], d['value'], r.get('ms_per_step_median'), r.get('ms_per_step_min'), r.get('ms_per_step_max'), d['final_loss'])"; }
ab "default" A=1
ab "fused" HOWL_LSTM_RIDE_X=0
ab "default" A=1
cd /tmp && export TMPDIR=/tmp
prof() { tag=$1; shift; echo "== $tag"; env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$tag -o c4 -- python $R/bench.py --config c4 --steps 20 --warmup 3 --prewarm 5 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof_$tag.log 2>&1
  t=$(find $R/$OUT/prof_$tag -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/tools/step_timeline.py "$t" | grep "lstm_fwd4\|step:" ; rm -rf $R/$OUT/prof_$tag; }
prof ride A=1
prof ride_waitall HOWL_LSTM_RIDE_DBG=2
