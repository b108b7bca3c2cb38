#!/bin/bash
# c4 (seq-lstm CTC step) on one GPU box: LSTM tests, bench line, rocprof step timeline.   bash tools/gpu_c4.sh [tag]
set -u
TAG=${1:-c4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_lstm.py -q 2>&1 | tail -3
for e in 0 1 2; do
  echo "== bench c4 (0: default, 1: HOWL_GEMM_NO_ROWGEMM, 2: HOWL_LSTM_NO_FUSED_X) $e"
  unset HOWL_GEMM_NO_ROWGEMM HOWL_LSTM_NO_FUSED_X; [ $e = 1 ] && export HOWL_GEMM_NO_ROWGEMM=1; [ $e = 2 ] && export HOWL_LSTM_NO_FUSED_X=1
  timeout 600 python bench.py --config c4 --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench_lines.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d.get('repeats',{}).get('ms_per_step_median'))"
done
unset HOWL_GEMM_NO_ROWGEMM HOWL_LSTM_NO_FUSED_X
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_c4 -o c4 -- python $R/bench.py --config c4 --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof_c4.log 2>&1
f=$(find $R/$OUT/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$OUT/c4_kernel_stats.csv
t=$(find $R/$OUT/prof_c4 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/tools/step_timeline.py "$t" > $R/$OUT/c4_step_timeline.txt && cat $R/$OUT/c4_step_timeline.txt
rm -rf $R/$OUT/prof_c4
