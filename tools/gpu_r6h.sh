#!/bin/bash
# round 6, visit H: rocprofv3 step timelines of c4 (fused head + CTC launch on / off)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6h; mkdir -p $O
export NUM_MELS=40
cd /tmp && export TMPDIR=/tmp
for fused in 1 0; do
  HOWL_SEQ_HEAD_FUSED=$fused timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4_f$fused -o c4 -- python $R/bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-unfused-leg --no-lookahead > $O/rocprof_c4_f$fused.log 2>&1
  f=$(find $O/prof_c4_f$fused -name "*kernel_trace.csv" | head -1)
  python $R/tools/step_timeline.py $f lstm_fwd4 > $O/c4_nolookahead_fused${fused}_step_timeline.txt 2>&1; cat $O/c4_nolookahead_fused${fused}_step_timeline.txt
done
