#!/bin/bash
# c1 (64 x 1 s): the two roles of the backward pair launched apart (HOWL_RES8_BWD_PAIR=0) under rocprof: where do the 24 us go?
OUT=gpurun_out/${1:-c1roles}
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in pair apart; do
  env=""; [ $mode = apart ] && export HOWL_RES8_BWD_PAIR=0
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$mode -o c1 -- python $R/bench.py --config c1 --steps 20 --warmup 3 --prewarm 5 --no-cpu-baseline --no-roofline > $R/$OUT/rocprof_$mode.log 2>&1
  t=$(find $R/$OUT/prof_$mode -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python $R/tools/step_timeline.py "$t" logmel > $R/$OUT/c1_${mode}_timeline.txt
  rm -rf $R/$OUT/prof_$mode
  unset HOWL_RES8_BWD_PAIR
done
cat $R/$OUT/c1_apart_timeline.txt
