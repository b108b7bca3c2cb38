#!/bin/bash
# rocprofv3 kernel stats of one BASELINE config:  bash tools/profile_config.sh c5 [tag]
set -u
C=${1:-c5}; TAG=${2:-r2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export NUM_MELS=40
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_$C -o $C -- python $GRAFT_REPO_ROOT/bench.py --config $C --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/$OUT/rocprof_$C.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof_$C -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-150 "$f" | head -${3:-40}
