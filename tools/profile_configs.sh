#!/bin/bash
# rocprofv3 kernel-trace summaries of the other BASELINE configurations (seq-lstm CTC step, mobilenet step) -> gpurun_out/prof_cfg/
set -u
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_cfg
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg -o mobilenet -- python $R/tools/mb_step.py 5 > $R/gpurun_out/prof_cfg/mobilenet.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg -o lstm -- python $R/tools/lstm_step.py 5 > $R/gpurun_out/prof_cfg/lstm.log 2>&1
ls $R/gpurun_out/prof_cfg | head
