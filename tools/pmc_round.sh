#!/bin/bash
# PMC passes (separate runs, counters only + kernel trace) over a short bench run.  Output: gpurun_out/pmc/
#   bash tools/pmc_round.sh [config]     (default: the headline config)
set -u
mkdir -p gpurun_out/pmc
export NUM_MELS=${NUM_MELS:-40}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
CFG=${1:-c3}
CMD="python $R/bench.py --config $CFG --steps 4 --warmup 2 --prewarm 0 --no-cpu-baseline --no-roofline --no-unfused-leg ${PMC_EXTRA:-}"     # PMC_EXTRA: e.g. "--seconds 2 --batch-per-gpu 256"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o pass$i -- $CMD > $R/gpurun_out/pmc/pass$i.log 2>&1
  echo "pass $i rc=$? : $set"
done
ls $R/gpurun_out/pmc | head -30
