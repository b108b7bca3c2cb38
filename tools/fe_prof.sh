#!/bin/bash
# rocprofv3 kernel trace + PMC passes over a loop of log-mel launches (512 x 1 s).  Output: gpurun_out/fe_prof/
set -u
mkdir -p gpurun_out/fe_prof
export NUM_MELS=40
R=$GRAFT_REPO_ROOT
cat > /tmp/fe_loop.py <<PY
import os, sys
sys.path.insert(0, "$R")
import torch
from howl_amd import ops
from howl_amd.data.transform.transform import StandardAudioTransform
from howl_amd.utils.synth import synthetic_pcm
dev = torch.device("cuda:0")
std = StandardAudioTransform().to(dev).eval()
fbp = std._standard_fb()
pair = torch.tensor([0.0, 1.0], device=dev)
for B in [int(v) for v in os.environ.get("FE_BATCHES", "512,64,16,1").split(",")]:
    pcm = synthetic_pcm(B, 16000).to(dev)
    for _ in range(20): ops.logmel(pcm, fbp, 40, pair, layout=1)
    torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fe_prof -o trace -- python /tmp/fe_loop.py > $R/gpurun_out/fe_prof/trace.log 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$R/gpurun_out/fe_prof/trace_kernel_trace.csv")))
d = collections.defaultdict(list)
for r in rows:
    if "logmel" in r["Kernel_Name"]:
        d[(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = sorted(v)
    print("logmel grid", k, "n", len(v), "median %.2f us min %.2f max %.2f" % (v[len(v)//2], v[0], v[-1]))
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  FE_BATCHES=512 timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/fe_prof -o pass$i -- python /tmp/fe_loop.py > $R/gpurun_out/fe_prof/pass$i.log 2>&1
  echo "pass $i rc=$? : $set"
done
python - <<PY
import csv, collections, json
for ps in range(1, 6):
    try: rows = list(csv.DictReader(open("$R/gpurun_out/fe_prof/pass%d_counter_collection.csv" % ps)))
    except Exception as e: print("pass", ps, e); continue
    agg = collections.defaultdict(list)
    for r in rows:
        if "logmel" in r["Kernel_Name"] and r.get("Grid_Size", r.get("Grid_Size_X", "")) in ("196608",):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("pass", ps, json.dumps({k: round(sum(v) / len(v), 1) for k, v in agg.items()}), "n", {k: len(v) for k, v in agg.items()})
PY
