#!/bin/bash
# round 6, visit B: where a c1 / c2 step's time goes (ablation libraries) + the grid-barrier skeleton
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6b; mkdir -p $O
timeout 120 build/grid_barrier_probe 256 > $O/grid_barrier_probe.txt 2>&1; echo "probe rc=$?"; cat $O/grid_barrier_probe.txt
timeout 900 python tools/variants6.py run --config c1 > $O/variants_c1.txt 2>&1; cat $O/variants_c1.txt
timeout 600 python tools/variants6.py run base fwd_empty fwd_skip fwd_prologue pair_empty pair_d_prologue --config c2 > $O/variants_c2.txt 2>&1; cat $O/variants_c2.txt
