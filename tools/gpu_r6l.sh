#!/bin/bash
# round 6, visit L: upper bound of a compact halo copy at 80 mel bins (timing only); soak (bit-repeatable training runs)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6l; mkdir -p $O
NUM_MELS=80 timeout 900 python tools/variants6.py run base halo_compact --config c3 > $O/variants_m80_halo.txt 2>&1; cat $O/variants_m80_halo.txt
timeout 900 python tools/soak.py > $O/soak.log 2>&1; tail -6 $O/soak.log
