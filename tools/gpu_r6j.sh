#!/bin/bash
# round 6, visit J: where the fused head + CTC launch's time goes (parts removed, timing only)
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6j; mkdir -p $O
timeout 600 python tools/variants6.py run sh_base sh_noctc sh_nobwd sh_nofwd --config c4 --no-lookahead > $O/variants_c4.txt 2>&1; cat $O/variants_c4.txt
