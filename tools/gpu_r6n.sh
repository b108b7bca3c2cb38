#!/bin/bash
# round 6, visit N: the whole -m gpu suite and smoke at HEAD (after the last kernel change), c1 / c3 / c4 lines
cd /root/repo; export TMPDIR=/tmp
O=gpurun_out/r6n; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tee $O/pytest_gpu.log | tail -10
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tee $O/smoke.log | tail -2
for a in "" "--config c1" "--config c4"; do timeout 600 python bench.py $a --cpu-baseline-seconds 6 2>&1 | tail -1 | tee -a $O/bench_lines.jsonl | cut -c1-240; done
