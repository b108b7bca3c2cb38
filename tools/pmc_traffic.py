"""HBM traffic per STEP and kernel from the PMC passes of tools/pmc_round.sh (pass3 = FETCH_SIZE, pass4 = WRITE_SIZE, KB;
separate rocprofv3 --pmc runs, read side doubled per the gfx950 correction in MI355X_MICROARCH.md) ->
profiles/<name>.  bench.py's c5 roofline reads the `total:` line.
    python tools/pmc_traffic.py round4_c5_hbm_traffic.txt 6 [algorithmic_MB] [kernel-name filter ...]"""
import collections
import csv
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
name, steps = sys.argv[1], int(sys.argv[2])
algo = sys.argv[3] if len(sys.argv) > 3 else None
keep = sys.argv[4:]
tot = collections.defaultdict(lambda: [0.0, 0.0, 0])
for ps, col in ((3, 0), (4, 1)):
    p = ROOT / "gpurun_out" / "pmc" / f"pass{ps}_counter_collection.csv"
    for r in csv.DictReader(p.open()):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if n.startswith(("__amd", "at::")) or (keep and not any(k in n for k in keep)):
            continue
        tot[n][col] += float(r["Counter_Value"]) * 1024.0 * (2.0 if col == 0 else 1.0) / 1e6
        if col == 0:
            tot[n][2] += 1
lines = [f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_round.sh) over `bench.py --steps {steps - 2} --warmup 2`:",
         f"HBM megabytes per STEP ({steps} steps incl. warm-up averaged; FETCH_SIZE x2: the gfx950 under-count of wide reads, "
         "MI355X_MICROARCH.md).", ""]
rd = wr = 0.0
for n, (r, w, c) in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    lines.append(f"{n:46s} {c / steps:5.1f} launches/step   read {r / steps:8.1f} MB   write {w / steps:8.1f} MB")
    rd += r / steps
    wr += w / steps
lines += ["", f"total: read {rd:.0f} MB + write {wr:.0f} MB = {rd + wr:.0f} MB per step" +
          (f"  (algorithmic bytes of the fused convolution launches: {algo} MB)" if algo else "")]
(ROOT / "profiles" / name).write_text("\n".join(lines) + "\n")
print("\n".join(lines[-1:]))
