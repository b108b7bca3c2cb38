"""Condense gpurun_out/pmc/pass*_counter_collection.csv (from tools/pmc_round.sh) into profiles/<name>."""
import collections
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
name = sys.argv[1] if len(sys.argv) > 1 else "pmc_summary.txt"
note = sys.argv[2] if len(sys.argv) > 2 else ""
merged = collections.defaultdict(dict)
for ps in (1, 2, 3, 4):
    p = ROOT / "gpurun_out" / "pmc" / f"pass{ps}_counter_collection.csv"
    if not p.exists():
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(p.open()):
        n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for n, d in agg.items():
        if n.startswith("__amd") or n.startswith("at::"):
            continue
        for c, v in d.items():
            merged[n][c] = round(sum(v) / len(v), 1)
        merged[n]["launches"] = len(next(iter(d.values())))
with (ROOT / "profiles" / name).open("w") as f:
    f.write("rocprofv3 --pmc passes over `python bench.py --steps 4 --warmup 2` (tools/pmc_round.sh); per-kernel means per launch.\n"
            "Separate passes: {SQ_*, GRBM_GUI_ACTIVE}, {SQ LDS/VALU}, {FETCH_SIZE}, {WRITE_SIZE} (KB; gfx950 FETCH_SIZE under-counts\n"
            "wide coalesced reads by 2x per MI355X_MICROARCH.md).  MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs /\n"
            "(GRBM_GUI_ACTIVE / 8 XCDs).\n" + (note + "\n" if note else "") + "\n")
    for n in sorted(merged):
        d = merged[n]
        extra = ""
        if d.get("SQ_VALU_MFMA_BUSY_CYCLES") and d.get("GRBM_GUI_ACTIVE"):
            extra = f"  mfma_util={d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / (d['GRBM_GUI_ACTIVE'] / 8):.3f}"
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            extra += f"  hbm_MB_per_launch(read x2 corrected)={(2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024 / 1e6:.1f}"
        f.write(f"{n}:{extra}\n    {json.dumps(d)}\n")
print("wrote", ROOT / "profiles" / name)
