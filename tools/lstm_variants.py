"""Ablation builds of the four-sequence LSTM recurrences (tools only): text edits on a temporary copy of lstm.hip, one shared
library per variant under build/diag/ (results of the variants are wrong by construction; only their timing is of interest).
   python tools/lstm_variants.py build            (here: hipcc cross-compiles)
   python tools/lstm_variants.py run              (GPU box: tools/lstm_recur.py against each library)"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "howl_amd" / "csrc" / "lstm.hip"
OUT = ROOT / "build" / "diag"
EDITS = {
    "base": [],
    "nomfma": [("        bcast_mfma64<0, 0>(alo, wB, acc);\n        bcast_mfma64<0, 64>(ahi, wB, acc);",
                "        acc[0][0] = alo.x * wB[0] + ahi.x * wB[64];"),
               ("        bcast_mfma64<0, 0>(alo, wk, acc);\n        bcast_mfma64<0, 64>(ahi, wk, acc);",
                "        acc[0][0] = alo.x * wk[0] + ahi.x * wk[64];")],
    "nostore": [("            stg(gates, gb[r], act[r]);\n", "            if (t == 0) stg(gates, gb[r], act[r]);\n"),
                ("        stg(cs, cb, cn);\n        stg(hseq, hb, live ? hn : 0.0f);      // padded outputs are zero\n",
                 "        if (t == 0) stg(cs, cb, cn);\n"),
                ("            go[0] = di;\n            go[HID] = df;\n            go[2 * HID] = dg;\n            go[3 * HID] = dov;\n            request(t - 1);",
                 "            if (t == 0) { go[0] = di; go[HID] = df; go[2 * HID] = dg; go[3 * HID] = dov; }\n            request(t - 1);")],
    "noload": [("            for (int r = 0; r < 4; ++r) nx[r] = ldg(gxn, gb[r]);\n        }\n        // A_j[i] comes from lane 4j+i: row i",
                "            for (int r = 0; r < 4; ++r) nx[r] += gxn == gx ? 0.5f : 0.25f;\n        }\n        // A_j[i] comes from lane 4j+i: row i"),
               ("            request(t - 1);\n        }\n        __syncthreads();", "            if (t == Tout - 1) request(t - 1);\n        }\n        __syncthreads();")],
}


def build():
    OUT.mkdir(parents=True, exist_ok=True)
    text = SRC.read_text()
    objs = [str(p) for p in sorted((ROOT / "build" / "obj").glob("*.o")) if p.name != "lstm.o"]
    for name, edits in EDITS.items():
        t = text
        for old, new in edits:
            assert t.count(old) == 1, (name, old[:50], t.count(old))
            t = t.replace(old, new)
        tmp = SRC.parent / f"_diag_{name}.hip"
        tmp.write_text(t)
        try:
            obj = OUT / f"lstm_{name}.o"
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-w", "-c",
                            str(tmp), "-o", str(obj)], check=True)
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT / f"libhowl_{name}.so"),
                            str(obj)] + objs, check=True)
        finally:
            tmp.unlink()
        print("built", name, flush=True)


def run():
    for name in EDITS:
        env = dict(os.environ, HOWL_HIP_LIBRARY=str(OUT / f"libhowl_{name}.so"))
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "lstm_recur.py")] + sys.argv[2:], env=env, capture_output=True, text=True)
        print(name, (r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
