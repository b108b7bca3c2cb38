# NOTE (round 5): the HOWL_DIAG_* branches this tool compiles were removed from the product kernels (tools/strip_diag.py);
# it builds against the sources of commit 37e3835 (`git worktree add /tmp/howl_r4 37e3835` and run it there).
"""Round-4 A/B harness for the res8 kernels: builds copies of the library from EDITED copies of csrc/res8.hip (text
substitutions, tools only -- nothing here ships) and times the c3 step with each on one GPU box, same minute.
    python tools/variants4.py build [name ...]     (here; hipcc cross-compiles, builds run side by side)
    python tools/variants4.py run [name ...]       (GPU box; prints step / pair / forward-conv / conv0 times per library)
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "howl_amd" / "csrc" / "res8.hip"
OUT = ROOT / "build" / "diag"

# name -> list of (old, new) substitutions; each `old` must occur exactly once unless a count is given as a third element
EDITS = {
    "base": [],
    "epi_after": [("-D", "HOWL_DIAG_EPI_AFTER")],      # epilogue behind the second barrier
    "nostage": [("-D", "HOWL_DIAG_NOSTAGE")],          # timing only (WRONG results): the conv phases without their staging work
    "desync": [("-D", "HOWL_DIAG_DESYNC")],
    "w_nostage": [("-D", "HOWL_DIAG_WNOSTAGE")], "w_nolds": [("-D", "HOWL_DIAG_WGRAD_NOLDS")], "w_nomfma": [("-D", "HOWL_DIAG_WGRAD_NOMFMA")],
    "w_nostage_nolds": [("-D", "HOWL_DIAG_WNOSTAGE"), ("-D", "HOWL_DIAG_WGRAD_NOLDS")],
    "conv_empty": [("-D", "HOWL_DIAG_CONV_EMPTY")], "conv_prologue": [("-D", "HOWL_DIAG_CONV_PROLOGUE_ONLY")],
    "k_nopipe": [("-D", "HOWL_DIAG_K_NOPIPE")], "knounroll": [("-D", "HOWL_DIAG_KNOUNROLL")], "wunroll2": [("-D", "HOWL_DIAG_WUNROLL2")], "nostair": [("-D", "HOWL_DIAG_NOSTAIR")],
    "wino": [("-D", "HOWL_DIAG_WINO")],                # timing skeleton of a Winograd F(2x2,3x3) forward (WRONG results), see res8.hip
    "c0_nostore": [("-D", "HOWL_DIAG_C0_NOSTORE")], "c0_nomfma": [("-D", "HOWL_DIAG_C0_NOMFMA")],
    "c0_noload": [("-D", "HOWL_DIAG_C0_NOLOAD")], "c0_noepi": [("-D", "HOWL_DIAG_C0_NOEPI")],
    "c0w_nocompute": [("        for (int cell = c0 + wave; cell < c1; cell += C0W_THREADS / 64) {", "        for (int cell = c0 + wave; cell < c1 && B < 0; cell += C0W_THREADS / 64) {")],
    "c0w_noload": [("        for (int i0 = tid; i0 < NMAP * nc; i0 += 8 * C0W_THREADS) {   // bulk, 8 loads in flight per thread: the slice's cells of",
                    "        for (int i0 = tid; i0 < NMAP * nc && B < 0; i0 += 8 * C0W_THREADS) {   // bulk, 8 loads in flight per thread: the slice's cells of")],
    "nostage_nomidbar": [("-D", "HOWL_DIAG_NOSTAGE"), ("-D", "HOWL_DIAG_NOMIDBAR")],
    "nostage_nolds": [("-D", "HOWL_DIAG_NOSTAGE"), ("-D", "HOWL_DIAG_CONV_NOLDS")],
    "nostage_nomfma": [("-D", "HOWL_DIAG_NOSTAGE"), ("-D", "HOWL_DIAG_CONV_NOMFMA")],
    "nostage_nok": [("-D", "HOWL_DIAG_NOSTAGE"), ("-D", "HOWL_DIAG_CONV_NOK")],            # the three waves of a SIMD take their staging bursts at different K groups
}

def build(names):
    OUT.mkdir(parents=True, exist_ok=True)
    text = SRC.read_text()
    objs = [str(p) for p in sorted((ROOT / "build" / "obj").glob("*.o")) if p.name != "res8.o"]
    jobs = []
    for name in names:
        t = text
        probe = name.endswith("+probe")      # "<variant>+probe": the same edits in a -DHOWL_DIAG_PROBE build (tools/probe_step4.py run <lib>)
        defs = ["-DHOWL_DIAG_PROBE"] if probe else []
        for e in EDITS[name[:-6] if probe else name]:
            if e[0] == "-D":
                defs.append("-D" + e[1])
                continue
            old, new = e[0], e[1]
            cnt = e[2] if len(e) > 2 else 1
            assert t.count(old) == cnt, (name, old[:70], t.count(old))
            t = t.replace(old, new)
        name = name.replace("+", "_")
        (OUT / "_src").mkdir(exist_ok=True)      # not next to the product sources: builders glob csrc/*.hip
        tmp = OUT / "_src" / f"{name}.hip"
        tmp.write_text(t)
        obj = OUT / f"res8_{name}.o"
        jobs.append((name, tmp, obj, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", f"-I{SRC.parent}"] +
                                                       defs + ["-c", str(tmp), "-o", str(obj)])))
    for name, tmp, obj, proc in jobs:
        rc = proc.wait()
        tmp.unlink()
        assert rc == 0, name
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT / f"libhowl_{name}.so"), str(obj)] + objs,
                       check=True)
        print("built", name, flush=True)


def run(names, extra):
    libs = [(n, OUT / f"libhowl_{n}.so") for n in names]
    for rep in range(2):
        for name, lib in libs:
            env = dict(os.environ, HOWL_HIP_LIBRARY=str(lib), NUM_MELS="40")
            r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--steps", "40", "--warmup", "8"] + extra,
                               env=env, capture_output=True, text=True, timeout=300)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                ro = d["roofline"]
                fwd = [v.get("avg_launch_ms") for k, v in ro["other_kernels"].items() if k.startswith("conv3x3")]
                c0 = [(v.get("fwd_avg_launch_ms"), v.get("wgrad_avg_launch_ms")) for k, v in ro["other_kernels"].items() if k.startswith("conv0")]
                print(f"{name:18s} conv0 {c0} step {d['ms_per_step']:.4f} ms (median {d['repeats']['ms_per_step_median']:.4f})  pair {ro['avg_launch_ms']:.4f}  fwd {fwd}  "
                      f"loss {d['final_loss']}", flush=True)
            except Exception:
                print(name, "FAILED", r.stderr[-600:], flush=True)


if __name__ == "__main__":
    mode, rest = sys.argv[1], sys.argv[2:]
    names = [a for a in rest if not a.startswith("-")] or list(EDITS)
    build(names) if mode == "build" else run(names, [a for a in rest if a.startswith("-")])
