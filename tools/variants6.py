"""Round-6 A/B harness: libraries built from EDITED copies of csrc/res8.hip (text substitutions; tools only, nothing here ships,
most variants compute WRONG results by design: they time a kernel with one part removed) and the c1 / c2 step timed with each on
one GPU box, same minute.  The product sources carry no ablation branches; the edits live here.
    python tools/variants6.py build [name ...]          (here; hipcc cross-compiles, builds run side by side)
    python tools/variants6.py run [name ...] [--config c1 ...]   (GPU box: step time per library, two alternating passes)
"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "howl_amd" / "csrc" / "res8.hip"
SRC_LSTM = ROOT / "howl_amd" / "csrc" / "lstm.hip"
OUT = ROOT / "build" / "variants6"

KERNEL_FWD_ENTRY = ("    if (bid >= nblk) return;\n    conv3x3_body<MODE, SLICES, HALO>(cfg, in_stats, wp, res, out, xs, xs_stats, part, pool, B, H, bid, nblk, slice, fold, bfold, wf, sg);",)
LOOP_SWITCH = "    switch (ntw) {\n        case 5: conv_loop<MODE, 5, ts, HALO>(cl, epi, cfg, pk0, pk1, b, st0, st1, pslot, hs, gx); break;"

# name -> list of (old, new[, count]) substitutions
EDITS = {
    "base": [],
    # the forward 3x3 kernel returns at entry: what its six launches cost beyond their dispatch
    "fwd_empty": [(KERNEL_FWD_ENTRY[0], "    if (bid >= nblk || MODE == 0) return;\n    conv3x3_body<MODE, SLICES, HALO>(cfg, in_stats, wp, res, out, xs, xs_stats, part, pool, B, H, bid, nblk, slice, fold, bfold, wf, sg);")],
    # ... is not launched at all: what the six launches cost in full
    "fwd_skip": [("            launch_conv3x3<0>(SL, G, lc, stream, plain_tile(sv->s[i - 1]), in_stats,", "            if (getenv(\"HOWL_NEVER\")) launch_conv3x3<0>(SL, G, lc, stream, plain_tile(sv->s[i - 1]), in_stats,")],
    # prologue only (weights, statistics fold, first half tile): no K loop, no epilogue, no statistics
    "fwd_prologue": [(LOOP_SWITCH, "    if (MODE == 0) return;\n" + LOOP_SWITCH)],
    # the statistics fold over 4 of the producer's partial rows (upper bound for a producer-side pre-fold)
    "fwd_nofold": [("            const double acc = fold_part_column(fold.part, part_stride(fold.nparts), fold.nparts, (c8 < 4 ? 0 : CP) + ch, lane);",
                    "            const double acc = fold_part_column(fold.part, part_stride(fold.nparts), fold.nparts < 4 ? fold.nparts : 4, (c8 < 4 ? 0 : CP) + ch, lane);")],
    # a third of the packed weights staged (what a workgroup of a cout-tile partition would stage): forward AND data gradient
    "w_third": [("            wv[j] = (i < 3 * KSTEPS * 16) ? reinterpret_cast<const float4*>(wp)[i] : make_float4(0.f, 0.f, 0.f, 0.f);",
                 "            wv[j] = (i < KSTEPS * 16) ? reinterpret_cast<const float4*>(wp)[i] : make_float4(0.f, 0.f, 0.f, 0.f);"),
                ("            if (i < 3 * KSTEPS * 16) reinterpret_cast<float4*>(wl)[i] = wv[j];", "            if (i < KSTEPS * 16) reinterpret_cast<float4*>(wl)[i] = wv[j];")],
    # the pair launch returns at entry / is not launched
    "pair_empty": [("    if (j >= nblk) return;\n    if (r < SD)", "    if (j >= nblk || B > 0) return;\n    if (r < SD)")],
    # the data-gradient role of the pair returns after its prologue / the weight-gradient role returns at entry
    "pair_d_prologue": [(LOOP_SWITCH, "    if (MODE == 1) return;\n" + LOOP_SWITCH)],
    "pair_w_empty": [("    else\n        wgrad_body<SW, HALO>(WStage{zc, s_prev, false}, in_stats, bfold, wpart, B, H, j, nblk, r - SD, sg);",
                      "    else if (B < 0)\n        wgrad_body<SW, HALO>(WStage{zc, s_prev, false}, in_stats, bfold, wpart, B, H, j, nblk, r - SD, sg);")],
    "pair_d_empty": [("    if (r < SD)\n        conv3x3_body<1, SD, HALO>(", "    if (r < SD && B < 0)\n        conv3x3_body<1, SD, HALO>(")],
    # 80 mel bins (NUM_MELS=80): the halo column of a strip read from CONTIGUOUS addresses of the neighbour's block instead of one
    # element per cache line (wrong values, timing only): the upper bound of what a compact side copy of the edge columns could save
    "halo_compact": [("    hs.g = (tid < nch * H) ? (c0 + c) * P + h * PW : 0;", "    hs.g = (tid < nch * H) ? (c0 + c) * H + h : 0;"),
                     ("    const unsigned off = hs.pk >= 0 ? 4u * (unsigned)(hs.g + gcol) : 0u;\n    v.s = v.k = make_float2(0.0f, 0.0f);", "    const unsigned off = hs.pk >= 0 ? 4u * (unsigned)(hs.g) : 0u;\n    v.s = v.k = make_float2(0.0f, 0.0f);"),
                     ("    return WHalo{ok ? c * CSX + (h + row0) * WPW : -1, ok ? c * P + h * PW : 0, ok ? c : 0, ok ? h : 0};", "    return WHalo{ok ? c * CSX + (h + row0) * WPW : -1, ok ? c * nsafe + hh : 0, ok ? c : 0, ok ? h : 0};"),
                     ("    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x + nbase) + 4u * (unsigned)(wh.g + gcol));", "    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x + nbase) + 4u * (unsigned)(wh.g));")],
    # ---- lstm.hip (c4): parts of seq_head_ctc_kernel removed (timing only)
    "sh_base": [("@lstm",)],
    "sh_noctc": [("@lstm",), ("        if (wave < U && grp * U + wave < a.B) {\n            const int b = grp * U + wave;", "        if (wave < U && grp * U + wave < a.B && a.B < 0) {\n            const int b = grp * U + wave;")],
    "sh_nobwd": [("@lstm",), ("            stage(0, 0);\n            __syncthreads();\n            for (int k = 0; k < NTL; ++k) {\n                const int cur = k & 1;\n                f32x4 acc0", "            stage(0, 0);\n            __syncthreads();\n            for (int k = 0; k < NTL && a.B < 0; ++k) {\n                const int cur = k & 1;\n                f32x4 acc0")],
    "sh_nofwd": [("@lstm",), ("            for (int k = 0; k < NTL; ++k) {\n                const int cur = k & 1;\n                if (k + 1 < NTL) p0 = fetch(k + 1);", "            for (int k = 0; k < NTL && a.B < 0; ++k) {\n                const int cur = k & 1;\n                if (k + 1 < NTL) p0 = fetch(k + 1);")],
}


def build(names):
    OUT.mkdir(parents=True, exist_ok=True)
    jobs = []
    for name in names:
        lstm = any(e[0] == "@lstm" for e in EDITS[name])
        src = SRC_LSTM if lstm else SRC
        t = src.read_text()
        objs = [str(p) for p in sorted((ROOT / "build" / "obj").glob("*.o")) if p.name != ("lstm.o" if lstm else "res8.o")]
        for e in EDITS[name]:
            if e[0] == "@lstm":
                continue
            old, new = e[0], e[1]
            cnt = e[2] if len(e) > 2 else 1
            assert t.count(old) == cnt, (name, old[:90], t.count(old))
            t = t.replace(old, new)
        (OUT / "_src").mkdir(exist_ok=True)      # not next to the product sources: builders glob csrc/*.hip
        tmp = OUT / "_src" / f"{name}.hip"
        tmp.write_text(t)
        obj = OUT / f"res8_{name}.o"
        jobs.append((name, tmp, obj, objs, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w",
                                                       f"-I{SRC.parent}", f"-I{ROOT / 'include'}", "-c", str(tmp), "-o", str(obj)])))
        if len(jobs) % 4 == 0:
            for j in jobs[-4:]:
                j[4].wait()
    for name, tmp, obj, objs, proc in jobs:
        rc = proc.wait()
        assert rc == 0, name
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(OUT / f"libhowl_{name}.so"), str(obj)] + objs,
                       check=True)
        obj.unlink()
        print("built", name, flush=True)


def run(names, extra):
    libs = [(n, OUT / f"libhowl_{n}.so") for n in names]
    cfg = extra if extra else ["--config", "c1"]
    for rep in range(2):
        for name, lib in libs:
            env = dict(os.environ, HOWL_HIP_LIBRARY=str(lib), NUM_MELS=os.environ.get("NUM_MELS", "40"))
            r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--no-roofline", "--no-unfused-leg", "--steps", "200",
                                "--warmup", "20"] + cfg, env=env, capture_output=True, text=True, timeout=300)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"{name:18s} step {d['ms_per_step'] * 1e3:8.1f} us (median {d['repeats']['ms_per_step_median'] * 1e3:8.1f})", flush=True)
            except Exception:
                print(name, "FAILED", r.stderr[-600:], flush=True)


if __name__ == "__main__":
    mode, rest = sys.argv[1], sys.argv[2:]
    names = [a for a in rest if not a.startswith("-") and a in EDITS] or list(EDITS)
    extra = rest[rest.index("--config"):] if "--config" in rest else []
    build(names) if mode == "build" else run(names, extra)
