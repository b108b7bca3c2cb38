"""Static audit of the gfx950 ISA of every kernel (no GPU needed):  python tools/isa_audit.py
Compiles howl_amd/csrc/*.hip to assembly (device side only) and lists, per kernel, the VGPR count, scratch bytes, scratch
loads/stores after the kernel's first loop header (inspect those), and `load ... s_waitcnt vmcnt(0|1)` alternations (a guarded load per loop iteration that
the compiler turned into one memory round trip per load).  These patterns cost 2 us per step in lstm_fwd, a serialised store
chain in the GEMM epilogues and several latency-bound small kernels their run time."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "howl_amd" / "csrc"


def main():
    out = Path(tempfile.mkdtemp(prefix="howl_isa_"))
    rows = []
    for src in sorted(CSRC.glob("*.hip")):
        asm = out / (src.stem + ".s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value",
                        f"-I{ROOT / 'include'}", f"-I{CSRC}", str(src), "-o", str(asm)], check=True, capture_output=True)
        text = asm.read_text()
        meta = {}
        for m in re.finditer(r"\.set (\S+)\.(num_vgpr|private_seg_size), (\d+)", text):
            meta.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
        cur, ev, depth_lines, scratch_in_loop = None, [], 0, 0
        stats = {}

        def flush():
            if cur is None:
                return
            seq = "".join(ev)
            stats[cur] = (len(re.findall(r"LW", seq)), seq.count("L"), scratch_in_loop)

        in_loop = False
        for line in text.split("\n"):
            m = re.match(r"^(_Z\S+):", line)
            if m:
                flush()
                cur, ev, scratch_in_loop, in_loop = m.group(1), [], 0, False
                continue
            t = line.strip()
            if "Loop Header" in line:
                in_loop = True
            if t.startswith(("global_load", "buffer_load")):
                ev.append("L")
            elif t.startswith("s_waitcnt vmcnt(0)") or t.startswith("s_waitcnt vmcnt(1)"):
                ev.append("W")
            elif t.startswith("scratch_") and in_loop:
                scratch_in_loop += 1
        flush()
        for k, (pairs, loads, sil) in stats.items():
            md = meta.get(k, {})
            name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            rows.append((src.name, name[:48], md.get("num_vgpr", 0), md.get("private_seg_size", 0), sil, pairs, loads))
    print(f"{'file':14s} {'kernel':48s} {'vgpr':>4s} {'scratchB':>8s} {'scr@loop':>8s} {'ld->wait':>8s} {'loads':>5s}")
    for r in sorted(rows, key=lambda r: (-r[3], -r[5])):
        flag = " <-- look" if (r[4] > 0 or (r[5] >= 3 and r[5] * 2 >= r[6])) else ""
        print(f"{r[0]:14s} {r[1]:48s} {r[2]:4d} {r[3]:8d} {r[4]:8d} {r[5]:8d} {r[6]:5d}{flag}")


if __name__ == "__main__":
    sys.exit(main())
