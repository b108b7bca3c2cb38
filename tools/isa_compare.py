"""Compare the gfx950 assembly of the kernels in two builds of one source file, function by function.

    git show <old>:howl_amd/csrc/res8.hip > /tmp/old/howl_amd/csrc/res8.hip      (+ the headers it includes, include/howl_hip.h)
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S <old tree>/howl_amd/csrc/res8.hip -o old.s
    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S howl_amd/csrc/res8.hip -o new.s
    python tools/isa_compare.py old.s new.s

Instruction streams are compared after dropping comments / directives and normalising local label and symbol names; template
arguments that only select the default (", 0>", ", false>", "<1>") are ignored when matching names.  Round 5 used it to check
that the strip instances (80 mel bins, more than 83 frames: HALO = 1 / 2 template arguments) left every 40-bin / <= 83-frame
kernel of res8.hip as it was: SAME for all of them except a kernel-argument offset in conv0's weight gradient and the head's
two kernels (one loop over the strips of an utterance added)."""
import re
import subprocess
import sys


def funcs(path):
    out, cur, body = {}, None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur, body = m.group(1), []
            out[cur] = body
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                cur = None
                continue
            t = line.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            t = re.sub(r"\s*;.*$", "", t)
            t = re.sub(r"\.LBB\d+_", ".LBBn_", t)
            t = re.sub(r"_Z\w+", "SYM", t)
            body.append(t)
    return out


def key(mangled):
    d = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
    d = d.replace("(anonymous namespace)::", "")
    d = re.sub(r"\(.*", "", d)
    return d.replace(", false>", ">").replace(", 0>", ">").replace("<1>", "").replace("void ", "")


def main():
    old, new = funcs(sys.argv[1]), funcs(sys.argv[2])
    ko, kn = {key(n): v for n, v in old.items()}, {key(n): v for n, v in new.items()}
    for k, v in ko.items():
        if k not in kn:
            print("MISSING", k)
        elif v == kn[k]:
            print("SAME   ", k, len(v))
        else:
            diff = sum(1 for a, b in zip(v, kn[k]) if a != b) + abs(len(v) - len(kn[k]))
            print("DIFF   ", k, len(v), len(kn[k]), f"({diff} lines differ)")
    for k in kn:
        if k not in ko:
            print("NEW    ", k, len(kn[k]))


if __name__ == "__main__":
    main()
