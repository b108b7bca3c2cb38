"""Remove every ``#if ... HOWL_DIAG_* ...`` branch from the kernel sources, keeping what the product build compiles (all
HOWL_DIAG_* undefined).  Round 5: the ablation variants of rounds 1-4 (tools/variants*.py, probe_*.py; results in DESIGN.md 5-5e)
lived inside the product kernels behind these macros; the sources at commit 37e3835 still carry them.

    python tools/strip_diag.py howl_amd/csrc/res8.hip howl_amd/csrc/frontend.hip
"""
import re
import sys


def cond_value(expr):
    """Value of a preprocessor condition made of defined(HOWL_DIAG_X) terms with every such macro undefined."""
    e = re.sub(r"//.*", "", expr).strip()
    e = re.sub(r"defined\s*\(\s*HOWL_DIAG_\w+\s*\)", "0", e)
    if not re.fullmatch(r"[01!|&() ]+", e):
        raise ValueError("condition mixes HOWL_DIAG with something else: " + expr)
    return bool(eval(e.replace("||", " or ").replace("&&", " and ").replace("!", " not ")))


def strip(text):
    out, stack = [], []      # stack entries: ["other"] or ["diag", taking, taken_any]
    emitting = lambda: all(s[0] == "other" or s[1] for s in stack)
    for line in text.splitlines(keepends=True):
        m = re.match(r"\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)", line)
        if not m:
            if emitting():
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            if "HOWL_DIAG_" in rest:
                if kind != "if":
                    v = kind == "ifndef"
                else:
                    v = cond_value(rest)
                stack.append(["diag", v, v])
            else:
                if emitting():
                    out.append(line)
                stack.append(["other"])
        elif kind == "elif":
            top = stack[-1]
            if top[0] == "diag":
                v = (not top[2]) and cond_value(rest)
                top[1], top[2] = v, top[2] or v
            elif emitting():
                out.append(line)
        elif kind == "else":
            top = stack[-1]
            if top[0] == "diag":
                top[1] = not top[2]
                top[2] = True
            elif emitting():
                out.append(line)
        else:
            top = stack.pop()
            if top[0] == "other" and emitting():
                out.append(line)
    assert not stack
    return "".join(out)


if __name__ == "__main__":
    for path in sys.argv[1:]:
        src = open(path).read()
        new = strip(src)
        open(path, "w").write(new)
        print(path, len(src.splitlines()), "->", len(new.splitlines()), "lines")
