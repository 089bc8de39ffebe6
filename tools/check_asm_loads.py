"""Static check of the device assembly for the one hazard of an inline-asm global load: hipcc does not know that the
load is asynchronous, so nothing stops it from copying or reusing the result registers before the explicit
`s_waitcnt vmcnt(0)` that follows in the source (seen once: k_lw_mid_pair<.., PrecBF16>, a memory fault — r03-37).
For every `global_load` inside an ASMSTART / ASMEND bracket, no instruction up to the next `s_waitcnt vmcnt(0)` may
name one of its result registers.  (Since r03-40 the kernels hold no such load: engine.h ld4_agent, a raw buffer load
the compiler counts.  The check stays as a guard.)

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S x.hip -o x.s && python tools/check_asm_loads.py x.s
"""
import re
import sys


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check(path, verbose=True):
    """-> (asm loads found, hazards found)"""
    lines = open(path).read().split("\n")
    func, pending, bad, nload = None, {}, 0, 0
    for i, line in enumerate(lines):
        if re.match(r"^_Z\w+:", line):
            func, pending = line.split(":")[0], {}
        t = line.strip()
        if not t or t.startswith(";"):
            continue
        if i > 0 and "ASMSTART" in lines[i - 1] and t.startswith("global_load"):
            for r in _regs(t.split(None, 1)[1].split(",")[0].strip()):
                pending[r] = i
            nload += 1
            continue
        if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
            pending = {}
            continue
        if pending:
            used = set()
            for tk in re.findall(r"v\[\d+:\d+\]|v\d+", t):
                used |= _regs(tk)
            hit = used & set(pending)
            if hit:
                bad += 1
                if verbose and bad <= 12:
                    print(f"{path}: {func}: line {i + 1}: `{t}` touches v{sorted(hit)[0]} of the asm load at line {pending[sorted(hit)[0]] + 1}")
    return nload, bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        n, b = check(p)
        print(f"{p}: {n} inline-asm loads, {b} hazards")
        rc |= 1 if b else 0
    sys.exit(rc)
