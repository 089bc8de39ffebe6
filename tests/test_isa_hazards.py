"""Build-time guard for the hand-written kernels' inline-asm loads (CPU: hipcc cross-compiles, no GPU needed).

`global_load_dwordx4 ... sc1` as inline asm (how the gated dW tiles first read their rows) is invisible to hipcc's
wait-count insertion AND to its register allocator: the result registers may be copied or reused before the explicit
`s_waitcnt vmcnt(0)` in the source.  Round 3 hit exactly that (k_lw_mid_pair<.., PrecBF16>: a GPU memory fault); the
kernels now read such rows with raw buffer loads the compiler counts (engine.h ld4_agent).  The check
(tools/check_asm_loads.py) walks the device assembly of the units with cross-workgroup hand-overs: an inline-asm load
that comes back and lands in the hazard fails here instead of on the GPU box."""
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "tools"))
UNITS = ["fused_ddpg.hip", "layerwise.hip", "kernels.hip"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    return None


@pytest.mark.skipif(_hipcc() is None, reason="hipcc not available")
def test_no_instruction_touches_an_inline_asm_loads_result_before_its_wait(tmp_path):
    import check_asm_loads as chk

    def asm(unit):
        out = tmp_path / (Path(unit).stem + ".s")
        subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only", "-S",
                        unit, "-o", str(out)], check=True, cwd=str(ROOT / "oprl_amd" / "csrc"),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return out

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        outs = list(ex.map(asm, UNITS))
    for out in outs:
        n, bad = chk.check(str(out))
        assert bad == 0, f"{out.name}: {bad} uses of an inline-asm load's result registers before its s_waitcnt"
    # (since the end of round 3 the kernels hold no inline-asm load at all — the sc1 rows come in through
    # __builtin_amdgcn_raw_buffer_load_b128, engine.h ld4_agent — and the sc1 loads must still be there)
    text = "".join(o.read_text() for o in outs)
    assert text.count("buffer_load_dwordx4") > 0 and " sc1" in text


def test_no_vector_memory_instruction_is_written_as_inline_asm():
    """Round 4 (profiles/r04_experiments.txt r04-18): `global_store_dwordx4 ... sc1` as inline asm — how the role
    workgroups wrote the rows their tile workgroups read — is invisible to hipcc's hazard recogniser: a vector-memory store
    of more than 64 bits reads its data registers late and a VALU write of those registers right behind it needs a wait
    state the compiler only inserts for stores it knows.  Under memory back-pressure the first dword of such a store came
    out wrong (bit-identical runs diverged; the other half of that story — stale L1 lines on the reader's side — is r04-23).  Loads had the mirror problem (above).
    Every vector-memory access of the kernels is therefore a builtin the compiler sees; inline asm is for waits, cache
    invalidates and scheduling fences only."""
    import re
    bad = []
    for src in sorted((ROOT / "oprl_amd" / "csrc").glob("*")):
        if src.suffix not in (".h", ".hip"):
            continue
        for n, line in enumerate(src.read_text().splitlines(), 1):
            code = line.split("//")[0]
            if "asm" in code and re.search(r'"\s*(global|buffer|flat|scratch|ds)_(load|store|atomic|read|write)', code):
                bad.append(f"{src.name}:{n}: {line.strip()}")
    assert not bad, "\n".join(bad)
