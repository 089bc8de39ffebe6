"""Build-time guard for the hand-written kernels' inline-asm loads (CPU: hipcc cross-compiles, no GPU needed).

`global_load_dwordx4 ... sc1` as inline asm (csrc/dw_body.h ld4_sc1: the gated dW tiles' rows) is invisible to hipcc's
wait-count insertion AND to its register allocator: the result registers may be copied or reused before the explicit
`s_waitcnt vmcnt(0)` in the source.  Round 3 hit exactly that (k_lw_mid_pair<.., PrecBF16>: a GPU memory fault).  The
check (tools/check_asm_loads.py) walks the device assembly of the units that use such loads; a recompile that moves the
hazard into a shipped kernel fails here instead of on the GPU box."""
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
    total = 0
    for out in outs:
        n, bad = chk.check(str(out))
        total += n
        assert bad == 0, f"{out.name}: {bad} uses of an inline-asm load's result registers before its s_waitcnt"
    assert total > 0        # (the gated tiles' sc1 loads are there: the check looks at something)
