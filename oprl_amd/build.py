"""Builds oprl_amd/lib/liboprl_amd.so with hipcc for gfx950 (cross-compiles
without a GPU).  ``python -m oprl_amd.build [--force]``."""
from __future__ import annotations

import shutil
import subprocess
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "lib" / "liboprl_amd.so"
SOURCES = ["kernels.hip", "fused_ddpg.hip", "slice_tp.hip", "layerwise.hip", "dw_wide.hip", "p2p.hip", "replay.hip", "learner.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-ldl"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (expected under /opt/rocm/bin)")


def is_stale() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    deps = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [HERE.parent / "include" / "oprl_amd.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not is_stale():
        return OUT
    OUT.parent.mkdir(parents=True, exist_ok=True)
    cmd = [_hipcc(), *FLAGS, *[str(CSRC / s) for s in SOURCES], "-o", str(OUT)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
