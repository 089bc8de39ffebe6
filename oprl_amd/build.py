"""Builds oprl_amd/lib/liboprl_amd.so with hipcc for gfx950 (cross-compiles
without a GPU).  ``python -m oprl_amd.build [--force]``.

Every translation unit is compiled to an object of its own, in parallel, and only
when it (or a header) is newer than its object; the link step follows."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
OUT = HERE / "lib" / "liboprl_amd.so"
OBJ = HERE / "lib" / "obj"
SOURCES = ["kernels.hip", "fused_ddpg.hip", "slice_tp.hip", "layerwise.hip", "dw_wide.hip", "p2p.hip", "replay.hip", "policy_act.hip", "learner.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (expected under /opt/rocm/bin)")


def _headers() -> list[Path]:
    return list(CSRC.glob("*.h")) + [HERE.parent / "include" / "oprl_amd.h"]


def is_stale() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    deps = list(CSRC.glob("*.hip")) + _headers()
    return any(d.stat().st_mtime > t for d in deps)


def _compile(hipcc: str, src: Path, obj: Path, verbose: bool) -> None:
    cmd = [hipcc, *CFLAGS, "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not is_stale():
        return OUT
    hipcc = _hipcc()
    OBJ.mkdir(parents=True, exist_ok=True)
    h_time = max(h.stat().st_mtime for h in _headers())
    jobs = []
    for s in SOURCES:
        src, obj = CSRC / s, OBJ / (Path(s).stem + ".o")
        if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, h_time):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        for f in [ex.submit(_compile, hipcc, s, o, verbose) for s, o in jobs]:
            f.result()
    cmd = [hipcc, *LDFLAGS, *[str(OBJ / (Path(s).stem + ".o")) for s in SOURCES], "-o", str(OUT)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(OUT)
