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
# the same sources with -DOPRL_TRACE: the in-kernel stage stamps of tools/trace_slice.py exist only there (in the
# production library they would be 10 KB of cold code inside the hot paths of a 113 KB kernel: measured, +1.4 % updates/s
# without them).  OPRL_AMD_TRACE=1 makes _capi load this one.
OUT_TRACE = HERE / "lib" / "liboprl_amd_trace.so"
OBJ_TRACE = HERE / "lib" / "obj_trace"
SOURCES = ["kernels.hip", "fused_ddpg.hip", "slice_tp.hip", "layerwise.hip", "dw_wide.hip", "p2p.hip", "replay.hip", "policy_act.hip", "learner.hip",
           "learner_create.hip", "learner_dp.hip", "learner_group.hip", "learner_misc.hip"]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (expected under /opt/rocm/bin)")


def _headers() -> list[Path]:
    return list(CSRC.glob("*.h")) + [HERE.parent / "include" / "oprl_amd.h"]


def is_stale(out: Path = OUT) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    deps = list(CSRC.glob("*.hip")) + _headers()
    return any(d.stat().st_mtime > t for d in deps)


def _compile(hipcc: str, src: Path, obj: Path, verbose: bool, extra=()) -> None:
    cmd = [hipcc, *CFLAGS, *extra, "-c", str(src), "-o", str(obj)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))


def build(force: bool = False, verbose: bool = True, trace: bool = False) -> Path:
    OUT, OBJ, extra = (OUT_TRACE, OBJ_TRACE, ("-DOPRL_TRACE",)) if trace else (globals()["OUT"], globals()["OBJ"], ())
    if not force and not is_stale(OUT):
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
        for f in [ex.submit(_compile, hipcc, s, o, verbose, extra) for s, o in jobs]:
            f.result()
    cmd = [hipcc, *LDFLAGS, *[str(OBJ / (Path(s).stem + ".o")) for s in SOURCES], "-o", str(OUT)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    if "--trace" in sys.argv or "--all" in sys.argv:
        print(build(force="--force" in sys.argv, trace=True))
