"""Debug tool: how often do two runs of the same update stream differ?  (DDPG B=256, step_n(500) x 4, per configuration.)"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.logging import NullLogger
from tests.test_gpu_callers import _filled_buffer

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
K = int(sys.argv[2]) if len(sys.argv) > 2 else 500
configs = [("f32 chain", "f32", {}), ("f32 chain=1", "f32", {"OPRL_AMD_CHAIN": "1"}), ("f32 two", "f32", {"OPRL_AMD_FORM": "two"}),
           ("x2 chain", "x2", {}), ("x2 one", "x2", {"OPRL_AMD_CHAIN": "1"}), ("x2 two", "x2", {"OPRL_AMD_FORM": "two"})]
only = [a for a in sys.argv[3:] if not a.startswith('-')]
NC = next((int(a.split('=')[1]) for a in sys.argv if a.startswith('--calls=')), 4)
buf = _filled_buffer()
for name, prec, env in configs:
    if only and not any(o in name for o in only):
        continue
    for k in ("OPRL_AMD_CHAIN", "OPRL_AMD_FORM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    ref = None
    bad = 0
    first_bad = []
    for r in range(reps):
        t.manual_seed(0)
        a = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=256, precision=prec).create()
        outs = []
        for c in range(NC):
            a.learner.step_n(buf.handle, K, 256, seed=21)
            if "--nosync" not in sys.argv:
                t.cuda.synchronize()
            if "--noclone" not in sys.argv or c == NC - 1:
                outs.append(t.cat([a.actor._oprl_arena, a.critic._oprl_arena]).clone())
        t.cuda.synchronize()
        a.learner.check()
        if "-v" in sys.argv: print("   rep", r, [f"{float(o.double().sum()):.9f}" for o in outs], flush=True)
        if ref is None:
            ref = outs
        else:
            eq = [bool(t.equal(x, y)) for x, y in zip(ref, outs)]
            if not all(eq):
                bad += 1
                first_bad.append(eq.index(False))
        del a
    print(f"{name:14s}: {bad} of {reps - 1} repeat runs differ from the first; first differing call: {first_bad}", flush=True)
