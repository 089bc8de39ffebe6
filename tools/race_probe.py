"""Debug tool: a COLD chain launch (the first of a fresh process) against one-update launches of the same stream."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.logging import NullLogger
from tests.test_gpu_callers import _filled_buffer

K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
prec = sys.argv[2] if len(sys.argv) > 2 else "x2"
buf = _filled_buffer()


def run(env):
    for k in ("OPRL_AMD_CHAIN", "OPRL_AMD_FORM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t.manual_seed(0)
    a = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=256, precision=prec).create()
    a.learner.step_n(buf.handle, K, 256, seed=21)
    t.cuda.synchronize()
    a.learner.check()
    out = {m: getattr(a, m)._oprl_arena.clone() for m in ("actor", "critic", "actor_target", "critic_target")}
    sd = a.learner.state_dict()
    del a
    return out, sd


cold, sd_cold = run({})
one, sd_one = run({"OPRL_AMD_CHAIN": "1"})
warm, sd_warm = run({})
for name, x in (("cold chain", cold), ("warm chain", warm)):
    line = []
    for m in x:
        d = (x[m] - one[m]).abs()
        nz = int((d > 0).sum())
        idx = int(d.argmax())
        line.append(f"{m}: {nz} differ, max {float(d.max()):.3e} @ {idx}")
    print(f"K={K} {prec} {name} vs one-update launches: " + " | ".join(line), flush=True)
