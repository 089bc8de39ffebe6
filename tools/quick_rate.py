"""Quick rate probe (needs a GPU): DDPG walker B = 256 through step_n, one learner per entry of argv
(`prec[:ENV=V[,ENV=V..]]`, e.g. `x2 x2:OPRL_AMD_CHAIN=1 x2:OPRL_AMD_FORM=two f32`): us per update over K updates, K = 20 and K = 4000,
and a finiteness + error-word check.  Environment switches are read at learner creation."""
import os
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench

dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0)
specs = [a for a in sys.argv[1:] if not a.startswith("--")] or ["x2"]
for spec in specs:
    prec, _, envs = spec.partition(":")
    kv = [e.split("=") for e in envs.split(",") if e]
    for k, v in kv:
        os.environ[k] = v
    t.manual_seed(0)
    algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, prec)
    for k, _ in kv:
        del os.environ[k]
    L = algo.learner
    L.step_n(replay.handle, 3000, 256, seed=0)
    t.cuda.synchronize()
    out = []
    for K, reps in ((20, 9), (4000, 3)):
        best = 1e9
        for _ in range(reps):
            L.step_n(replay.handle, 5, 256, seed=0)
            t.cuda.synchronize()
            t0 = time.perf_counter()
            L.step_n(replay.handle, K, 256, seed=0)
            t.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out.append(f"K={K}: {best / K * 1e6:.2f} us/update ({K / best / 1e3:.1f}k/s)")
    L.check()
    fin = all(bool(t.isfinite(getattr(algo, m)._oprl_arena).all()) for m in ("actor", "critic", "actor_target", "critic_target"))
    print(f"{spec:40s} " + "   ".join(out) + f"   finite={fin}", flush=True)
    del algo, L
