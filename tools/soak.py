"""Soak test of the fused paths (needs a GPU): long step_n runs of every algorithm, twice from the same
seed — parameters must stay finite (a timed-out cluster / role exchange would surface as NaN) and the two
runs must agree bit for bit (every cross-workgroup sum is taken in a fixed order)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.algos.sac import SAC
from oprl_amd.algos.td3 import TD3
from oprl_amd.algos.tqc import TQC
from oprl_amd.logging import NullLogger

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
prec = sys.argv[2] if len(sys.argv) > 2 else "x2"      # the arithmetic mode of every learner of the run
CASES = [("DDPG B=256", DDPG, 256, 300_000, {}),
         # (every call starts on an idle GPU: where hand-over races of the several-updates launch showed, r04-18)
         ("DDPG B=256, short idle-start calls", DDPG, 256, 60_000, dict(_calls=(33, 4, 7, 20, 500))), ("DDPG B=512 (over-subscribed grid)", DDPG, 512, 100_000, {}),
         ("TD3 B=256", TD3, 256, 300_000, dict(log_every=10 ** 9)),
         ("SAC B=256 tuned alpha", SAC, 256, 300_000, dict(log_every=10 ** 9, tune_alpha=True)),
         ("SAC B=1024", SAC, 1024, 50_000, dict(log_every=10 ** 9)),
         ("TQC B=256", TQC, 256, 40_000, dict(log_every=10 ** 9))]
replay = bench.make_replay(t.device("cuda"), seed=0)
ok = True
for name, cls, B, n, kw in CASES:
    n = max(1000, int(n * scale))
    sums = []
    t0 = time.perf_counter()
    for rep in range(2):
        t.manual_seed(0)
        calls = kw.get("_calls")
        algo = cls(logger=NullLogger(), state_dim=bench.S, action_dim=bench.A, device="cuda", max_batch=B, precision=prec,
                   **{k_: v_ for k_, v_ in kw.items() if not k_.startswith("_")}).create()
        done = 0
        c = 0
        while done < n:
            k = min(20_000 if calls is None else calls[c % len(calls)], n - done)
            algo.learner.step_n(replay.handle, k, B, seed=7)
            if calls is not None:
                t.cuda.synchronize()
            done += k
            c += 1
        t.cuda.synchronize()
        arenas = [algo.actor._oprl_arena, algo.critic._oprl_arena]
        finite = all(bool(t.isfinite(a).all()) for a in arenas)
        sums.append((finite, [a.clone() for a in arenas]))
        del algo
    same = all(t.equal(a, b) for a, b in zip(sums[0][1], sums[1][1]))
    good = sums[0][0] and sums[1][0] and same
    ok = ok and good
    print(f"[{prec}] {name:36s} {n:7d} updates x2  finite={sums[0][0] and sums[1][0]}  runs identical={same}  "
          f"({time.perf_counter() - t0:.1f} s)", flush=True)
print("SOAK_OK" if ok else "SOAK_FAILED")
sys.exit(0 if ok else 1)
