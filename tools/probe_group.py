"""Aggregate steps/s of a LearnerGroup (N packed DDPG learners, four launches per update for all) for a few N:
``python tools/probe_group.py 8,16,32 [f32|x2|bf16] [ddpg|td3|sac]`` (OPRL_AMD_GROUP_NC=4: exact fp32 on clusters of four as well)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.group import LearnerGroup
from oprl_amd.logging import NullLogger

dev = t.device("cuda", 0)
replay = bench.make_replay(dev, seed=7)
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"
algo = sys.argv[3] if len(sys.argv) > 3 else "ddpg"
from oprl_amd.algos.sac import SAC
from oprl_amd.algos.td3 import TD3
CLS = {"ddpg": (DDPG, {}), "td3": (TD3, dict(log_every=10 ** 9)), "sac": (SAC, dict(log_every=10 ** 9, tune_alpha=True))}[algo]
for n in [int(x) for x in sys.argv[1].split(",")]:
    algos = []
    for i in range(n):
        t.manual_seed(100 + i)
        algos.append(CLS[0](logger=NullLogger(), state_dim=24, action_dim=6, device="cuda:0", max_batch=256, precision=prec, **CLS[1]).create())
    g = LearnerGroup(algos)
    seeds = [1000 + i for i in range(n)]
    g.step_n(replay.handle, 200, 256, seeds)
    t.cuda.synchronize()
    K = 1000
    t0 = time.perf_counter()
    g.step_n(replay.handle, K, 256, seeds)
    t.cuda.synchronize()
    dt = time.perf_counter() - t0
    finite = all(bool(t.isfinite(a.critic._oprl_arena).all()) for a in algos)
    print(f"[{algo} {prec}] group of {n}: {n * K / dt:.0f} steps/s aggregate ({K / dt:.0f} group updates/s, {dt / K * 1e6:.1f} us per group update), finite={finite}", flush=True)
    g.close()
    del algos, g
