"""update() throughput of all four algorithms at the BASELINE.json configs
(parity-test cases 2-4 + DDPG), fixed synthetic minibatch resident in HBM, noise
drawn on device — one Python call per update, so the fast paths are bound by the
call rate — and, second column, oprl_learner_step_n (sampling from an HBM replay +
update, K steps per call).  Not the headline bench (bench.py); numbers for DESIGN.md."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.algos.sac import SAC
from oprl_amd.algos.td3 import TD3
from oprl_amd.algos.tqc import TQC
from oprl_amd.logging import NullLogger

CASES = [("DDPG walker B=256", DDPG, 24, 6, 256, {}),
         ("TD3 cheetah B=256", TD3, 17, 6, 256, dict(log_every=10 ** 9)),
         ("SAC humanoid B=1024", SAC, 67, 21, 1024, dict(log_every=10 ** 9)),
         ("SAC walker tuned B=256", SAC, 24, 6, 256, dict(log_every=10 ** 9, tune_alpha=True)),
         ("TQC walker B=256", TQC, 24, 6, 256, dict(log_every=10 ** 9))]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
only = sys.argv[2] if len(sys.argv) > 2 else ""     # substring filter on the case name
precs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32"]   # e.g. f32,bf16
for name, cls, S, A, B, kw in [(f"{c[0]} [{p}]", *c[1:5], dict(c[5], precision=p)) for c in CASES for p in precs]:
    if only and only not in name:
        continue
    t.manual_seed(0)
    algo = cls(logger=NullLogger(), state_dim=S, action_dim=A, device="cuda", max_batch=B, **kw).create()
    batch = [t.randn(B, S, device="cuda"), t.rand(B, A, device="cuda") * 2 - 1, t.rand(B, 1, device="cuda"),
             t.zeros(B, 1, device="cuda"), t.randn(B, S, device="cuda")]
    L = algo.learner
    for _ in range(100):
        L.update(*batch)
    dt = 1e9
    for _rep in range(2):           # best of two passes (the first one still sees clock ramp-up)
        t.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            L.update(*batch)
        t.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    # step_n: device-side sampling from a replay of the same dims
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    E, LEN = 200, 1000
    buf = EpisodicReplayBuffer(buffer_size_transitions=E * LEN, state_dim=S, action_dim=A, device="cuda", seed=0).create()
    g = t.Generator(device="cuda").manual_seed(5)
    buf._tensors["states"].copy_(t.randn((E, LEN + 1, S), device="cuda", generator=g))
    buf._tensors["actions"].copy_(t.rand((E, LEN, A), device="cuda", generator=g) * 2 - 1)
    buf._tensors["rewards"].copy_(t.rand((E, LEN, 1), device="cuda", generator=g))
    buf._tensors["dones"].zero_()
    buf.ep_lens = [LEN] * E
    buf.episodes_counter = E
    buf._number_transitions = E * LEN
    buf._lens_dirty = True
    L.step_n(buf.handle, 100, B, seed=1)
    dt2 = 1e9
    for _rep in range(2):
        t.cuda.synchronize()
        t0 = time.perf_counter()
        L.step_n(buf.handle, n, B, seed=2)
        t.cuda.synchronize()
        dt2 = min(dt2, time.perf_counter() - t0)
    print(f"{name:32s} update(): {n / dt:9.1f}/s {dt / n * 1e6:7.1f} us   step_n: {n / dt2:9.1f}/s {dt2 / n * 1e6:7.1f} us", flush=True)
    del buf
