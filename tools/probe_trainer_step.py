"""Where a trainer step's time goes: each piece of the loop timed on its own (host call time and time to completion)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch as t
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
from oprl_amd.logging import NullLogger

prec = sys.argv[1] if len(sys.argv) > 1 else "x2"
S, A, B, N = 24, 6, 256, 3000
algo = DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device="cuda", precision=prec).create()
buf = EpisodicReplayBuffer(buffer_size_transitions=int(1e6), state_dim=S, action_dim=A, device="cuda").create()
rs = np.random.RandomState(0)
for i in range(2000):
    buf.add_transition(rs.standard_normal(S).astype(np.float32), rs.uniform(-1, 1, A), 0.1, False, episode_done=(i % 1000 == 999))
obs = rs.standard_normal(S).astype(np.float32)
act = rs.uniform(-1, 1, A)
mlp = algo._actor_mlp()


def timed(name, fn, sync=True):
    for _ in range(200):
        fn()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    host = time.perf_counter() - t0
    t.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"{name:58s} host {host / N * 1e6:7.2f} us/call   with the GPU drained {tot / N * 1e6:7.2f} us/call", flush=True)


n_add = [0]


def add():
    n_add[0] += 1
    buf.add_transition(obs, act, 0.1, False, episode_done=(n_add[0] % 1000 == 0))


timed("add_transition", add)
timed("update_from_buffer (no new transition)", lambda: algo.update_from_buffer(buf, B))
timed("hip_act (oprl_mlp_act: launch + sync)", lambda: mlp.hip_act(obs))


def two():
    add()
    algo.update_from_buffer(buf, B)
    mlp.hip_act(obs)


def ride():
    add()
    algo.update_from_buffer(buf, B, act_next=obs)
    mlp.hip_act(obs)


def ride_noadd():
    algo.update_from_buffer(buf, B, act_next=obs)
    mlp.hip_act(obs)


timed("add + update_from_buffer + hip_act (the old step)", two)
timed("add + update_from_buffer(act_next) + collect (the new step)", ride)
timed("update_from_buffer(act_next) + collect, no add", ride_noadd)
