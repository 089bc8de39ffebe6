"""The learner rank's chunk loop without actors: step_n(chunk) + policy read-out per chunk (what learner_rank_loop does)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch as t
import bench
from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
from oprl_amd.distrib.shm import flatten_state_dict

dev = t.device("cuda", 0)
algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, sys.argv[1] if len(sys.argv) > 1 else "x2")
buf = EpisodicReplayBuffer(buffer_size_transitions=1_000_000, state_dim=24, action_dim=6, max_episode_lenth=200,
                           device="cuda", seed=0).create()
rows = np.random.RandomState(0).standard_normal((200, 32)).astype(np.float32)
rows[:, -1] = 0
for _ in range(100):
    buf.add_transitions(rows, episode_done=True)
L = algo.learner
L.step_n(buf.handle, 500, 256, seed=1)
t.cuda.synchronize()
for add in (False, True, 'sync-only'):
    t0 = time.perf_counter()
    te = tp = 0.0
    for c in range(40):
        a = time.perf_counter()
        L.step_n(buf.handle, 500, 256, seed=2)
        b = time.perf_counter()
        if add is True:
            for _ in range(3):
                buf.add_transitions(rows, episode_done=True)
        if add == 'sync-only':
            t.cuda.synchronize()
        else:
            flat = flatten_state_dict(algo.get_policy_state_dict())
        d = time.perf_counter()
        te += b - a
        tp += d - b
    dt = time.perf_counter() - t0
    print(f"40 chunks of 500{' + 600 new transitions per chunk' if add is True else (' (synchronize only)' if add else '')}: {dt / 20000 * 1e6:6.2f} us per update "
          f"(enqueue {te / 40 * 1e3:.1f} ms, drain + policy read-out {tp / 40 * 1e3:.1f} ms per chunk)", flush=True)
