"""step_n's pace against the replay's shape (episodes x length) and fill level."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch as t
import bench
from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer

dev = t.device("cuda", 0)
algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, "x2")
L = algo.learner
for ep_len, n_fill in ((1000, 1_000_000), (200, 1_000_000), (200, 100_000), (200, 20_000), (1000, 20_000)):
    buf = EpisodicReplayBuffer(buffer_size_transitions=1_000_000, state_dim=24, action_dim=6, max_episode_lenth=ep_len,
                               device="cuda", seed=0).create()
    rows = np.random.RandomState(0).standard_normal((ep_len, 24 + 6 + 2)).astype(np.float32)
    rows[:, -1] = 0
    for _ in range(n_fill // ep_len):
        buf.add_transitions(rows, episode_done=True)
    L.step_n(buf.handle, 300, 256, seed=1)
    t.cuda.synchronize()
    t0 = time.perf_counter()
    L.step_n(buf.handle, 3000, 256, seed=2)
    t.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"episodes of {ep_len:4d}, {n_fill:8d} transitions ({buf.episodes_counter} episodes): {dt / 3000 * 1e6:6.2f} us per update", flush=True)
    del buf
