"""step_n(K) calls issued back to back without host syncs (needs a GPU): us per call and per update for K = 1, 2, 4, 8, 32 —
the fixed cost of a call as the GPU sees it (launch boundary + a launch's first update), separated from the host's."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0)
algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, prec)
L = algo.learner
h = replay.handle
L.step_n(h, 3000, 256, seed=0)
for K in (1, 2, 4, 8, 32):
    n = 2000 // K + 50
    for _ in range(50):
        L.step_n(h, K, 256, seed=0)
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        L.step_n(h, K, 256, seed=0)
    t1 = time.perf_counter()
    t.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{prec} K={K:2d}: host enqueue {(t1 - t0) / n * 1e6:7.2f} us per call, total {(t2 - t0) / n * 1e6:7.2f} us per call = {(t2 - t0) / n / K * 1e6:6.2f} us per update", flush=True)
