"""Host time of one step_n(K) call (enqueue only) and of the drained call, K = 20: where the fixed cost of a short region sits."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench

dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0)
algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, sys.argv[1] if len(sys.argv) > 1 else "x2")
L = algo.learner
L.step_n(replay.handle, 3000, 256, seed=0)
t.cuda.synchronize()
for K in (1, 20):
    enq, tot = [], []
    for rep in range(200):
        t.cuda.synchronize()
        t0 = time.perf_counter()
        L.step_n(replay.handle, K, 256, seed=0)
        t1 = time.perf_counter()
        t.cuda.synchronize()
        t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e6); tot.append((t2 - t0) * 1e6)
    enq.sort(); tot.sort()
    print(f"K={K}: enqueue median {enq[100]:.1f} us (min {enq[0]:.1f}), call + drain median {tot[100]:.1f} us (min {tot[0]:.1f}, p90 {tot[180]:.1f}, p99 {tot[197]:.1f}, max {tot[-1]:.1f})", flush=True)
    ev = t.cuda.Event()
    tot = []
    for rep in range(200):
        t.cuda.synchronize()
        t0 = time.perf_counter()
        L.step_n(replay.handle, K, 256, seed=0)
        ev.record()
        while not ev.query():
            pass
        t.cuda.synchronize()
        tot.append((time.perf_counter() - t0) * 1e6)
    tot.sort()
    print(f"K={K}: the same with an event spin before synchronize: median {tot[100]:.1f} us (min {tot[0]:.1f}, p90 {tot[180]:.1f}, p99 {tot[197]:.1f}, max {tot[-1]:.1f})", flush=True)
