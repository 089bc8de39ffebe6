"""bench.py's sequence around its timed region, the region repeated: which repetitions are slow?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench
dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0)
algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, "x2")
L = algo.learner
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
L.step_n(replay.handle, pre, 256, seed=0)
for _ in range(3):
    L.step_n(replay.handle, 20, 256, seed=0)
L.step_n(replay.handle, 5, 256, seed=0)
out = []
for rep in range(12):
    t.cuda.synchronize(dev)
    if gap:
        time.sleep(gap)
    t0 = time.perf_counter()
    L.step_n(replay.handle, 20, 256, seed=0)
    t.cuda.synchronize(dev)
    out.append((time.perf_counter() - t0) * 1e6)
print(f"pre-warm {pre}, idle gap {gap} s: " + " ".join(f"{x:.0f}" for x in out), flush=True)
