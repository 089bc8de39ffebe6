"""The floor of 'enqueue one kernel, synchronise' on this box next to step_n(K)'s call and drain times (needs a GPU)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench
dev = t.device("cuda", 0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
replay = bench.make_replay(dev, 0)
algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, prec)
L = algo.learner
L.step_n(replay.handle, 3000, 256, seed=0)
x = t.zeros(64, device=dev)
def med(v):
    v = sorted(v); return v[len(v) // 2]
a, b = [], []
for _ in range(200):
    t.cuda.synchronize(dev)
    t0 = time.perf_counter(); x.add_(1.0); t1 = time.perf_counter(); t.cuda.synchronize(dev); t2 = time.perf_counter()
    a.append((t1 - t0) * 1e6); b.append((t2 - t0) * 1e6)
print(f"torch x.add_(1): call {med(a):.1f} us, call + synchronize {med(b):.1f} us")
for K in (1, 20):
    a, b = [], []
    for _ in range(60):
        L.step_n(replay.handle, 5, 256, seed=0)
        t.cuda.synchronize(dev)
        t0 = time.perf_counter(); L.step_n(replay.handle, K, 256, seed=0); t1 = time.perf_counter(); t.cuda.synchronize(dev); t2 = time.perf_counter()
        a.append((t1 - t0) * 1e6); b.append((t2 - t0) * 1e6)
    print(f"{prec} step_n({K}): call {med(a):.1f} us, call + synchronize {med(b):.1f} us")
