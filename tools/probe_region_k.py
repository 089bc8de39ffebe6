"""One step_n(K) call between two synchronisations, K = 1 .. 64: which part of the driver's 20-step region is per call, which
per update (needs a GPU).  usage: probe_region_k.py [precision]"""
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
for K in (1, 2, 4, 8, 16, 20, 32, 40, 64):
    out = []
    for rep in range(15):
        L.step_n(replay.handle, 5, 256, seed=0)
        t.cuda.synchronize(dev)
        t0 = time.perf_counter()
        L.step_n(replay.handle, K, 256, seed=0)
        t.cuda.synchronize(dev)
        out.append((time.perf_counter() - t0) * 1e6)
    out.sort()
    print(f"{prec} K={K:3d}: min {out[0]:7.1f} median {out[len(out) // 2]:7.1f} us  -> {out[len(out) // 2] / K:6.2f} us per update", flush=True)
