"""Only step_n(K = 1) calls, back to back (for rocprofv3 --kernel-trace: tools/kernel_gaps.py reads the trace)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "step_n"
dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0)
algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, "f32")
L = algo.learner
h = replay.handle
batch = replay.sample(256)
t.cuda.synchronize()
for _ in range(1500):
    if mode == "step_n":
        L.step_n(h, 1, 256, seed=0)
    else:
        algo.update(*batch)
t.cuda.synchronize()
