"""OPRL_AMD_GRAPH_PROBE=1: K updates captured into one hipGraph and replayed, against the plain launch loop."""
import os
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
st = t.cuda.Stream(device=dev)
with t.cuda.stream(st):
    L.step_n(replay.handle, 3000, 256, seed=1)
    t.cuda.synchronize()
    for K in (200, 1000):
        t0 = time.perf_counter()
        L.step_n(replay.handle, K, 256, seed=2)
        t.cuda.synchronize()
        print(f"K={K}: {(time.perf_counter() - t0) / K * 1e6:.2f} us per update (whole call incl. capture + instantiate when probing)", flush=True)
    L.check()
