"""Fixed cost of a short timed region: step_n(K) between two drained points, K = 1 .. 100 (intercept of the line)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import ctypes
import os
import numpy as np
import torch as t
if "--spin-flag" in sys.argv:      # hipDeviceScheduleSpin before the context exists: synchronize busy-waits
    hip = ctypes.CDLL(os.path.join(os.path.dirname(t.__file__), "lib", "libamdhip64.so"))
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", hip.hipSetDeviceFlags(ctypes.c_uint(1)), flush=True)
import bench

dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0)
algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, "x2")
L = algo.learner
L.step_n(replay.handle, 3000, 256, seed=0)
ev = t.cuda.Event()


def drained(spin):
    if spin:
        ev.record()
        while not ev.query():
            pass
    t.cuda.synchronize()


for spin in (False, True):
    xs, ys = [], []
    for K in (1, 2, 5, 10, 20, 50, 100):
        best = 1e9
        for rep in range(7):
            L.step_n(replay.handle, 5, 256, seed=0)
            drained(spin)
            t0 = time.perf_counter()
            L.step_n(replay.handle, K, 256, seed=0)
            drained(spin)
            best = min(best, time.perf_counter() - t0)
        xs.append(K); ys.append(best * 1e6)
    a, b = np.polyfit(xs, ys, 1)
    print(f"spin={spin}: " + "  ".join(f"K={k}: {y:.1f}us" for k, y in zip(xs, ys)) + f"   fit: {a:.2f} us/update + {b:.1f} us", flush=True)
