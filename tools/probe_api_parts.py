"""Where a sample() + update() pair's GPU time goes (needs a GPU): the rate of step_n(K = 1) calls issued back to back,
of sample() alone, of update() alone on a fixed batch, and of the pair — all without host syncs inside the loops."""
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
L.step_n(replay.handle, 3000, 256, seed=0)
batch = replay.sample(256)


def rate(fn, n=3000):
    for _ in range(200):
        fn()
    t.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    t.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6


h = replay.handle
for name, fn in (("step_n(K=1)", lambda: L.step_n(h, 1, 256, seed=0)), ("sample()", lambda: replay.sample(256)),
                 ("update(fixed batch)", lambda: algo.update(*batch)), ("sample() + update()", lambda: algo.update(*replay.sample(256)))):
    enq, tot = rate(fn)
    print(f"{prec} {name:24s} host enqueue {enq:6.2f} us   total {tot:6.2f} us per call", flush=True)
