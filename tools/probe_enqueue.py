"""Host time of step_n's launch loop per update (call return vs GPU drained), alone and beside N spinning processes."""
import multiprocessing as mp
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))


def spin(stop):
    x = 0
    while not stop.is_set():
        x += 1


if __name__ == "__main__":
    import torch as t
    import bench
    prec = sys.argv[1] if len(sys.argv) > 1 else "x2"
    dev = t.device("cuda", 0)
    replay = bench.make_replay(dev, 0)
    algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, prec)
    L = algo.learner
    L.step_n(replay.handle, 2000, 256, seed=1)
    t.cuda.synchronize()
    def throttled():
        try:
            d = dict(l.split() for l in open('/sys/fs/cgroup/cpu.stat'))
            return int(d.get('nr_throttled', 0)), int(d.get('throttled_usec', 0))
        except Exception:
            return (0, 0)
    for n_spin in (0, 12, 32):
        ctx = mp.get_context("spawn")
        stop = ctx.Event()
        procs = [ctx.Process(target=spin, args=(stop,)) for _ in range(n_spin)]
        for p in procs:
            p.start()
        time.sleep(1.0 if n_spin else 0.0)
        th0 = throttled()
        for K in (200, 2000, 10000):
            t.cuda.synchronize()
            t0 = time.perf_counter()
            L.step_n(replay.handle, K, 256, seed=2)
            t1 = time.perf_counter()
            t.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"{prec} spinners={n_spin:2d} K={K:5d}: call returned after {(t1 - t0) / K * 1e6:6.2f} us/update, "
                  f"GPU drained after {(t2 - t0) / K * 1e6:6.2f} us/update", flush=True)
        th1 = throttled()
        print(f'   cgroup cpu.stat over this block: throttled {th1[0] - th0[0]} times, {(th1[1] - th0[1]) / 1e3:.0f} ms', flush=True)
        stop.set()
        for p in procs:
            p.join()
