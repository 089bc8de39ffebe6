"""Does an idle gap / a D2H copy between two step_n chunks slow the chunk after it?"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench

dev = t.device("cuda", 0)
algo = bench._make_algo("DDPG", 24, 6, 256, {}, dev, "x2")
replay = bench.make_replay(dev, 0)
L = algo.learner
L.step_n(replay.handle, 3000, 256, seed=1)
t.cuda.synchronize()
pinned = t.empty_like(algo.actor._oprl_arena, device="cpu").pin_memory()


def between(kind):
    if kind == "nothing":
        return
    if kind.startswith("sleep"):
        time.sleep(float(kind[5:]) * 1e-3)
    elif kind == "arena.cpu()":
        algo.actor._oprl_arena.cpu()
    elif kind == "arena -> pinned, non_blocking + sync":
        pinned.copy_(algo.actor._oprl_arena, non_blocking=True)
        t.cuda.synchronize()
    elif kind == "6 x param.cpu()":
        for p in algo.actor.parameters():
            p.detach().cpu()
    elif kind == "state_dict() only":
        algo.get_policy_state_dict()
    elif kind == "flatten a state_dict taken once":
        from oprl_amd.distrib.shm import flatten_state_dict
        flatten_state_dict(SD)
    elif kind == "cat of 6 cpu tensors":
        t.cat([p.detach().reshape(-1).to(device="cpu", dtype=t.float32) for p in algo.actor.parameters()]).numpy()
    elif kind == "state_dict read-out":
        from oprl_amd.distrib.shm import flatten_state_dict
        flatten_state_dict(algo.get_policy_state_dict())


SD = algo.get_policy_state_dict()
n_sync = [0]
_orig = L.sync_params
def _counted():
    n_sync[0] += 1
    _orig()
L.sync_params = _counted
for kind in ("nothing", "state_dict() only", "flatten a state_dict taken once", "cat of 6 cpu tensors", "state_dict read-out"):
    tot = 0.0
    for c in range(20):
        t.cuda.synchronize()
        between(kind)
        a = time.perf_counter()
        L.step_n(replay.handle, 500, 256, seed=2)
        t.cuda.synchronize()
        tot += time.perf_counter() - a
    print(f"[sync_params calls so far: {n_sync[0]}] between chunks: {kind:40s} chunk of 500 takes {tot / 20 * 1e3:6.2f} ms = {tot / 20 / 500 * 1e6:6.2f} us per update", flush=True)
