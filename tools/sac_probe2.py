"""SAC learners after DDPG f32 / x2 learners have come and gone in the same process (the flaky expiry's setting)."""
import gc
import importlib
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch as t
from oprl_amd.logging import NullLogger
from tests.test_gpu_callers import _filled_buffer
keep = "--keep" in sys.argv
sync = "--sync" in sys.argv


def run(algo_name, precision):
    def make():
        t.manual_seed(0)
        cls = getattr(importlib.import_module(f"oprl_amd.algos.{algo_name}"), algo_name.upper())
        return cls(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=64, precision=precision).create()
    a1, a2 = make(), make()
    b1, b2 = _filled_buffer(), _filled_buffer()
    rs = np.random.RandomState(4)
    for k in range(5):
        obs = rs.standard_normal(24).astype(np.float32)
        if "--noride" in sys.argv:
            a1.update_from_buffer(b1, 64)
        else:
            a1.update_from_buffer(b1, 64, act_next=obs)
        a1._actor_mlp().hip_act(obs)
        a2.update_from_buffer(b2, 64)
        a2._actor_mlp().hip_act(obs)
    if sync:
        t.cuda.synchronize()
    a1.learner.check(); a2.learner.check()
    return (a1, a2, b1, b2) if keep else None


held = []
for rep in range(6):
    combos = (("ddpg", "f32"), ("ddpg", "x2"), ("sac", "f32"), ("td3", "x2"))
    if "--f32only" in sys.argv:
        combos = (("ddpg", "f32"), ("sac", "f32"), ("td3", "f32"))
    for name, prec in combos:
        try:
            held.append(run(name, prec))
            print(rep, name, prec, "ok", flush=True)
        except Exception as e:
            print(rep, name, prec, "FAILED", str(e)[:150], flush=True)
        gc.collect()
