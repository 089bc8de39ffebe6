"""Two SAC learners alternating on one stream (a flaky bounded-wait expiry seen in tests): which call mix trips it?"""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch as t
from oprl_amd.algos.sac import SAC
from oprl_amd.logging import NullLogger
from tests.test_gpu_callers import _filled_buffer
mode = sys.argv[1]
prec = sys.argv[2] if len(sys.argv) > 2 else "f32"


def make():
    t.manual_seed(0)
    return SAC(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=64, precision=prec).create()


a1, a2 = make(), make()
b1, b2 = _filled_buffer(), _filled_buffer()
rs = np.random.RandomState(4)
k = -1
try:
    for k in range(300):
        obs = rs.standard_normal(24).astype(np.float32)
        if mode == "ride+plain":
            a1.update_from_buffer(b1, 64, act_next=obs); a1._actor_mlp().hip_act(obs)
            a2.update_from_buffer(b2, 64); a2._actor_mlp().hip_act(obs)
        elif mode == "plain+plain":
            a1.update_from_buffer(b1, 64); a1._actor_mlp().hip_act(obs)
            a2.update_from_buffer(b2, 64); a2._actor_mlp().hip_act(obs)
        elif mode == "noact+noact":
            a1.update_from_buffer(b1, 64)
            a2.update_from_buffer(b2, 64)
        elif mode == "ride+ride":
            a1.update_from_buffer(b1, 64, act_next=obs); a1._actor_mlp().hip_act(obs)
            a2.update_from_buffer(b2, 64, act_next=obs); a2._actor_mlp().hip_act(obs)
        a1.learner.check(); a2.learner.check()
    t.cuda.synchronize()
    a1.learner.check(); a2.learner.check()
    print(mode, prec, "ok")
except Exception as e:
    print(mode, prec, "FAILED at", k, str(e)[:200])
