"""Device-side failures are visible to the host (include/oprl_amd.h, "Device-side failures"): a bounded
cross-workgroup wait that expires poisons its result with NaN AND stores (kernel, wait site) into the
learner's host-mapped error word; the next update / step_n / read_scalars call returns OPRL_ERR_STATE with
the text.  The expiry is forced through the test hook oprl_learner_debug_expire."""
import pytest
import torch as t

from oracle import fixtures as fx
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.logging import NullLogger

pytestmark = pytest.mark.gpu
S, A, B = 24, 6, 256


def _algo(**kw):
    t.manual_seed(0)
    return DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device="cuda", max_batch=B, **kw).create()


@pytest.mark.parametrize("site,text", [(2, "TD-target hand-off"), (1, "cluster all-reduce"),
                                       (7, "gate of the dW tiles")])   # 7: the tiles riding on phase 1's launch
def test_expired_wait_is_reported_not_silent(site, text):
    algo = _algo()
    L = algo.learner
    batch = [x.cuda() for x in fx.make_batch(3, B, S, A)]
    for _ in range(3):
        algo.update(*batch)
    t.cuda.synchronize()
    L.check()                                   # clean so far
    good = L.state_dict()
    assert L.lib.oprl_learner_debug_expire(L.handle, site) == 0
    algo.update(*batch)                         # its kernels give up their wait at once
    t.cuda.synchronize()
    with pytest.raises(RuntimeError, match=text):
        algo.update(*batch)                     # reported at the next call, with kernel and site
    with pytest.raises(RuntimeError, match="expired"):
        L.read_scalars()
    with pytest.raises(RuntimeError, match="expired"):
        L.check()
    assert not bool(t.isfinite(algo.critic._oprl_arena).all())      # (and poisoned, as before)
    # recovery: hook off, error cleared, state restored from the checkpoint taken before
    assert L.lib.oprl_learner_debug_expire(L.handle, 0) == 0
    L.clear_error()
    L.load_state_dict(good)
    for _ in range(3):
        algo.update(*batch)
    t.cuda.synchronize()
    L.check()
    assert bool(t.isfinite(algo.critic._oprl_arena).all()) and bool(t.isfinite(algo.actor._oprl_arena).all())


def test_expired_hand_over_inside_the_hidden_layer_pair_launch_is_reported():
    """TQC: two hidden layers run as one launch (k_lw_mid_pair); a second-layer workgroup whose first-layer producers
    never flag their rows gives up after its bound, poisons its rows with NaN and reports (kernel, site)."""
    from oprl_amd.algos.tqc import TQC
    t.manual_seed(0)
    algo = TQC(logger=NullLogger(), state_dim=S, action_dim=A, device="cuda", max_batch=B, log_every=10 ** 9).create()
    L = algo.learner
    batch = [x.cuda() for x in fx.make_batch(3, B, S, A)]
    for _ in range(2):
        algo.update(*batch)
    t.cuda.synchronize()
    L.check()
    good = L.state_dict()
    assert L.lib.oprl_learner_debug_expire(L.handle, 8) == 0
    with pytest.raises(RuntimeError, match="hand-over inside a layer-by-layer launch"):
        algo.update(*batch)                     # (the actor phase's entry check may already see the critic phase's report)
        t.cuda.synchronize()
        algo.update(*batch)
    t.cuda.synchronize()
    assert not bool(t.isfinite(algo.critic._oprl_arena).all())
    assert L.lib.oprl_learner_debug_expire(L.handle, 0) == 0
    L.clear_error()
    L.load_state_dict(good)
    for _ in range(2):
        algo.update(*batch)
    t.cuda.synchronize()
    L.check()
    assert bool(t.isfinite(algo.critic._oprl_arena).all()) and bool(t.isfinite(algo.actor._oprl_arena).all())


@pytest.mark.parametrize("prec", ["f32", "x2"])
def test_packed_learners_and_long_runs_stay_clean(prec):
    """Two learners driven from two host threads on two streams (the default multi-seed layout), 2000
    updates each: no wait expires, every parameter stays finite.  (Their whole-update launches — 32 updates, the whole
    chip — take turns: ChipTurn in csrc/learner.hip.)"""
    import threading
    import bench
    replay = bench.make_replay(t.device("cuda", 0), seed=3)
    handle = replay.handle
    algos = [_algo(precision=prec) for _ in range(2)]
    streams = [t.cuda.Stream() for _ in range(2)]

    def run(i):
        with t.cuda.stream(streams[i]):
            for _ in range(20):
                algos[i].learner.step_n(handle, 100, B, seed=10 + i)

    ths = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    t.cuda.synchronize()
    for a in algos:
        a.learner.check()
        assert bool(t.isfinite(a.critic._oprl_arena).all()) and bool(t.isfinite(a.actor._oprl_arena).all())


def test_learners_come_and_go_in_one_process():
    """Regression (profiles/r03_experiments.txt r03-15): PrecX2 learners keep their packs and workspace in UNCACHED device
    memory; after such a block had been handed back to the runtime, later learners of the same process — exact-fp32 ones
    included — lost flag granules and their bounded waits expired.  The blocks now stay in a process-wide cache.  Here:
    learners of every kind created, stepped and destroyed in turn; no wait may expire, every result stays finite."""
    import gc
    import importlib
    import numpy as np
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer
    rs = np.random.RandomState(0)
    for rep in range(4):
        for name, prec in (("ddpg", "f32"), ("ddpg", "x2"), ("sac", "f32"), ("td3", "x2"), ("sac", "x2")):
            cls = getattr(importlib.import_module(f"oprl_amd.algos.{name}"), name.upper())
            t.manual_seed(rep)
            pair = [cls(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=64, precision=prec).create()
                    for _ in range(2)]
            bufs = [_filled_buffer(), _filled_buffer()]
            for k in range(4):
                obs = rs.standard_normal(24).astype(np.float32)
                pair[0].update_from_buffer(bufs[0], 64, act_next=obs)
                pair[0]._actor_mlp().hip_act(obs)
                pair[1].update_from_buffer(bufs[1], 64)
            t.cuda.synchronize()
            for a in pair:
                a.learner.check()
                assert bool(t.isfinite(a.critic._oprl_arena).all()), (rep, name, prec)
            del pair, bufs
            gc.collect()
