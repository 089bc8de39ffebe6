"""The fused DDPG path (csrc/fused_ddpg.hip: 2 slice kernels, in-kernel gather, three
concurrent roles with granule hand-off, each role a tensor-parallel cluster of CUs —
csrc/tp3.h) against the generic per-net launch sequence.  With clusters the output
layer and the first-layer gradient are sums of per-member partials, so fp32 rounding
differs from the single-CU order: compared at 1e-5 relative after 4 updates (the 1e-4
gate against the reference is checked by test_gpu_algos.py, which runs this path)."""
import pytest
import torch as t

from oracle import fixtures as fx

pytestmark = pytest.mark.gpu


def _ddpg(**kw):
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    t.manual_seed(0)
    return DDPG(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", **kw).create()


def _close(a, b, tol=1e-5):
    return (a - b).abs().max().item() <= tol * max(b.abs().max().item(), 1e-12)


# cluster "4g": clusters of 4 through the generic tp3.h passes instead of the lean tp4.h ones
@pytest.mark.parametrize("cluster", [4, "4g", 2, 1])
@pytest.mark.parametrize("B", [256, 8, 100])
def test_fused_equals_generic(B, cluster, monkeypatch):
    monkeypatch.setenv("OPRL_AMD_NO_LEAN", "1" if cluster == "4g" else "0")
    cluster = 4 if cluster == "4g" else cluster
    monkeypatch.setenv("OPRL_AMD_CLUSTER", str(cluster))
    fused, generic = _ddpg(), _ddpg(no_fuse=True)
    for step in range(4):
        batch = [x.cuda() for x in fx.make_batch(70 + step, B, 24, 6)]
        fused.update(*batch)
        generic.update(*batch)
    t.cuda.synchronize()
    assert t.isfinite(fused.actor._oprl_arena).all()     # a timed-out exchange would surface as NaN
    # parameters: 1e-4 (Adam's m/sqrt(v) amplifies summation-order noise on elements whose
    # minibatch gradient nearly cancels — tests/scenarios.py::compare); outputs below: 1e-5
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m
    for which in ("actor_m", "critic_m"):
        assert _close(getattr(fused.learner, which), getattr(generic.learner, which), 1e-4), which
    qf, yf = fused.learner.debug_q_y(B)
    qg, yg = generic.learner.debug_q_y(B)
    assert _close(qf, qg) and _close(yf, yg)
    sf, sg = fused.learner.read_scalars(), generic.learner.read_scalars()
    for k in sf:
        assert abs(sf[k] - sg[k]) <= 1e-5 * max(abs(sg[k]), 1e-12), k


def test_fused_step_n_equals_generic_step_n():
    from tests.test_gpu_callers import _filled_buffer
    fused, generic = _ddpg(max_batch=64), _ddpg(max_batch=64, no_fuse=True)
    buf = _filled_buffer()
    fused.learner.step_n(buf.handle, 10, 64, seed=11)
    generic.learner.step_n(buf.handle, 10, 64, seed=11)
    t.cuda.synchronize()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m
