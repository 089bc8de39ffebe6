"""The fused DDPG path (csrc/fused_ddpg.hip: 2 slice kernels, in-kernel gather,
actor forward overlapped) must reproduce the generic per-net launch sequence
bit for bit — same engine routines, same summation order."""
import pytest
import torch as t

from oracle import fixtures as fx

pytestmark = pytest.mark.gpu


def _ddpg(**kw):
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    t.manual_seed(0)
    return DDPG(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", **kw).create()


@pytest.mark.parametrize("B", [256, 8, 100])
def test_fused_equals_generic_bitwise(B):
    fused, generic = _ddpg(), _ddpg(no_fuse=True)
    for step in range(4):
        batch = [x.cuda() for x in fx.make_batch(70 + step, B, 24, 6)]
        fused.update(*batch)
        generic.update(*batch)
    t.cuda.synchronize()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert t.equal(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena), m
    for which in ("actor_m", "actor_v", "critic_m", "critic_v"):
        assert t.equal(getattr(fused.learner, which), getattr(generic.learner, which)), which
    qf, yf = fused.learner.debug_q_y(B)
    qg, yg = generic.learner.debug_q_y(B)
    assert t.equal(qf, qg) and t.equal(yf, yg)
    sf, sg = fused.learner.read_scalars(), generic.learner.read_scalars()
    assert sf == sg


def test_fused_step_n_equals_generic_step_n():
    from tests.test_gpu_callers import _filled_buffer
    fused, generic = _ddpg(max_batch=64), _ddpg(max_batch=64, no_fuse=True)
    buf = _filled_buffer()
    fused.learner.step_n(buf.handle, 10, 64, seed=11)
    generic.learner.step_n(buf.handle, 10, 64, seed=11)
    t.cuda.synchronize()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert t.equal(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena), m
