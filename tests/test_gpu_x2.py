"""The split-precision mode (OPRL_PREC_X2, csrc/engine.h PrecX2: every operand as the sum of two fp16 numbers, three
v_mfma_f32_16x16x32_f16 per product with fp32 accumulation) against the SAME golden vectors — produced by running the
reference — and the same CPU oracle as the exact-fp32 mode, at the SAME gates (tests/test_gpu_algos.py: outputs 2e-5,
parameter digests 1e-4 = north_star's gate): it is a parity mode, not a reduced-precision one.  The reference has no
such arithmetic; what pins it is that nothing here is compared with anything but the reference's own numbers."""
import functools

import numpy as np
import pytest
import torch as t

from tests import scenarios as sc
from tests import hip_adapters as ha

pytestmark = pytest.mark.gpu
TOL = 2e-5
MOMENT_TOL = 5e-3      # Adam-moment digests: one ReLU mask flip of one row (tests/scenarios.py::compare)


def _x2(cls):
    return functools.partial(cls, precision="x2")


def _both(got, name, oracle_out, skip=()):
    gold = sc.load_golden(name)
    w1 = sc.compare(got, gold, TOL, skip=skip, param_tol=sc.PARAM_TOL, moment_tol=MOMENT_TOL)
    w2 = sc.compare(got, {k: v for k, v in oracle_out.items()}, TOL, skip=skip, param_tol=sc.PARAM_TOL, moment_tol=MOMENT_TOL)
    print(f"{name} [x2]: worst vs golden {w1}, vs oracle {w2}")


def test_x2_learner_actually_runs_the_split_kernels():
    """The mode is not a label: a learner created with precision='x2' owns two-plane fp16 packs that follow the
    master weights through the updates (the fused kernels read nothing else)."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    t.manual_seed(0)
    a = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", precision="x2").create()
    b = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", precision="f32").create()
    b.actor.load_state_dict(a.actor.state_dict()); b.critic.load_state_dict(a.critic.state_dict())
    b.actor_target.load_state_dict(a.actor_target.state_dict()); b.critic_target.load_state_dict(a.critic_target.state_dict())
    from oracle import fixtures as fx
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(900 + step, 256, 24, 6)]
        a.update(*batch); b.update(*batch)
    t.cuda.synchronize()
    pa, pb = a.critic._oprl_arena, b.critic._oprl_arena
    dev = (pa - pb).abs().max().item() / pb.abs().max().item()
    assert t.isfinite(pa).all()
    assert 0.0 < dev < 1e-4, dev          # another arithmetic (not bit-identical), the same numbers


def test_x2_ddpg_walker_b256():
    got = sc.ddpg_scenario(_x2(ha.HipDDPG))
    _both(got, "ddpg_walker_b256", sc.ddpg_scenario(sc.OracleDDPG))


def test_x2_td3_cheetah_b256():
    got = sc.td3_scenario(_x2(ha.HipTD3))
    _both(got, "td3_cheetah_b256", sc.td3_scenario(sc.OracleTD3))


def test_x2_sac_humanoid_b1024():
    got = sc.sac_scenario(_x2(ha.HipSAC), "humanoid", 1024, 300, False, 2)
    _both(got, "sac_humanoid_b1024", sc.sac_scenario(sc.OracleSAC, "humanoid", 1024, 300, False, 2))


def test_x2_sac_walker_tuned_alpha():
    got = sc.sac_scenario(_x2(ha.HipSAC), "walker", 256, 350, True, 3)
    _both(got, "sac_walker_tune_b256", sc.sac_scenario(sc.OracleSAC, "walker", 256, 350, True, 3))


def test_x2_tqc_walker_b256():
    """TQC: the 512-wide critics' hidden layers — 97 % of a critic's FLOPs — through the split-fp16 packs
    (k_lw_mid_run2<*, PrecX2>, the A operand scaled per wave); the folded first layer, the heads and the 256-wide actor
    stay exact fp32.  Same golden vectors, same gates as the exact-fp32 learner (tests/test_gpu_algos.py)."""
    got = sc.tqc_scenario(_x2(ha.HipTQC))
    _both(got, "tqc_walker_b256", sc.tqc_scenario(sc.OracleTQC), skip=("qh.",))


@pytest.mark.parametrize("B", [100, 8, 1])
def test_x2_ragged_batches_against_the_oracle(B):
    cases = [("ddpg", sc.ddpg_scenario(_x2(ha.HipDDPG), B=B), sc.ddpg_scenario(sc.OracleDDPG, B=B)),
             ("td3", sc.td3_scenario(_x2(ha.HipTD3), B=B), sc.td3_scenario(sc.OracleTD3, B=B)),
             ("sac", sc.sac_scenario(_x2(ha.HipSAC), "walker", B, 350, True, 3),
              sc.sac_scenario(sc.OracleSAC, "walker", B, 350, True, 3)),
             ("tqc", sc.tqc_scenario(_x2(ha.HipTQC), B=B), sc.tqc_scenario(sc.OracleTQC, B=B))]
    for name, got, want in cases:
        worst = sc.compare(got, {k: v for k, v in want.items()}, TOL, skip=("qh.",) if name == "tqc" else (),
                           param_tol=sc.PARAM_TOL, moment_tol=MOMENT_TOL)
        print(f"{name} B={B} [x2]: worst vs oracle {worst}")


@pytest.mark.parametrize("scale", [1e-3, 1.0, 300.0])
def test_x2_holds_its_accuracy_over_input_magnitudes(scale):
    """States three orders of magnitude smaller / larger than the fixtures' N(0, 1) (the A operand goes into the MFMAs
    through fp16: a static 2^4 scale on forward activations, a seed-normalised one on gradient tiles): Q of the x2
    learner against the exact-fp32 learner after 5 updates from the same state."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    from oracle import fixtures as fx
    t.manual_seed(0)
    a = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", precision="x2").create()
    b = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", precision="f32").create()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        getattr(b, m).load_state_dict(getattr(a, m).state_dict())
    for step in range(5):
        s, ac, r, d, s2 = [x.cuda() for x in fx.make_batch(700 + step, 256, 24, 6)]
        a.update(s * scale, ac, r, d, s2 * scale); b.update(s * scale, ac, r, d, s2 * scale)
    s, ac, *_ = [x.cuda() for x in fx.make_batch(799, 256, 24, 6)]
    qa, qb = a.critic(s * scale, ac), b.critic(s * scale, ac)
    dev = (qa - qb).abs().max().item() / qb.abs().max().item()
    print(f"scale {scale}: x2 vs f32 Q deviation {dev:.2e}")
    assert np.isfinite(dev) and dev < 2e-5, dev


@pytest.mark.parametrize("scale", [1e4, 1e6])
def test_x2_out_of_range_inputs_raise_on_the_same_update(scale):
    """Beyond the split's range (|x| >= 4094 after nothing but the static 2^4 scale: engine.h PrecX2) the mode must not be
    silent: the forward stages check what they hand on (NaN products of overflowed inputs included) and report through
    the learner's error word — check() after the very update that saw the values raises, naming the mode and the way out
    (the exact-fp32 learner takes the same batch without complaint: the reference is fp32 end to end,
    nn_models.py:43-45)."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    from oracle import fixtures as fx
    t.manual_seed(0)
    a = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", precision="x2").create()
    b = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", precision="f32").create()
    s, ac, r, d, s2 = [x.cuda() for x in fx.make_batch(700, 256, 24, 6)]
    a.update(s, ac, r, d, s2)                     # in range: clean
    t.cuda.synchronize()
    a.learner.check()
    a.update(s * scale, ac, r, d, s2 * scale)
    t.cuda.synchronize()
    with pytest.raises(RuntimeError, match="split-fp16"):
        a.learner.check()
    with pytest.raises(RuntimeError, match="precision='f32'"):
        a.update(s, ac, r, d, s2)                 # ... and the learner stays in the error state until cleared
    b.update(s * scale, ac, r, d, s2 * scale)
    t.cuda.synchronize()
    b.learner.check()
    assert all(t.isfinite(getattr(b, m)._oprl_arena).all() for m in ("actor", "critic"))


@pytest.mark.parametrize("algo_name,prec", [("ddpg", "x2"), ("td3", "x2"), ("ddpg", "f32")])
def test_x2_launch_forms_agree(algo_name, prec, monkeypatch):
    """(f32: the exact-fp32 DDPG learner takes the same three forms — k_ddpg_chain<PrecF32> with 16 x 64 fp32 tiles on
    library-owned uncached mirrors of the fragment packs; "two" is then merged phase 1 + phase 2 + the actor's dW.)
    The three launch structures of a PrecX2 learner — whole updates, 20 per launch (k_ddpg_chain; DDPG), the two
    merged launches (phase 1 + the critic's tiles | phase 2 + the actor's tiles), and the plain sequence with dW launches
    of their own — are the same arithmetic cut differently (the merged forms' 16 x 64 split-product tiles against the
    16 x 32 tiles' sums: last-bits differences): parameters after 20 step_n updates within 2e-6 of each other."""
    import importlib
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer

    def run(form):
        monkeypatch.setenv("OPRL_AMD_FORM", form)
        t.manual_seed(0)
        cls = getattr(importlib.import_module(f"oprl_amd.algos.{algo_name}"), algo_name.upper())
        a = cls(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=256, precision=prec).create()
        buf = _filled_buffer(n_eps=8, L=120)
        a.learner.step_n(buf.handle, 20, 256, seed=3)
        t.cuda.synchronize()
        a.learner.check()
        return {m: getattr(a, m)._oprl_arena.clone() for m in ("actor", "critic", "actor_target", "critic_target")}

    whole = run("chain")
    two = run("two")
    plain = run("plain")
    for m in whole:
        scale = float(plain[m].abs().max())
        d1 = float((whole[m] - two[m]).abs().max()) / scale
        d2 = float((two[m] - plain[m]).abs().max()) / scale
        print(f"{algo_name} {m}: whole vs two launches {d1:.2e}, two launches vs plain {d2:.2e}")
        assert d1 < 2e-6 and d2 < 2e-6, (m, d1, d2)


@pytest.mark.parametrize("prec", ["x2", "f32", "bf16"])
def test_chain_launches_equal_one_update_launches_bitwise(prec, monkeypatch):
    """k_ddpg_chain (several updates per launch, roles of update u + 1 behind the flags of update u) against the same
    kernel with ONE update per launch (OPRL_AMD_CHAIN=1: a kernel boundary between updates): the same arithmetic, so the
    same bits — over many short calls that each start on an idle GPU, where hand-over races showed (r04-18 … r04-23:
    inline-asm stores whose data registers were reused early, and plain loads that hit stale L1 lines behind a no-op
    invalidate; one call in ~30 differed then)."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer
    buf = _filled_buffer()

    def make(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        t.manual_seed(0)
        a = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=256, precision=prec).create()
        for k in env:
            monkeypatch.delenv(k)
        return a

    ref, chain = make({"OPRL_AMD_CHAIN": "1"}), make({})
    for c in range(80):
        K = (33, 4, 7, 20)[c % 4]
        chain.learner.step_n(buf.handle, K, 256, seed=21 + c)
        t.cuda.synchronize()
        ref.learner.step_n(buf.handle, K, 256, seed=21 + c)
        t.cuda.synchronize()
        for m in ("actor", "critic", "actor_target", "critic_target"):
            assert t.equal(getattr(ref, m)._oprl_arena, getattr(chain, m)._oprl_arena), (c, K, m)
    ref.learner.check()
    chain.learner.check()


@pytest.mark.parametrize("prec", ["x2", "f32"])
def test_module_forward_after_chain_updates_reads_current_weights(prec):
    """The whole-update launches of both parity modes never touch the caller's fp32 packs (x2: they run on the fp16
    packs; exact fp32: on library-owned uncached mirrors); what reads those packs afterwards — a module's forward,
    oprl_mlp_forward — must find them rebuilt from the masters.  Module forward after 70 chain updates == a plain
    fp32 matmul chain over the master arenas, and == the same forward after an explicit sync_params()."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer
    t.manual_seed(0)
    a = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=256, precision=prec).create()
    buf = _filled_buffer()
    x = t.randn(64, 24, device="cuda")
    before = a.actor(x).clone()
    a.learner.step_n(buf.handle, 70, 256, seed=5)
    t.cuda.synchronize()
    after = a.actor(x).clone()
    assert float((after - before).abs().max()) > 1e-5            # (the updates did move the actor)
    # masters -> a plain fp32 forward
    th = a.actor._oprl_arena
    dims, off, h = [24, 256, 256, 6], 0, x
    for l in range(3):
        W = th[off:off + dims[l + 1] * dims[l]].view(dims[l + 1], dims[l]); off += dims[l + 1] * dims[l]
        b = th[off:off + dims[l + 1]]; off += dims[l + 1]
        h = h @ W.t() + b
        if l < 2:
            h = t.relu(h)
    want = t.tanh(h)                                             # (DeterministicPolicy.forward: tanh(mlp(s)))
    assert t.allclose(after, want, atol=2e-5), float((after - want).abs().max())
    a.learner.sync_params()
    t.cuda.synchronize()
    assert t.equal(a.actor(x), after)
