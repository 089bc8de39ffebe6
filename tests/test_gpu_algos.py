"""GPU parity of update() for the four algorithms: the HIP learner (through
its Python API -> C-ABI) against (1) the golden vectors produced by running the
reference and (2) the CPU oracle on the same scripted scenario.

Gate: 1e-4 relative (north_star); fp32 MFMA vs torch-CPU differs only in
summation order; measured output deviations (Q, pi, log pi) are ~1e-6 and are
asserted at 2e-5; parameter digests are asserted at the 1e-4 gate (see
scenarios.compare for why Adam amplifies summation noise on a few elements)."""
import numpy as np
import pytest
import torch as t

from tests import scenarios as sc
from tests import hip_adapters as ha

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _both(got, name, oracle_out, skip=(), moment_tol=None):
    gold = sc.load_golden(name)
    w1 = sc.compare(got, gold, TOL, skip=skip, param_tol=sc.PARAM_TOL, moment_tol=moment_tol)
    w2 = sc.compare(got, {k: v for k, v in oracle_out.items()}, TOL, skip=skip, param_tol=sc.PARAM_TOL, moment_tol=moment_tol)
    print(f"{name}: worst vs golden {w1}, vs oracle {w2}")


def test_ddpg_walker_b256():
    got = sc.ddpg_scenario(ha.HipDDPG)
    _both(got, "ddpg_walker_b256", sc.ddpg_scenario(sc.OracleDDPG))


def test_td3_cheetah_b256():
    got = sc.td3_scenario(ha.HipTD3)
    _both(got, "td3_cheetah_b256", sc.td3_scenario(sc.OracleTD3))


def test_sac_humanoid_b1024():
    got = sc.sac_scenario(ha.HipSAC, "humanoid", 1024, 300, False, 2)
    _both(got, "sac_humanoid_b1024", sc.sac_scenario(sc.OracleSAC, "humanoid", 1024, 300, False, 2))


def test_sac_walker_tuned_alpha():
    got = sc.sac_scenario(ha.HipSAC, "walker", 256, 350, True, 3)
    # The Adam moments (round 6: every scenario returns them; a parameter digest cannot show a gradient-scale error, Adam's
    # step being scale-invariant) sit at 4e-7 .. 1.4e-6 in every other scenario.  This one holds rows whose hidden
    # pre-activation is within rounding of zero in the actor's first update: measured (tools/probe_moments.py, MI355X box)
    # the reference's vectors vs the oracle run on ANOTHER host 3.3e-4 (actor moments only; 1e-5 on the host that generated
    # them), this learner vs the vectors 1.1e-4, vs that oracle 3.3e-4, the x2 learner 3.4e-4 / 1.1e-4 — one ReLU mask of
    # one row each way (scenarios.compare); the actor's parameters agree to 1e-5 all the same.  Gate: 1e-3.
    _both(got, "sac_walker_tune_b256", sc.sac_scenario(sc.OracleSAC, "walker", 256, 350, True, 3), moment_tol=1e-3)


def test_tqc_walker_b256():
    got = sc.tqc_scenario(ha.HipTQC)
    _both(got, "tqc_walker_b256", sc.tqc_scenario(sc.OracleTQC), skip=("qh.",))


@pytest.mark.parametrize("B", [128, 100, 8, 1])
def test_all_algos_against_the_oracle_at_ragged_batches(B):
    """B = 128 — the batch the reference's scripts really train at (trainers/base_trainer.py:28; runners/train.py:67-79
    never forwards another) — and batches that are not a multiple of the 16-row slice (and a single row): every algorithm through its
    default (fused / layer-wise) path against the oracle computed on the spot — the oracle itself is pinned
    by the golden vectors at the reference's batch sizes (tests/test_oracle_golden.py)."""
    cases = [("ddpg", sc.ddpg_scenario(ha.HipDDPG, B=B), sc.ddpg_scenario(sc.OracleDDPG, B=B), ()),
             ("td3", sc.td3_scenario(ha.HipTD3, B=B), sc.td3_scenario(sc.OracleTD3, B=B), ()),
             ("sac", sc.sac_scenario(ha.HipSAC, "walker", B, 350, True, 3),
              sc.sac_scenario(sc.OracleSAC, "walker", B, 350, True, 3), ()),
             ("tqc", sc.tqc_scenario(ha.HipTQC, B=B), sc.tqc_scenario(sc.OracleTQC, B=B), ("qh.",))]
    # (Adam moments: a ReLU mask of ONE row that falls the other way is 1 / B of a gradient element — see
    # test_sac_walker_tuned_alpha; at a handful of rows the moment gate only says "same shape, finite, same scale")
    for name, got, want, skip in cases:
        worst = sc.compare(got, {k: v for k, v in want.items()}, TOL, skip=skip, param_tol=sc.PARAM_TOL,
                           moment_tol=max(1e-3, 0.25 / B))
        print(f"{name} B={B}: worst vs oracle {worst}")


def test_reference_style_smoke_all_algos():
    """What the reference's own test does (tests/functional/test_rl_algos.py:17-31):
    batch of 8, int64 dones, next_state aliasing state, no injected noise."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.algos.sac import SAC
    from oprl_amd.algos.td3 import TD3
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger
    S, A = 24, 6
    obs = np.random.RandomState(0).standard_normal(S).astype(np.float32)
    for cls, kw in ((DDPG, {}), (TD3, {}), (SAC, {}), (SAC, dict(tune_alpha=True)), (TQC, {})):
        algo = cls(logger=NullLogger(), state_dim=S, action_dim=A, device="cuda", **kw).create()
        assert algo.actor.exploit(obs).ndim == 1
        assert algo.actor.explore(obs).ndim == 1
        bo = t.randn(8, S)
        ba = t.clamp(t.randn(8, A), -1, 1)
        br = t.randn(8, 1)
        bd = t.randint(2, (8, 1))
        before = [p.detach().cpu().clone() for p in algo.actor.parameters()]
        algo.update(bo, ba, br, bd, bo)
        algo.update(bo, ba, br, bd, bo)
        t.cuda.synchronize()
        after = [p.detach().cpu() for p in algo.actor.parameters()]
        assert all(t.isfinite(x).all() for x in after)
        assert any((a - b).abs().max() > 0 for a, b in zip(after, before))
        assert algo.get_policy_state_dict().keys() == algo.actor.state_dict().keys()


def test_ddpg_q_and_target_rows_match_oracle():
    """Per-row Q(s,a) and TD target of one critic step, fixed minibatch."""
    from oracle import fixtures as fx
    S, A, B = 24, 6, 256
    actor = fx.make_net(1, fx.actor_dims(S, A))
    critic = fx.make_net(2, fx.critic_dims(S, A))
    hip = ha.HipDDPG(S, A, actor, critic)
    ora = sc.OracleDDPG(S, A, actor, critic)
    batch = fx.make_batch(3, B, S, A)
    hip.update(*batch)
    ora.update(*batch)
    q, y = hip.algo.learner.debug_q_y(B)
    assert sc.rel_dev(q.cpu().numpy(), ora.o.last["q"].reshape(-1).numpy()) < TOL
    assert sc.rel_dev(y.cpu().numpy(), ora.o.last["y"].reshape(-1).numpy()) < TOL
    s = hip.algo.learner.read_scalars()
    assert abs(s["critic_loss"] - float(ora.o.last["critic_loss"])) < 1e-4 * abs(float(ora.o.last["critic_loss"])) + 1e-7
    assert abs(s["actor_loss"] - float(ora.o.last["actor_loss"])) < 1e-4 * abs(float(ora.o.last["actor_loss"])) + 1e-7


def test_sac_logged_scalars_are_the_reference_tags():
    """What sac.py:108-155 logs — q1 (critic 0 alone, not the twin mean), q_target, abs_q_err, critic_loss,
    loss_actor = alpha * mean(log pi) - mean(min q), log_pi, loss_alpha — from the kernels' partial sums,
    against the oracle's values of the same update."""
    from oracle import fixtures as fx
    S, A = fx.ENVS["walker"]
    B, seed = 256, 350
    actor = fx.make_net(seed + 1, fx.actor_dims(S, A, gaussian=True))
    c1, c2 = fx.make_net(seed + 2, fx.critic_dims(S, A)), fx.make_net(seed + 3, fx.critic_dims(S, A))
    hip, ora = ha.HipSAC(S, A, actor, c1, c2, True), sc.OracleSAC(S, A, actor, c1, c2, True)
    logged = {}
    hip.algo.logger.log_scalars = lambda vals, step: logged.update(vals)
    hip.algo.logger.log_scalar = lambda tag, v, step: logged.__setitem__(tag, v)
    hip.algo.log_every = 1
    log_alpha_before = float(ora.o.log_alpha)
    args = (*fx.make_batch(seed + 10, B, S, A), fx.make_noise(seed + 50, (B, A)), fx.make_noise(seed + 70, (B, A)))
    hip.update(*args)
    ora.update(*args)
    L = ora.o.last
    want = {"algo/q1": float(L["q1"].mean()), "algo/q_target": float(L["y"].mean()),
            "algo/abs_q_err": float((L["q1"] - L["y"]).mean()), "algo/critic_loss": float(L["critic_loss"]),
            "algo/log_pi": float(L["logp"].mean())}
    # loss_actor = alpha * mean(log pi) - mean(min q): logged with the temperature AFTER its step of this update
    # (one host read at the end of the update); the reference forms it with the temperature before that step
    import math
    lp_mean = float(L["logp"].mean())
    min_q_mean = math.exp(log_alpha_before) * lp_mean - float(L["actor_loss"])
    want["algo/loss_actor"] = math.exp(float(ora.o.log_alpha)) * lp_mean - min_q_mean
    assert abs(want["algo/loss_actor"] - float(L["actor_loss"])) < 2e-3
    assert set(logged) == set(want) | {"algo/alpha", "algo/loss_alpha"}
    for k, v in want.items():
        assert abs(float(logged[k]) - v) <= 2e-5 * max(abs(v), 1.0), (k, logged[k], v)
    # (the temperature loss is logged with the log_alpha AFTER its step; the reference reads it before)
    lp = float(L["logp"].mean())
    assert abs(float(logged["algo/loss_alpha"]) - (-float(ora.o.log_alpha) * (-A + lp))) < 1e-4
    assert abs(-log_alpha_before * (-A + lp) - float(logged["algo/loss_alpha"])) < 2e-2
