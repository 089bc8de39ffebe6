"""GPU tests of the callers either side of the hot path:
trainer loop end to end, fused step_n == python sample()+update() loop,
export_grads split == fused update, policy save / load interop."""
import io

import numpy as np
import pytest
import torch as t

from oracle import fixtures as fx

pytestmark = pytest.mark.gpu


def _ddpg(**kw):
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    t.manual_seed(0)
    return DDPG(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", **kw).create()


def _filled_buffer(n_eps=6, L=50, seed=3):
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    buf = EpisodicReplayBuffer(buffer_size_transitions=n_eps * L, state_dim=24, action_dim=6,
                               max_episode_lenth=L, device="cuda", seed=seed).create()
    rs = np.random.RandomState(0)
    for e in range(n_eps - 1):
        for i in range(L - e):
            buf.add_transition(rs.standard_normal(24).astype(np.float32), rs.uniform(-1, 1, 6),
                               float(rs.uniform()), False, episode_done=(i == L - e - 1))
    return buf


def test_step_n_equals_python_loop_bitwise():
    K, B = 12, 64
    a1, a2 = _ddpg(max_batch=B), _ddpg(max_batch=B)
    assert t.equal(a1.actor._oprl_arena, a2.actor._oprl_arena)
    buf = _filled_buffer()
    a1.learner.step_n(buf.handle, K, B, seed=7)
    buf.seed = 7
    for k in range(K):
        buf._sample_counter = k
        a2.update(*buf.sample(B))
    t.cuda.synchronize()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert t.equal(getattr(a1, m)._oprl_arena, getattr(a2, m)._oprl_arena), m
    assert a1.update_step == a2.update_step == K


def test_tqc_step_n_equals_python_loop_bitwise():
    """TQC's step_n — rows of update k + 1 gathered by riding workgroups of update k's launches (csrc/batch_rows.h),
    every other rider of DESIGN.md 4.5 active — against the reference's call pattern sample() + update() per step:
    the same rows (same Philox draw and index map), the same kernels: bit-identical."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger

    def make():
        t.manual_seed(0)
        return TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=64).create()

    K, B = 10, 64
    a1, a2 = make(), make()
    buf = _filled_buffer()
    a1.learner.step_n(buf.handle, K, B, seed=7)
    buf.seed = 7
    for k in range(K):
        buf._sample_counter = k
        a2.update(*buf.sample(B))
    t.cuda.synchronize()
    a1.learner.check()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(a1, m)._oprl_arena, getattr(a2, m)._oprl_arena), m


def test_sac_b1024_step_n_equals_python_loop_bitwise():
    """SAC at B = 1024: phase 2's grid fills the chip, so step_n's rows for the next update are gathered by riding
    workgroups of the actor's dW launch (batch_rows.h prefetch_rows_direct) instead of phase 2's prefetch row —
    against sample() + update() per step (k_replay_gather launches): the same rows, bit-identical."""
    from oprl_amd.algos.sac import SAC
    from oprl_amd.logging import NullLogger

    def make():
        t.manual_seed(0)
        return SAC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=1024,
                   tune_alpha=True).create()

    K, B = 8, 1024
    a1, a2 = make(), make()
    buf = _filled_buffer()
    a1.learner.step_n(buf.handle, K, B, seed=7)
    buf.seed = 7
    for k in range(K):
        buf._sample_counter = k
        a2.update(*buf.sample(B))
    t.cuda.synchronize()
    a1.learner.check()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(a1, m)._oprl_arena, getattr(a2, m)._oprl_arena), m
    assert a1.alpha == a2.alpha


def test_export_grads_split_equals_fused_update(monkeypatch):
    # (the one-call update with the actor's backward in phase 2 itself, as the split path runs it: the merged phase 2
    # — role U's unit seeds — sums in another order and is held to the generic path by tests/test_gpu_fused.py)
    monkeypatch.setenv("OPRL_AMD_FORM", "p2")
    fused, split = _ddpg(), _ddpg(export_grads=True)
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(50 + step, 256, 24, 6)]
        fused.update(*batch)
        L = split.learner
        L.update_phase(0, *batch); L.apply(0, 1.0)
        L.update_phase(1, *batch); L.apply(1, 1.0)
    t.cuda.synchronize()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        a, b = getattr(fused, m)._oprl_arena, getattr(split, m)._oprl_arena
        assert (a - b).abs().max().item() <= 2e-7 * max(1.0, a.abs().max().item()), m     # (two summation orders: an ulp or two)
    # packs were rebuilt by apply(): the module forward (reads packs) agrees with the masters
    s, a, *_ = (x.cuda() for x in fx.make_batch(99, 64, 24, 6))
    q_split = split.critic(s, a)
    split.critic.q1.mark_dirty()          # force a rebuild from the master
    assert t.equal(q_split, split.critic(s, a))


def test_td3_export_grads_split_equals_fused_update(monkeypatch):
    """TD3 through the fused twin-critic kernels in data-parallel mode: update_phase / apply (the
    k_dw_adam apply_only launch over both critics' layer tables) against the one-call update, over
    critic-only and actor steps."""
    from oprl_amd.algos.td3 import TD3
    from oprl_amd.logging import NullLogger

    def make(**kw):
        t.manual_seed(0)
        return TD3(logger=NullLogger(), state_dim=17, action_dim=6, device="cuda", max_batch=256, **kw).create()

    monkeypatch.setenv("OPRL_AMD_FORM", "p2")     # (as in the DDPG test above)
    fused, split = make(), make(export_grads=True)
    for step in range(4):
        batch = [x.cuda() for x in fx.make_batch(60 + step, 256, 17, 6)]
        noise = fx.make_noise(160 + step, (256, 6)).cuda()
        fused.update(*batch, noise=noise)
        L = split.learner
        L.update_phase(0, *batch, noise0=noise); L.apply(0, 1.0)
        L.update_phase(1, *batch, noise0=noise); L.apply(1, 1.0)
    t.cuda.synchronize()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        a, b = getattr(fused, m)._oprl_arena, getattr(split, m)._oprl_arena
        assert (a - b).abs().max().item() <= 1e-7 * max(1.0, a.abs().max().item()), m


@pytest.mark.parametrize("tune_alpha", [False, True])
def test_sac_export_grads_split_equals_fused_update(tune_alpha):
    """SAC through the fused kernels in data-parallel mode (role C's pi(s) sample is taken in phase 0
    and consumed in phase 1; the temperature gradient is exported and applied like the arenas')."""
    from oprl_amd.algos.sac import SAC
    from oprl_amd.logging import NullLogger

    def make(**kw):
        t.manual_seed(0)
        return SAC(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=256,
                   tune_alpha=tune_alpha, **kw).create()

    fused, split = make(), make(export_grads=True)
    for step in range(4):
        batch = [x.cuda() for x in fx.make_batch(60 + step, 256, 24, 6)]
        n0, n1 = fx.make_noise(160 + step, (256, 6)).cuda(), fx.make_noise(260 + step, (256, 6)).cuda()
        fused.update(*batch, noise=(n0, n1))
        L = split.learner
        L.update_phase(0, *batch, noise0=n0, noise1=n1); L.apply(0, 1.0)
        L.update_phase(1, *batch, noise0=n0, noise1=n1); L.apply(1, 1.0)
    t.cuda.synchronize()
    for m in ("actor", "critic", "critic_target"):
        a, b = getattr(fused, m)._oprl_arena, getattr(split, m)._oprl_arena
        assert (a - b).abs().max().item() <= 1e-7 * max(1.0, a.abs().max().item()), m
    if tune_alpha:
        assert abs(fused.alpha - split.alpha) <= 1e-9 * fused.alpha


def test_load_state_dict_is_picked_up_by_the_learner():
    algo = _ddpg()
    other = _ddpg()
    with t.no_grad():
        for p in other.actor.parameters():
            p.mul_(0.5)
    algo.actor.load_state_dict(other.actor.state_dict())       # bumps versions -> packs resync
    s = t.randn(32, 24, device="cuda")
    assert t.equal(algo.actor(s), other.actor(s))
    batch = [x.cuda() for x in fx.make_batch(5, 256, 24, 6)]
    algo.update(*batch)                                        # must see the loaded weights
    ref = _ddpg()
    ref.actor.load_state_dict(other.actor.state_dict())
    ref.update(*batch)
    t.cuda.synchronize()
    assert t.equal(algo.actor._oprl_arena, ref.actor._oprl_arena)


def test_policy_save_load_interop_with_plain_cpu_module():
    from oprl_amd.algos.nn_models import DeterministicPolicy
    algo = _ddpg()
    algo.update(*[x.cuda() for x in fx.make_batch(1, 128, 24, 6)])
    algo.actor.explore(np.zeros(24))        # the acting path caches a C descriptor in the module: not pickled
    buf = io.BytesIO()
    t.save(algo.actor, buf)                 # whole-module pickle, like base_trainer.py:113-120
    buf.seek(0)
    loaded = t.load(buf, weights_only=False)
    obs = np.random.RandomState(0).standard_normal(24).astype(np.float32)
    assert np.allclose(loaded.exploit(obs), algo.actor.exploit(obs), atol=1e-6)
    cpu_policy = DeterministicPolicy(24, 6, device="cpu")      # what an actor process holds
    cpu_policy.load_state_dict({k: v.cpu() for k, v in algo.get_policy_state_dict().items()})
    assert np.allclose(cpu_policy.exploit(obs), algo.actor.exploit(obs), atol=1e-5)
    assert list(algo.actor.state_dict()) == [f"mlp.nn.{i}.{w}" for i in (0, 2, 4) for w in ("weight", "bias")]


def test_trainer_end_to_end_on_synthetic_env():
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    from oprl_amd.environment import make_env
    from oprl_amd.logging import NullLogger
    from oprl_amd.trainers.base_trainer import BaseTrainer
    algo = _ddpg()
    before = algo.actor._oprl_arena.clone()
    buf = EpisodicReplayBuffer(buffer_size_transitions=2000, state_dim=24, action_dim=6,
                               max_episode_lenth=100, device="cuda").create()
    env = lambda seed: __import__("oprl_amd.environment.synthetic", fromlist=["SyntheticEnv"]).SyntheticEnv(
        "walker-walk", seed=seed, episode_length=100)
    BaseTrainer(logger=NullLogger("/tmp/oprl_amd_test"), env=env(0), make_env_test=env, replay_buffer=buf,
                algo=algo, num_steps=400, start_steps=150, batch_size=32, eval_interval=250,
                num_eval_episodes=1, save_policy_every=0, stdout_log_every=10 ** 9).train()
    t.cuda.synchronize()
    assert algo.update_step == 401 - 31
    assert t.isfinite(algo.actor._oprl_arena).all() and not t.equal(before, algo.actor._oprl_arena)
    assert len(buf) == 401 and buf.episodes_counter == 5
    tr = BaseTrainer(logger=NullLogger("/tmp/oprl_amd_test"), env=env(0), make_env_test=env, replay_buffer=buf,
                     algo=algo, num_steps=1, estimate_q_every=1)
    q_true, q_critic = tr.estimate_true_q(2), tr.estimate_critic_q(2)
    assert np.isfinite(q_true) and np.isfinite(q_critic)
    tr._estimate_q(1)


def test_policy_act_equals_batched_forward():
    """oprl_mlp_act (one observation, host in / host out) runs the same slice kernel as the batched
    forward: exploit() = row 0 of forward(), for both policy classes."""
    from oprl_amd.algos.sac import SAC
    from oprl_amd.logging import NullLogger
    rs = np.random.RandomState(3)
    obs = rs.standard_normal(24)            # float64, like a gym observation
    d = _ddpg()
    want = d.actor(t.as_tensor(obs, dtype=t.float32, device="cuda").unsqueeze(0)).cpu().numpy()[0]
    assert np.array_equal(d.actor.exploit(obs), want)
    a = d.actor.explore(obs)
    assert a.shape == (6,) and np.all(np.abs(a) <= 1.0)
    s = SAC(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda").create()
    s.actor.eval()
    want, _ = s.actor(t.as_tensor(obs, dtype=t.float32, device="cuda").unsqueeze(0))
    s.actor.train()
    assert np.array_equal(s.actor.exploit(obs), want.cpu().numpy()[0])
    t.manual_seed(5)
    a1 = s.actor.explore(obs)
    t.manual_seed(5)
    a2 = s.actor.explore(obs)
    assert a1.shape == (6,) and np.array_equal(a1, a2) and not np.array_equal(a1, s.actor.exploit(obs))


def test_gaussian_actor_pickles_after_acting():
    from oprl_amd.algos.sac import SAC
    from oprl_amd.logging import NullLogger
    s = SAC(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda").create()
    obs = np.random.RandomState(1).standard_normal(24)
    want = s.actor.exploit(obs)
    buf = io.BytesIO()
    t.save(s.actor, buf)
    buf.seek(0)
    loaded = t.load(buf, weights_only=False)
    assert np.allclose(loaded.exploit(obs), want, atol=1e-6)


def test_update_from_buffer_is_step_n_of_one():
    """The trainer's fused sample+update call = oprl_learner_step_n with K = 1 and the buffer's seed,
    and = the python loop sample(inds from the same Philox draw) + update (checked by
    test_step_n_equals_python_loop_bitwise for step_n)."""
    B = 64
    a1, a2 = _ddpg(max_batch=B), _ddpg(max_batch=B)
    b1, b2 = _filled_buffer(), _filled_buffer()
    for _ in range(5):
        a1.update_from_buffer(b1, B)
    a2.learner.step_n(b2.handle, 5, B, seed=b2.seed)
    t.cuda.synchronize()
    assert a1.update_step == a2.update_step == 5
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert t.equal(getattr(a1, m)._oprl_arena, getattr(a2, m)._oprl_arena), m


@pytest.mark.parametrize("algo_name,precision", [("ddpg", "f32"), ("ddpg", "x2"), ("sac", "f32"), ("td3", "x2")])
def test_step_act_is_update_then_policy_call(algo_name, precision):
    """oprl_learner_step_act (the trainer's step: one update and, behind it in the same call, the actor's forward of the
    next observation; the row collected from host-mapped memory) against the two calls it replaces:
    update_from_buffer, then actor.explore's oprl_mlp_act — same parameters afterwards, the same raw output row."""
    import importlib
    from oprl_amd.logging import NullLogger

    def make():
        t.manual_seed(0)
        cls = getattr(importlib.import_module(f"oprl_amd.algos.{algo_name}"), algo_name.upper())
        return cls(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=64, precision=precision).create()

    a1, a2 = make(), make()
    b1, b2 = _filled_buffer(), _filled_buffer()
    rs = np.random.RandomState(4)
    for k in range(5):
        obs = rs.standard_normal(24).astype(np.float32)
        a1.update_from_buffer(b1, 64, act_next=obs)
        mlp1 = a1._actor_mlp()
        got = mlp1.hip_act(obs)                       # collects the pending row
        assert mlp1._pending is None
        a2.update_from_buffer(b2, 64)
        want = a2._actor_mlp().hip_act(obs)           # oprl_mlp_act: the MFMA kernels over the fp32 packs
        assert got.shape == want.shape
        assert sc_rel(got, want) < 2e-6, (k, sc_rel(got, want))
    t.cuda.synchronize()
    for m in ("actor", "critic"):
        assert t.equal(getattr(a1, m)._oprl_arena, getattr(a2, m)._oprl_arena), m
    # a pending row for another observation is dropped, not returned
    obs, other = rs.standard_normal(24).astype(np.float32), rs.standard_normal(24).astype(np.float32)
    a1.update_from_buffer(b1, 64, act_next=obs)
    a2.update_from_buffer(b2, 64)
    assert sc_rel(a1._actor_mlp().hip_act(other), a2._actor_mlp().hip_act(other)) < 2e-6


@pytest.mark.parametrize("algo_name,S,A", [("sac", 67, 21), ("ddpg", 67, 21), ("tqc", 24, 6)])
def test_step_act_other_shapes(algo_name, S, A):
    """The policy kernel's other forms: a net input wider than 64 (two register chunks per row), an output row wider
    than 16 (humanoid's Gaussian head, 42: the layer-by-layer kernel), TQC's actor behind the generic update sequence."""
    import importlib
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    from oprl_amd.logging import NullLogger
    t.manual_seed(0)
    cls = getattr(importlib.import_module(f"oprl_amd.algos.{algo_name}"), algo_name.upper())
    a1 = cls(logger=NullLogger(), state_dim=S, action_dim=A, device="cuda", max_batch=64).create()
    buf = EpisodicReplayBuffer(buffer_size_transitions=600, state_dim=S, action_dim=A, max_episode_lenth=50,
                               device="cuda", seed=3).create()
    rs = np.random.RandomState(0)
    for e in range(5):
        for i in range(50):
            buf.add_transition(rs.standard_normal(S).astype(np.float32), rs.uniform(-1, 1, A), float(rs.uniform()), False,
                               episode_done=(i == 49))
    for k in range(3):
        obs = rs.standard_normal(S).astype(np.float32)
        a1.update_from_buffer(buf, 64, act_next=obs)
        mlp = a1._actor_mlp()
        got = mlp.hip_act(obs)
        want = mlp.hip_act(obs)               # nothing pending now: oprl_mlp_act on the same weights
        assert got.shape == want.shape == (mlp.dims[-1],)
        assert sc_rel(got, want) < 2e-6, (k, sc_rel(got, want))


def sc_rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


def test_trainer_with_the_policy_call_riding_on_the_update_equals_the_plain_loop():
    """BaseTrainer: the actions taken, the transitions stored and the parameters reached are the same whether the next
    step's explore() collects the row that rode on the update or makes its own call (exploration noise drawn from the
    same torch generator state in both runs)."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    from oprl_amd.environment.synthetic import SyntheticEnv
    from oprl_amd.logging import NullLogger
    from oprl_amd.trainers.base_trainer import BaseTrainer

    def run(ride):
        t.manual_seed(0)
        algo = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=32).create()
        if not ride:
            algo._actor_mlp = lambda: None
        buf = EpisodicReplayBuffer(buffer_size_transitions=4000, state_dim=24, action_dim=6, max_episode_lenth=40,
                                   device="cuda", seed=1).create()
        env = SyntheticEnv("walker-walk", seed=0, episode_length=40)
        tr = BaseTrainer(logger=NullLogger("/tmp/oprl_amd_test"), env=env,
                         make_env_test=lambda s: SyntheticEnv("walker-walk", seed=s, episode_length=40),
                         replay_buffer=buf, algo=algo, num_steps=200, start_steps=60, batch_size=32,
                         eval_interval=10 ** 9, save_policy_every=0, stdout_log_every=10 ** 9)
        np.random.seed(0)
        t.manual_seed(1)
        tr.train()
        t.cuda.synchronize()
        return algo, buf

    a1, b1 = run(True)
    a2, b2 = run(False)
    assert a1.update_step == a2.update_step > 100
    for name in ("states", "actions", "rewards"):
        x, y = b1._tensors[name], b2._tensors[name]
        assert float((x - y).abs().max()) < 1e-5, name
    for m in ("actor", "critic"):
        d = float((getattr(a1, m)._oprl_arena - getattr(a2, m)._oprl_arena).abs().max())
        assert d < 1e-4, (m, d)


def test_distributed_learner_with_in_host_actors():
    """Two CPU actor processes feed the GPU learner through the in-host queues."""
    from oprl_amd.algos.nn_models import DeterministicPolicy
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    from oprl_amd.environment.synthetic import SyntheticEnv
    from oprl_amd.logging import NullLogger
    from oprl_amd.runners.config import DistribConfig
    from oprl_amd.runners.train_distrib import run_distrib_training
    from tests import test_gpu_callers as me
    cfg = DistribConfig(batch_size=32, num_env_workers=2, episodes_per_worker=4, warmup_epochs=0,
                        episode_length=40, learner_num_waits=20, warmup_env_steps=40)
    run_distrib_training(make_env=me._mk_env, make_algo=me._mk_algo, make_policy=me._mk_policy,
                         make_replay_buffer=me._mk_buffer, make_logger=me._mk_logger, config=cfg,
                         max_epochs=4)


def test_data_parallel_training_with_ring_actors_on_one_rank_over_rccl():
    """BASELINE.json config 5 end to end on this GPU (runners/train_distrib.py::run_dp_training, the --gpus N layout
    with N = 1): two CPU actor processes feed shared-memory rings, the learner rank runs the DATA-PARALLEL step —
    update_phase / ncclAllReduce / apply on a world-size-1 RCCL communicator (LearnerPlan.force_exchange), not the
    fused single-GPU update — drains the rings between chunks, publishes the policy, and stops the actors."""
    from oprl_amd.distrib.dp_learner import LearnerPlan
    from oprl_amd.runners.config import DistribConfig
    from oprl_amd.runners.train_distrib import run_dp_training
    from tests import test_gpu_callers as me
    cfg = DistribConfig(batch_size=64, num_env_workers=2, episodes_per_worker=10 ** 6, warmup_epochs=0,
                        episode_length=40, warmup_env_steps=40)
    plan = LearnerPlan(total_updates=600, batch_size=64, chunk=100, warmup_transitions=128,
                       updates_per_transition=4.0, force_exchange=True, wall_timeout_s=240.0, idle_timeout_s=30.0)
    stats = run_dp_training(make_env=me._mk_env, make_algo=me._mk_algo_kw, make_policy=me._mk_policy,
                            make_replay_buffer=me._mk_buffer_kw, make_logger=me._mk_logger, config=cfg,
                            learners=1, plan=plan, backend="nccl")
    assert len(stats) == 1
    s0 = stats[0]
    assert s0["updates"] == 600 and s0["stopped_early"] is None, s0
    assert s0["received"] >= 150 and s0["chunks"] >= 6, s0
    assert s0["replica_spread"] == 0.0 and s0["policy_version"] >= 6, s0


def _mk_algo_kw(logger, **kw):
    from oprl_amd.algos.ddpg import DDPG
    return DDPG(logger=logger, state_dim=24, action_dim=6, max_batch=64, **{"device": "cuda", **kw}).create()


def _mk_buffer_kw(**kw):
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    return EpisodicReplayBuffer(buffer_size_transitions=40000, state_dim=24, action_dim=6,
                                max_episode_lenth=40, **{"device": "cuda", **kw}).create()


def _mk_env(seed=0):
    from oprl_amd.environment.synthetic import SyntheticEnv
    return SyntheticEnv("walker-walk", seed=seed, episode_length=40)


def _mk_policy():
    from oprl_amd.algos.nn_models import DeterministicPolicy
    return DeterministicPolicy(24, 6, device="cpu")


def _mk_algo(logger):
    from oprl_amd.algos.ddpg import DDPG
    return DDPG(logger=logger, state_dim=24, action_dim=6, device="cuda").create()


def _mk_buffer():
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    return EpisodicReplayBuffer(buffer_size_transitions=4000, state_dim=24, action_dim=6,
                                max_episode_lenth=40, device="cuda").create()


def _mk_logger():
    from oprl_amd.logging import NullLogger
    return NullLogger("/tmp/oprl_amd_test")


@pytest.mark.parametrize("precision", ["f32", "x2"])
def test_native_dp_single_rank_equals_export_split(precision):
    """World-size-1 RCCL communicator on this GPU: the C data-parallel loop
    (dp_step_n: phase -> ncclAllReduce -> apply, twice per update) must reproduce the
    python-driven export_grads split bit for bit.  (x2: the phases are the merged launches whose tiles leave dW in the
    gradient arena; and the result must be the single-GPU update's within the parity gate.)"""
    import os
    import tempfile
    import torch.distributed as dist
    from oprl_amd.parallel import DataParallelLearner
    created = False
    if not dist.is_initialized():
        f = tempfile.NamedTemporaryFile(delete=False)
        dist.init_process_group("nccl", init_method=f"file://{f.name}", rank=0, world_size=1,
                                device_id=t.device("cuda", 0))
        created = True
    try:
        K, B = 6, 64
        a1, a2 = _ddpg(max_batch=B, export_grads=True, precision=precision), _ddpg(max_batch=B, export_grads=True, precision=precision)
        buf = _filled_buffer()
        dp = DataParallelLearner(a1)
        dp.init_native_comm()
        # the library's own broadcast (oprl_comm_broadcast_params: ncclBroadcast of every arena, packs rebuilt);
        # with one rank it must leave the replica exactly as it is
        before = [x.clone() for x in dp._state_tensors()]
        dp.broadcast_parameters(src=0)
        t.cuda.synchronize()
        assert all(t.equal(a, b) for a, b in zip(before, dp._state_tensors()))
        dp.step_n(buf.handle, K, B, seed=5)
        # python-driven reference: same shard key as the C loop derives for rank 0
        buf.seed = (5 * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
        L = a2.learner
        for k in range(K):
            buf._sample_counter = k
            batch = buf.sample(B)
            L.update_phase(0, *batch); L.apply(0, 1.0)
            L.update_phase(1, *batch); L.apply(1, 1.0)
        t.cuda.synchronize()
        for m in ("actor", "critic", "actor_target", "critic_target"):
            assert t.equal(getattr(a1, m)._oprl_arena, getattr(a2, m)._oprl_arena), m
        assert float(dp.replica_checksum().abs().max()) == 0
        # ... and against the single-GPU fused update on the same rows (another launch structure: tolerance)
        a3 = _ddpg(max_batch=B, precision=precision)
        buf3 = _filled_buffer()
        a3.learner.step_n(buf3.handle, K, B, seed=buf.seed)
        t.cuda.synchronize()
        for m in ("actor", "critic"):
            x, y = getattr(a1, m)._oprl_arena, getattr(a3, m)._oprl_arena
            assert float((x - y).abs().max() / y.abs().max()) < 2e-5, m
    finally:
        if created:
            dist.destroy_process_group()


def test_export_split_in_the_x2_mode_matches_the_fp32_split_for_td3():
    """The data-parallel phases of an x2 TD3 learner (twin critics: un-merged phase 1 + an exporting dW launch; the actor
    step every other update as the merged phase-2 launch whose tiles leave dW in the gradient arena) against the same
    split in exact fp32, same rows, same device noise keys: parameters within the parity gate after six updates."""
    from oprl_amd.algos.td3 import TD3
    from oprl_amd.logging import NullLogger

    def make(prec):
        t.manual_seed(0)
        return TD3(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=64,
                   export_grads=True, precision=prec, log_every=10 ** 9).create()
    a32, ax2 = make("f32"), make("x2")
    buf = _filled_buffer()
    for k in range(6):
        buf._sample_counter = k
        batch = buf.sample(64)
        for a in (a32, ax2):
            L = a.learner
            L.update_phase(0, *batch); L.apply(0, 1.0)
            L.update_phase(1, *batch); L.apply(1, 1.0)
    t.cuda.synchronize()
    for a in (a32, ax2):
        a.learner.check()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        x, y = getattr(ax2, m)._oprl_arena, getattr(a32, m)._oprl_arena
        dev = float((x - y).abs().max() / y.abs().max())
        assert dev < 1e-4, (m, dev)


@pytest.mark.parametrize("algo,precision", [("ddpg", "f32"), ("td3", "f32"), ("sac", "f32"), ("tqc", "f32"),
                                            ("ddpg", "bf16"), ("tqc", "bf16"), ("sac", "bf16"),
                                            ("ddpg", "x2"), ("td3", "x2"), ("sac", "x2"), ("tqc", "x2")])
def test_checkpoint_resume_is_bit_exact(algo, precision, tmp_path):
    """learner state (theta, targets, Adam moments, temperature, counters) + replay
    (storage, write position, sample counter) saved mid-run; a fresh process-like restore must
    continue bit for bit like the uninterrupted run — in every arithmetic mode (the bf16 / split-fp16 weight packs
    are not part of a checkpoint: they are rebuilt from the restored masters and must come out the same)."""
    from oprl_amd.logging import NullLogger

    def make():
        import importlib
        t.manual_seed(0)
        cls = getattr(importlib.import_module(f"oprl_amd.algos.{algo}"), algo.upper())
        extra = {"tune_alpha": True} if algo == "sac" else {}
        return cls(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=64, precision=precision,
                   **extra).create()

    K, B = 7, 64
    a1, b1 = make(), _filled_buffer()
    a1.learner.step_n(b1.handle, K, B, seed=5)
    path = tmp_path / "ckpt.pt"
    t.save({"algo": a1.state_dict(), "replay": b1.state_dict()}, path)
    a1.learner.step_n(b1.handle, K, B, seed=5)          # the uninterrupted run goes on

    a2, b2 = make(), _filled_buffer(seed=99)            # different replay seed until restored
    ck = t.load(path, weights_only=False)
    a2.load_state_dict(ck["algo"])
    b2.load_state_dict(ck["replay"])
    assert a2.update_step == K
    a2.learner.step_n(b2.handle, K, B, seed=5)
    t.cuda.synchronize()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(a1, m)._oprl_arena, getattr(a2, m)._oprl_arena), m
    assert t.equal(a1.learner.actor_m, a2.learner.actor_m) and t.equal(a1.learner.critic_v, a2.learner.critic_v)
    if algo in ("sac", "tqc") and a1.learner.log_alpha is not None:
        assert t.equal(a1.learner.log_alpha, a2.learner.log_alpha)
    # and the replay's next sample is the same draw
    s1, s2 = b1.sample(8), b2.sample(8)
    for x, y in zip(s1, s2):
        assert t.equal(x, y)


def test_policy_io_matches_the_reference_vectors_on_the_hip_path(monkeypatch):
    """G6 (tests/golden/policy_io.npz, generated by running the reference's DeterministicPolicy /
    GaussianActor): exploit / explore of both policy classes through oprl_mlp_act, with the reference's
    noise draws injected — including the reference's quirk that DeterministicPolicy.explore applies NO
    tanh (nn_models.py:144-150) and GaussianActor.explore samples in train mode and returns tanh(mean)
    in eval mode (:180-195)."""
    from oracle import fixtures as fx
    from tests import hip_adapters as ha
    from tests import scenarios as sc
    from oprl_amd.algos.sac import SAC
    from oprl_amd.logging import NullLogger
    gold = sc.load_golden("policy_io")
    S, A, seed = (int(x) for x in gold["meta"])
    obs = np.random.RandomState(seed + 3).standard_normal(S).astype(np.float32)
    d = _ddpg()
    ha.load_params(d.actor, fx.make_net(seed + 1, fx.actor_dims(S, A)))
    assert sc.rel_dev(d.actor.exploit(obs), gold["det.exploit"]) < 2e-6
    real_randn = t.randn
    monkeypatch.setattr(t, "randn", lambda *a, **k: fx.make_noise(seed + 4, (A,)))
    raw_explore = d.actor.explore(obs)
    monkeypatch.setattr(t, "randn", real_randn)
    assert sc.rel_dev(raw_explore, gold["det.explore"]) < 2e-6
    # the quirk is observable in the vector: tanh would have changed it
    assert sc.rel_dev(np.tanh(raw_explore), gold["det.explore"]) > 1e-3
    s = SAC(logger=NullLogger(), state_dim=S, action_dim=A, device="cuda").create()
    ha.load_params(s.actor, fx.make_net(seed + 2, fx.actor_dims(S, A, gaussian=True)))
    assert sc.rel_dev(s.actor.exploit(obs), gold["ga.exploit"]) < 2e-6
    s.actor.train()
    monkeypatch.setattr(t, "randn", lambda *a, **k: fx.make_noise(seed + 5, (1, A))[0])
    sampled = s.actor.explore(obs)
    monkeypatch.setattr(t, "randn", real_randn)
    assert sc.rel_dev(sampled, gold["ga.explore"]) < 2e-6
    s.actor.eval()
    assert sc.rel_dev(s.actor.explore(obs), gold["ga.exploit"]) < 2e-6      # eval mode: tanh(mean)
    s.actor.train()
    assert list(d.actor.state_dict().keys()) == list(gold["det.keys"])
    assert list(s.actor.state_dict().keys()) == list(gold["ga.keys"])


def test_device_noise_stream_is_standard_normal_and_keyed_by_seed_rank_counter():
    """The N(0,1) draws the TD3 / SAC / TQC kernels take on device when no noise is injected (Philox4x32-10 +
    Box-Muller, csrc/philox.h), exported through oprl_debug_noise: moments, a Kolmogorov-Smirnov test against
    the normal CDF, no correlation between neighbouring rows / columns / counters, and distinct streams for
    distinct (seed, rank, stream, counter)."""
    from scipy import stats
    from oprl_amd import _capi
    d = _ddpg()
    L = d.learner
    rows, cols = 4096, 64

    def draw(stream_id, counter):
        out = t.empty((rows, cols), dtype=t.float32, device="cuda")
        _capi.check(L.lib.oprl_debug_noise(L.handle, stream_id, counter, rows, cols, _capi.ptr(out), _capi.current_stream()))
        t.cuda.synchronize()
        return out.cpu().numpy().astype(np.float64)

    x = draw(1, 0)
    n = x.size
    assert abs(x.mean()) < 4 / np.sqrt(n) and abs(x.var() - 1) < 4 * np.sqrt(2 / n)
    assert abs(stats.skew(x.ravel())) < 4 * np.sqrt(6 / n) and abs(stats.kurtosis(x.ravel())) < 4 * np.sqrt(24 / n)
    ks = stats.kstest(x.ravel(), "norm")
    assert ks.pvalue > 1e-3, ks
    assert abs(np.abs(x).max() - 5.2) < 1.0          # 24-bit uniforms: the tail reaches ~5.6 sigma, no infinities
    for a, b in ((x[:-1], x[1:]), (x[:, :-1], x[:, 1:]), (x, draw(1, 1)), (x, draw(2, 0))):
        r = np.corrcoef(a.ravel(), b.ravel())[0, 1]
        assert abs(r) < 4 / np.sqrt(a.size), r
    assert np.array_equal(x, draw(1, 0))              # a pure function of (key, counter, row, col)
    L.set_seed(7, 0)
    y = draw(1, 0)
    L.set_seed(7, 3)
    z = draw(1, 0)
    assert not np.array_equal(x, y) and not np.array_equal(y, z)
    assert abs(np.corrcoef(x.ravel(), y.ravel())[0, 1]) < 4 / np.sqrt(n)
    assert abs(np.corrcoef(y.ravel(), z.ravel())[0, 1]) < 4 / np.sqrt(n)


@pytest.mark.parametrize("precision", ["f32", "x2"])
def test_packed_learner_group_members_against_the_cpu_oracle(precision):
    """N3 against the ORACLE (not only against the same kernels run solo): three members with the oracle's
    fixture weights, stepped as a group over the replay; the oracle's DDPG (oracle/oprl_oracle.py, pinned to the
    reference by tests/test_oracle_golden.py) is fed the same rows — the buffer's own Philox draw for each member's
    seed — and every member must end within the parity gate of its oracle twin."""
    from oracle import fixtures as fx
    from oracle import oprl_oracle as orc
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.group import LearnerGroup
    from oprl_amd.logging import NullLogger
    from tests import hip_adapters as ha
    from tests import scenarios as sc
    S, A, B, K = 24, 6, 64, 6
    buf = _filled_buffer()
    seeds = [21, 22, 23]
    nets = [(fx.make_net(300 + 2 * i, fx.actor_dims(S, A)), fx.make_net(301 + 2 * i, fx.critic_dims(S, A))) for i in range(3)]
    members = []
    for actor, critic in nets:
        m = DDPG(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=S, action_dim=A, device="cuda", max_batch=B,
                 precision=precision).create()
        for mod, p in ((m.actor, actor), (m.actor_target, actor), (m.critic, critic), (m.critic_target, critic)):
            ha.load_params(mod, p)
        members.append(m)
    g = LearnerGroup(members)
    g.step_n(buf.handle, K, B, seeds)
    t.cuda.synchronize()
    worst = 0.0
    for (actor, critic), m, seed in zip(nets, members, seeds):
        o = orc.DDPGOracle(S, A, actor, critic)
        buf.seed = seed
        for k in range(K):
            buf._sample_counter = k
            o.update(*[x.cpu() for x in buf.sample(B)])
        for name in ("actor", "critic", "actor_target", "critic_target"):
            for got, want in zip(ha.cpu_params(getattr(m, name)), getattr(o, name)):
                dev = sc.rel_dev(got, want)
                worst = max(worst, dev)
                assert dev < sc.PARAM_TOL, (name, dev)     # (north_star's parameter gate, per tensor in max-norm)
        m.learner.check()
    print(f"group members vs oracle after {K} updates: worst relative deviation {worst:.2e}")
    g.close()


@pytest.mark.parametrize("precision", ["f32", "x2"])
def test_packed_learner_group_ragged_batch_and_checkpoint(precision):
    """Eight members at a ragged batch (B = 100: seven slices, the last one partial — the member -> XCD deal of the group
    launches with a slice count that is not a multiple of eight), a member restored from its own checkpoint in between:
    still bit-identical to the solo twins."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.group import LearnerGroup
    from oprl_amd.logging import NullLogger
    B, n = 100, 8
    buf = _filled_buffer()

    def member(i):
        t.manual_seed(140 + i)
        return DDPG(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=128,
                    precision=precision).create()
    members, solo = [member(i) for i in range(n)], [member(i) for i in range(n)]
    seeds = [51 + i for i in range(n)]
    g = LearnerGroup(members)
    g.step_n(buf.handle, 5, B, seeds)
    t.cuda.synchronize()
    sd = members[3].learner.state_dict()
    members[3].learner.load_state_dict(sd)           # (packs rebuilt from the masters, counters restored)
    g.step_n(buf.handle, 4, B, seeds)
    for a, s in zip(solo, seeds):
        assert a.learner.lib.oprl_learner_set_cluster(a.learner.handle, 1 if precision == "f32" else 4) == 0
        a.learner.step_n(buf.handle, 5, B, seed=s)
        a.learner.step_n(buf.handle, 4, B, seed=s)
    t.cuda.synchronize()
    for a, b in zip(members, solo):
        for m in ("actor", "critic", "actor_target", "critic_target"):
            assert t.equal(getattr(a, m)._oprl_arena, getattr(b, m)._oprl_arena), m
        a.learner.check()
    g.close()


@pytest.mark.parametrize("algo,precision,n_members", [("td3", "f32", 3), ("td3", "x2", 3), ("td3", "bf16", 3), ("sac", "f32", 3),
                                                      ("sac", "x2", 3), ("sac_tuned", "x2", 3), ("sac_tuned", "bf16", 3),
                                                      ("td3", "x2", 8), ("sac_tuned", "x2", 8)])
def test_packed_twin_critic_group_equals_solo_learners(algo, precision, n_members, monkeypatch):
    """N3 for the twin-critic algorithms (the reference's --seeds fan-out is algorithm-agnostic, runners/train.py:24-50):
    three TD3 / SAC learners stepped as a group end bit-identical to each of them alone on clusters of four with the
    twin critics back to back (the form group members run: no co-residency requirement between clusters) — TD3's
    delayed actor steps (phase 2 every other update for all members at once), SAC's learned temperature riding on the
    actor's dW launch, the device noise streams keyed per member."""
    from oprl_amd.group import LearnerGroup
    from oprl_amd.logging import NullLogger
    monkeypatch.setenv("OPRL_AMD_NO_SIDE_BY_SIDE", "1")
    B, K = 64, 7
    buf = _filled_buffer()

    def member(i):
        t.manual_seed(70 + i)
        kw = dict(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=B,
                  precision=precision, log_every=10 ** 9)
        if algo == "td3":
            from oprl_amd.algos.td3 import TD3
            return TD3(**kw).create()
        from oprl_amd.algos.sac import SAC
        return SAC(tune_alpha=algo == "sac_tuned", **kw).create()
    group_members, solo = [member(i) for i in range(n_members)], [member(i) for i in range(n_members)]
    seeds = [31 + i for i in range(n_members)]
    g = LearnerGroup(group_members)
    g.step_n(buf.handle, K, B, seeds)
    g.step_n(buf.handle, 3, B, seeds)             # a second call continues the streams (TD3: starting on a critic-only step)
    for a, s in zip(solo, seeds):
        assert a.learner.lib.oprl_learner_set_cluster(a.learner.handle, 4) == 0
        a.learner.step_n(buf.handle, K, B, seed=s)
        a.learner.step_n(buf.handle, 3, B, seed=s)
    t.cuda.synchronize()
    nets = ("actor", "critic", "actor_target", "critic_target") if algo == "td3" else ("actor", "critic", "critic_target")
    for a, b in zip(group_members, solo):
        for m in nets:
            assert t.equal(getattr(a, m)._oprl_arena, getattr(b, m)._oprl_arena), m
        assert t.equal(a.learner.critic_m, b.learner.critic_m) and t.equal(a.learner.actor_v, b.learner.actor_v)
        assert a.update_step == b.update_step == K + 3
        if algo == "sac_tuned":
            assert a.alpha == b.alpha and a.alpha != 0.2
        a.learner.check()
    # a member out of phase is refused, not silently stepped
    if algo == "td3":
        batch = buf.sample(B)
        group_members[1].update(*batch)
        with pytest.raises(RuntimeError, match="out of phase"):
            g.step_n(buf.handle, 2, B, seeds)
    g.close()


@pytest.mark.parametrize("precision,n_members", [("f32", 3), ("x2", 3), ("bf16", 3), ("f32", 8), ("x2", 8), ("f32", 16)])
def test_packed_learner_group_equals_solo_learners(precision, n_members):
    """N3: three / eight / sixteen independent DDPG learners stepped as a group (four launches per update for all of them) end
    with exactly the parameters, targets and Adam moments each of them reaches alone with the same launch form
    (exact fp32: cluster size 1; x2 / bf16: the un-merged lean launches on clusters of four, set_cluster(h, 4)) — and remain ordinary learners afterwards."""
    from oprl_amd.group import LearnerGroup
    B, K = 64, 7
    buf = _filled_buffer()

    def member(i):
        t.manual_seed(40 + i)
        from oprl_amd.algos.ddpg import DDPG
        from oprl_amd.logging import NullLogger
        return DDPG(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=B,
                    precision=precision).create()
    # (a multiple of eight members: member l's workgroups are dealt out to XCD l % 8, csrc/fused_ddpg.hip k_ddpg_phase1_group)
    group_members, solo = [member(i) for i in range(n_members)], [member(i) for i in range(n_members)]
    seeds = [11 + i for i in range(n_members)]
    g = LearnerGroup(group_members)
    g.step_n(buf.handle, K, B, seeds)
    g.step_n(buf.handle, 2, B, seeds)             # a second call continues the streams
    for a, s in zip(solo, seeds):
        assert a.learner.lib.oprl_learner_set_cluster(a.learner.handle, 1 if precision == "f32" else 4) == 0
        a.learner.step_n(buf.handle, K, B, seed=s)
        a.learner.step_n(buf.handle, 2, B, seed=s)
    t.cuda.synchronize()
    for a, b in zip(group_members, solo):
        for m in ("actor", "critic", "actor_target", "critic_target"):
            assert t.equal(getattr(a, m)._oprl_arena, getattr(b, m)._oprl_arena), m
        assert t.equal(a.learner.critic_m, b.learner.critic_m) and t.equal(a.learner.actor_v, b.learner.actor_v)
        assert a.update_step == b.update_step == K + 2
        a.learner.check()
    # members are still ordinary learners
    batch = buf.sample(B)
    group_members[0].update(*batch)
    solo[0].update(*batch)
    t.cuda.synchronize()
    assert t.equal(group_members[0].critic._oprl_arena, solo[0].critic._oprl_arena)
