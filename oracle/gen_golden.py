"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (python, torch-CPU).

Run only in the build container (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

The reference's python never travels to the GPU box; only the vectors written
here do.  Inputs are regenerated from seeds by oracle/fixtures.py, so each
fixture stores seeds/dims + expected outputs (digests of the big tensors).

Stubs: the reference's logging module imports tensorboard (absent here); a
no-op SummaryWriter is injected.  All Gaussian noise drawn inside update() is
replaced by pre-generated RandomState draws (patched torch.randn_like /
Normal.sample), recorded by seed.
"""
from __future__ import annotations

import os
import sys
import types
from pathlib import Path

import numpy as np
import torch as t

sys.dont_write_bytecode = True
REF_SRC = "/root/reference/src"
OUT = Path(__file__).resolve().parents[1] / "tests" / "golden"

from . import fixtures as fx  # noqa: E402


def _install_stubs():
    tb = types.ModuleType("torch.utils.tensorboard")
    wr = types.ModuleType("torch.utils.tensorboard.writer")

    class SummaryWriter:  # noqa: D401
        def __init__(self, *a, **k): ...
        def add_scalar(self, *a, **k): ...

    wr.SummaryWriter = SummaryWriter
    tb.writer = wr
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    sys.modules["torch.utils.tensorboard.writer"] = wr
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    # ``oprl`` must be the REFERENCE here: it has no __init__.py (a namespace package), so this repo's
    # ``oprl/`` alias package (a regular package on sys.path[0] when run from the repo root) would win the
    # import — pin the name to the reference's directory explicitly
    for name in [m for m in sys.modules if m == "oprl" or m.startswith("oprl.")]:
        del sys.modules[name]
    pkg = types.ModuleType("oprl")
    pkg.__path__ = [os.path.join(REF_SRC, "oprl")]
    sys.modules["oprl"] = pkg


class NullLogger:
    log_dir = Path("/tmp/oprl_golden_logs")

    def log_scalar(self, *a, **k): ...
    def log_scalars(self, *a, **k): ...


class NoiseFeed:
    """Replaces torch.randn_like and Normal.sample with queued draws."""

    def __init__(self):
        self.queue: list[t.Tensor] = []
        self._orig_randn_like = t.randn_like
        self._orig_sample = t.distributions.Normal.sample

    def __enter__(self):
        feed = self

        def randn_like(x, *a, **k):
            z = feed.queue.pop(0)
            assert z.shape == x.shape, (z.shape, x.shape)
            return z

        def sample(self_, sample_shape=t.Size()):
            z = feed.queue.pop(0)
            assert z.shape == self_.loc.shape, (z.shape, self_.loc.shape)
            return z

        t.randn_like = randn_like
        t.distributions.Normal.sample = sample
        return self

    def __exit__(self, *exc):
        t.randn_like = self._orig_randn_like
        t.distributions.Normal.sample = self._orig_sample


def load_params(module, params):
    ps = list(module.parameters())
    assert len(ps) == len(params), (len(ps), len(params))
    with t.no_grad():
        for dst, src in zip(ps, params):
            assert dst.shape == src.shape, (dst.shape, src.shape)
            dst.copy_(src)


def plist(module):
    return [p.detach().clone() for p in module.parameters()]


def grads(module):
    return [p.grad.detach().clone() for p in module.parameters()]


def moments(out, tag, algo, n=256):
    """Adam moments of both optimizers (torch.optim.Adam state: exp_avg / exp_avg_sq) in parameters() order — what a
    parameter digest cannot show: Adam's step is scale-invariant in the gradient, the moments are not."""
    oc = algo.optim_critic if hasattr(algo, "optim_critic") else algo.critic_optimizer      # (tqc.py:109-110 names them the other way round)
    oa = algo.optim_actor if hasattr(algo, "optim_actor") else algo.actor_optimizer
    for w, opt, mod in (("critic", oc, algo.critic), ("actor", oa, algo.actor)):
        ps = list(mod.parameters())
        if not all(p in opt.state and "exp_avg" in opt.state[p] for p in ps):
            continue
        out[f"{tag}.m_{w}"] = fx.digest_list([opt.state[p]["exp_avg"] for p in ps], n=n)
        out[f"{tag}.v_{w}"] = fx.digest_list([opt.state[p]["exp_avg_sq"] for p in ps], n=n)


def save(name, **arrays):
    OUT.mkdir(parents=True, exist_ok=True)
    flat = {}
    for k, v in arrays.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}.{kk}"] = np.asarray(vv)
        elif isinstance(v, t.Tensor):
            flat[k] = v.detach().numpy()
        else:
            flat[k] = np.asarray(v)
    np.savez_compressed(OUT / f"{name}.npz", **flat)
    print(f"wrote {name}.npz  ({(OUT / (name + '.npz')).stat().st_size / 1024:.0f} KB)")


# ------------------------------------------------------------------ DDPG (G1)
def gen_ddpg():
    from oprl.algos.ddpg import DDPG
    S, A = fx.ENVS["walker"]
    B, seed = 256, 100
    algo = DDPG(logger=NullLogger(), state_dim=S, action_dim=A).create()
    actor = fx.make_net(seed + 1, fx.actor_dims(S, A))
    critic = fx.make_net(seed + 2, fx.critic_dims(S, A))
    for m, p in ((algo.actor, actor), (algo.actor_target, actor),
                 (algo.critic, critic), (algo.critic_target, critic)):
        load_params(m, p)
    out = dict(meta=np.array([S, A, B, seed, 10], np.int64))
    for step in range(10):
        s, a, r, d, s2 = fx.make_batch(seed + 10 + step, B, S, A)
        if step == 0:
            with t.no_grad():
                tq = algo.critic_target(s2, algo.actor_target(s2))
                out["y0"] = r + (1.0 - d) * algo.gamma * tq
                out["q0"] = algo.critic(s, a)
            # grads of step 1, captured via hooks on optimiser steps
            cg, ag = {}, {}
            def _hc(*_):
                cg.setdefault("g", grads(algo.critic))

            def _ha(*_):
                ag.setdefault("g", grads(algo.actor))

            h1 = algo.optim_critic.register_step_pre_hook(_hc)
            h2 = algo.optim_actor.register_step_pre_hook(_ha)
        algo.update(s, a, r, d, s2)
        if step == 0:
            h1.remove(); h2.remove()
            out["g_critic_1"] = fx.digest_list(cg["g"])
            out["g_actor_1"] = fx.digest_list(ag["g"])
        if step in (0, 9):
            tag = f"after{step + 1}"
            s, a, r, d, s2 = fx.make_batch(seed + 99, B, S, A)  # probe batch
            with t.no_grad():
                out[f"{tag}.q"] = algo.critic(s, a)
                out[f"{tag}.q_target"] = algo.critic_target(s2, algo.actor_target(s2))
                out[f"{tag}.pi"] = algo.actor(s)
            out[f"{tag}.actor"] = fx.digest_list(plist(algo.actor))
            out[f"{tag}.critic"] = fx.digest_list(plist(algo.critic))
            out[f"{tag}.actor_target"] = fx.digest_list(plist(algo.actor_target))
            out[f"{tag}.critic_target"] = fx.digest_list(plist(algo.critic_target))
            out[f"{tag}.m_critic"] = fx.digest_list(
                [algo.optim_critic.state[p]["exp_avg"] for p in algo.critic.parameters()])
            out[f"{tag}.v_critic"] = fx.digest_list(
                [algo.optim_critic.state[p]["exp_avg_sq"] for p in algo.critic.parameters()])
            # (the actor's moments too: Adam's step is scale-invariant in the gradient, so the actor's PARAMETERS alone
            # would not show an actor-gradient scale error — the moments do)
            out[f"{tag}.m_actor"] = fx.digest_list(
                [algo.optim_actor.state[p]["exp_avg"] for p in algo.actor.parameters()])
            out[f"{tag}.v_actor"] = fx.digest_list(
                [algo.optim_actor.state[p]["exp_avg_sq"] for p in algo.actor.parameters()])
    save("ddpg_walker_b256", **out)


# ------------------------------------------------------------------- TD3 (G2)
def gen_td3():
    from oprl.algos.td3 import TD3
    S, A = fx.ENVS["cheetah"]
    B, seed = 256, 200
    algo = TD3(logger=NullLogger(), state_dim=S, action_dim=A, log_every=10 ** 9).create()
    actor = fx.make_net(seed + 1, fx.actor_dims(S, A))
    c1 = fx.make_net(seed + 2, fx.critic_dims(S, A))
    c2 = fx.make_net(seed + 3, fx.critic_dims(S, A))
    load_params(algo.actor, actor); load_params(algo.actor_target, actor)
    load_params(algo.critic, c1 + c2); load_params(algo.critic_target, c1 + c2)
    out = dict(meta=np.array([S, A, B, seed, 3], np.int64))
    with NoiseFeed() as feed:
        for step in range(3):
            s, a, r, d, s2 = fx.make_batch(seed + 10 + step, B, S, A)
            feed.queue.append(fx.make_noise(seed + 50 + step, (B, A)))
            algo.update(s, a, r, d, s2)
            tag = f"after{step + 1}"
            s, a, r, d, s2 = fx.make_batch(seed + 99, B, S, A)
            with t.no_grad():
                q1, q2 = algo.critic(s, a)
                out[f"{tag}.q1"], out[f"{tag}.q2"] = q1, q2
                out[f"{tag}.pi"] = algo.actor(s)
                out[f"{tag}.pi_target"] = algo.actor_target(s)
                tq1, tq2 = algo.critic_target(s, a)
                out[f"{tag}.tq1"], out[f"{tag}.tq2"] = tq1, tq2
            out[f"{tag}.actor"] = fx.digest_list(plist(algo.actor))
            out[f"{tag}.critic"] = fx.digest_list(plist(algo.critic))
            moments(out, tag, algo)
    save("td3_cheetah_b256", **out)


# ------------------------------------------------------------------- SAC (G3)
def gen_sac(env, B, seed, tune_alpha, n_steps, name):
    from oprl.algos.sac import SAC
    S, A = fx.ENVS[env]
    algo = SAC(logger=NullLogger(), state_dim=S, action_dim=A, tune_alpha=tune_alpha,
               log_every=10 ** 9).create()
    actor = fx.make_net(seed + 1, fx.actor_dims(S, A, gaussian=True))
    c1 = fx.make_net(seed + 2, fx.critic_dims(S, A))
    c2 = fx.make_net(seed + 3, fx.critic_dims(S, A))
    load_params(algo.actor, actor)
    load_params(algo.critic, c1 + c2); load_params(algo.critic_target, c1 + c2)
    out = dict(meta=np.array([S, A, B, seed, n_steps, int(tune_alpha)], np.int64))
    alphas = []
    with NoiseFeed() as feed:
        for step in range(n_steps):
            s, a, r, d, s2 = fx.make_batch(seed + 10 + step, B, S, A)
            feed.queue.append(fx.make_noise(seed + 50 + step, (B, A)))   # next-state draw
            feed.queue.append(fx.make_noise(seed + 70 + step, (B, A)))   # actor-step draw
            algo.update(s, a, r, d, s2)
            alphas.append(algo.alpha)
            tag = f"after{step + 1}"
            s, a, r, d, s2 = fx.make_batch(seed + 99, B, S, A)
            with t.no_grad():
                q1, q2 = algo.critic(s, a)
                out[f"{tag}.q1"], out[f"{tag}.q2"] = q1, q2
                tq1, tq2 = algo.critic_target(s, a)
                out[f"{tag}.tq1"], out[f"{tag}.tq2"] = tq1, tq2
                feed.queue.append(fx.make_noise(seed + 98, (B, A)))
                pi, lp = algo.actor(s)
                out[f"{tag}.pi"], out[f"{tag}.logp"] = pi, lp
            out[f"{tag}.actor"] = fx.digest_list(plist(algo.actor))
            out[f"{tag}.critic"] = fx.digest_list(plist(algo.critic))
            moments(out, tag, algo)
    out["alphas"] = np.array(alphas, np.float64)
    save(name, **out)


# ------------------------------------------------------------------- TQC (G4)
def gen_tqc():
    from oprl.algos.tqc import TQC, quantile_huber_loss_f
    S, A = fx.ENVS["walker"]
    B, seed, n_steps = 256, 400, 2
    algo = TQC(logger=NullLogger(), state_dim=S, action_dim=A, log_every=10 ** 9).create()
    actor = fx.make_net(seed + 1, fx.actor_dims(S, A, gaussian=True))
    crit = []
    for n in range(5):
        crit += fx.make_net(seed + 2 + n, fx.critic_dims(S, A, out=25, hidden=(512, 512, 512)))
    load_params(algo.actor, actor)
    load_params(algo.critic, crit); load_params(algo.critic_target, crit)
    out = dict(meta=np.array([S, A, B, seed, n_steps], np.int64))
    log_alphas = []
    with NoiseFeed() as feed:
        for step in range(n_steps):
            s, a, r, d, s2 = fx.make_batch(seed + 10 + step, B, S, A)
            feed.queue.append(fx.make_noise(seed + 50 + step, (B, A)))
            feed.queue.append(fx.make_noise(seed + 70 + step, (B, A)))
            algo.update(s, a, r, d, s2)
            log_alphas.append(float(algo.log_alpha.item()))
            tag = f"after{step + 1}"
            s, a, r, d, s2 = fx.make_batch(seed + 99, B, S, A)
            with t.no_grad():
                out[f"{tag}.z"] = algo.critic(s, a)
                out[f"{tag}.tz"] = algo.critic_target(s, a)
                feed.queue.append(fx.make_noise(seed + 98, (B, A)))
                pi, lp = algo.actor(s)
                out[f"{tag}.pi"], out[f"{tag}.logp"] = pi, lp
            out[f"{tag}.actor"] = fx.digest_list(plist(algo.actor))
            out[f"{tag}.critic"] = fx.digest_list(plist(algo.critic), n=64)
            moments(out, tag, algo, n=64)
    out["log_alphas"] = np.array(log_alphas, np.float64)

    # standalone quantile-Huber known-answer case (value + gradient)
    rs = np.random.RandomState(seed + 500)
    z = t.from_numpy((rs.standard_normal((8, 5, 25)) * 1.5).astype(np.float32)).requires_grad_(True)
    y = t.from_numpy((rs.standard_normal((8, 123)) * 1.5).astype(np.float32))
    loss = quantile_huber_loss_f(z, y, "cpu")
    loss.backward()
    out["qh.loss"] = loss.detach()
    out["qh.dz"] = z.grad
    save("tqc_walker_b256", **out)


# ---------------------------------------------------------------- replay (G5)
def replay_script(buf, S, A, record):
    """Scripted add sequence: ring wrap, eviction, in-progress tail, add_episode
    with and without a terminal last row."""
    k = [0]

    def tr(done=False):
        k[0] += 1
        s = np.full((S,), k[0], np.float32)
        a = np.full((A,), -k[0], np.float64)       # float64 actions are accepted (cast)
        return s, a, float(k[0]), done

    def add_ep(n):
        for i in range(n):
            s, a, r, d = tr()
            buf.add_transition(s, a, r, d, episode_done=(i == n - 1))
        record()

    for n in (4, 3, 4, 2):
        add_ep(n)
    s, a, r, d = tr()
    buf.add_transition(s, a, r, d, episode_done=False)
    record()
    ep = []
    for i in range(3):
        s, a, r, d = tr()
        ep.append([s, a, r, False, s + 1])
    buf.add_episode(ep)
    record()
    ep = []
    for i in range(2):
        s, a, r, d = tr()
        ep.append([s, a, r, i == 1, s + 1])
    buf.add_episode(ep)
    record()
    add_ep(4)
    for _ in range(2):
        s, a, r, d = tr()
        buf.add_transition(s, a, r, d, episode_done=False)
    record()


def gen_replay():
    from oprl.buffers.episodic_buffer import EpisodicReplayBuffer
    S, A = 2, 3
    buf = EpisodicReplayBuffer(buffer_size_transitions=16, state_dim=S, action_dim=A,
                               max_episode_lenth=4).create()
    for k in ("states", "actions", "rewards", "dones"):
        buf._tensors[k].fill_(-99.0)
    trace = []
    gathers = {}

    def record():
        i = len(trace)
        trace.append([len(buf), buf.episodes_counter, buf._ep_pointer, buf.last_episode_length,
                      *buf.ep_lens])
        n = len(buf)
        if n > 0:
            inds = np.arange(n)
            e, st = buf._inds_to_episodic(inds)
            gathers[f"g{i}.ep"] = e
            gathers[f"g{i}.step"] = st
            gathers[f"g{i}.s"] = buf.states[e, st].numpy().copy()
            gathers[f"g{i}.a"] = buf.actions[e, st].numpy().copy()
            gathers[f"g{i}.r"] = buf.rewards[e, st].numpy().copy()
            gathers[f"g{i}.d"] = buf.dones[e, st].numpy().copy()
            gathers[f"g{i}.s2"] = buf.states[e, st + 1].numpy().copy()

    replay_script(buf, S, A, record)
    save("replay_script", trace=np.array(trace, np.int64), meta=np.array([16, S, A, 4], np.int64), **gathers)


# ------------------------------------------------------------ policy I/O (G6)
def gen_policy_io():
    from oprl.algos.nn_models import DeterministicPolicy, GaussianActor
    import torch.nn as nn
    S, A = fx.ENVS["walker"]
    seed = 600
    det = DeterministicPolicy(S, A)
    load_params(det, fx.make_net(seed + 1, fx.actor_dims(S, A)))
    ga = GaussianActor(S, A, (256, 256), nn.ReLU(), "cpu")
    load_params(ga, fx.make_net(seed + 2, fx.actor_dims(S, A, gaussian=True)))
    obs = np.random.RandomState(seed + 3).standard_normal(S).astype(np.float32)
    out = dict(meta=np.array([S, A, seed], np.int64))
    out["det.exploit"] = det.exploit(obs)
    orig = t.randn
    nz = fx.make_noise(seed + 4, (A,))
    t.randn = lambda *a, **k: nz
    try:
        out["det.explore"] = det.explore(obs)
    finally:
        t.randn = orig
    out["ga.exploit"] = ga.exploit(obs)
    with NoiseFeed() as feed:
        feed.queue.append(fx.make_noise(seed + 5, (1, A)))
        out["ga.explore"] = ga.explore(obs)
    out["det.keys"] = np.array(list(det.state_dict().keys()))
    out["ga.keys"] = np.array(list(ga.state_dict().keys()))
    save("policy_io", **out)


# --------------------------------------------------------------------------------------------------
# G7: the CALLERS of the hot path — the reference's BaseTrainer.train,
# run_policy_update_worker and run_env_worker driven with recording fakes (tests/caller_fakes.py); the
# stored traces are what this repo's loop counterparts must reproduce (tests/test_callers_golden.py).
# --------------------------------------------------------------------------------------------------
TRAINER_KW = dict(num_steps=60, start_steps=12, batch_size=8, eval_interval=20, num_eval_episodes=2,
                  save_policy_every=30, stdout_log_every=25, seed=3)
WORKER_CFG = dict(batch_size=4, num_env_workers=3, episodes_per_worker=5, warmup_epochs=1, episode_length=6,
                  learner_num_waits=2, warmup_env_steps=8)


LEARNER_EPOCHS_FED = 12      # epochs 0..11: training from epoch 2 on, evaluation + policy save at epoch 10


def _install_caller_stubs():
    """Import-time dependencies of the reference's trainer / workers that are absent here (simulators, the
    broker client, pydantic-settings): empty modules — the fakes replace everything they would do."""
    import pydantic
    for name, attrs in (("dm_control", {"suite": types.SimpleNamespace()}), ("gymnasium", {}), ("pika", {}),
                        ("pydantic_settings", {"BaseSettings": pydantic.BaseModel})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m


def gen_callers():
    import tempfile
    import time
    from tests import caller_fakes as cf
    _install_caller_stubs()
    from oprl.trainers.base_trainer import BaseTrainer
    import oprl.distrib.policy_update_worker as puw
    import oprl.distrib.env_worker as ew
    out = {}
    # ---- BaseTrainer.train (base_trainer.py:38-74)
    with tempfile.TemporaryDirectory() as td:
        tr = cf.Trace()
        env = cf.FakeEnv(tr, "env", length=9, terminate_at=31)
        BaseTrainer(logger=cf.FakeLogger(tr, Path(td)), env=env,
                    make_env_test=lambda seed: (tr(f"make_env_test {seed}"), cf.FakeEnv(tr, "test_env", length=4))[1],
                    replay_buffer=cf.FakeBuffer(tr), algo=cf.FakeAlgo(tr), **TRAINER_KW).train()
        out["trainer"] = np.array(cf.compress(tr.events))
        out["trainer_files"] = np.array(sorted(str(f.relative_to(td)) for f in Path(td).rglob("*") if f.is_file()))
    # ---- run_policy_update_worker (policy_update_worker.py:22-92): three actors' queues pre-filled with twelve
    # epochs of episodes; the learner then finds them empty and gives up after learner_num_waits polls
    cfg = types.SimpleNamespace(**WORKER_CFG)
    names = [f"{k}_{i}" for i in range(cfg.num_env_workers) for k in ("env", "policy")]
    with tempfile.TemporaryDirectory() as td:
        tr = cf.Trace()
        reg = cf.Registry(tr, names)
        import pickle
        for epoch in range(LEARNER_EPOCHS_FED):
            for i in range(cfg.num_env_workers):
                ep = [[np.zeros(cf.S, np.float32), np.zeros(cf.A, np.float32), 0.0, False, np.zeros(cf.S, np.float32)]
                      for _ in range(cfg.episode_length - (i == 1))]
                reg.fifo[f"env_{i}"].append(pickle.dumps(ep))
        saved_q, saved_sleep = puw.Queue, time.sleep
        puw.Queue = reg.reference_queue_class()
        time.sleep = lambda s: tr("sleep")
        try:
            puw.run_policy_update_worker(
                make_algo=lambda lg: cf.FakeAlgo(tr, lg),
                make_env_test=lambda seed: (tr(f"make_env_test {seed}"), cf.FakeEnv(tr, "test_env", length=4))[1],
                make_buffer=lambda: cf.FakeBuffer(tr), make_logger=lambda: cf.FakeLogger(tr, Path(td)), config=cfg)
        finally:
            puw.Queue, time.sleep = saved_q, saved_sleep
        out["learner"] = np.array(cf.compress([e for e in tr.events if e != "sleep"]))
        out["learner_files"] = np.array(sorted(str(f.relative_to(td)) for f in Path(td).rglob("*") if f.is_file()))
    # ---- run_env_worker (env_worker.py:15-64): the policy queue answers every episode at once
    tr = cf.Trace()
    reg = cf.Registry(tr, names)
    for _ in range(cfg.episodes_per_worker):
        reg.fifo["policy_1"].append(pickle.dumps({"w": t.zeros(1)}))
    saved_q, saved_sleep = ew.Queue, time.sleep
    ew.Queue = reg.reference_queue_class()
    time.sleep = lambda s: tr("sleep")
    import builtins
    saved_print = builtins.print
    builtins.print = lambda *a, **k: None
    try:
        ew.run_env_worker(make_env=lambda seed: cf.FakeEnv(tr, "env", length=cfg.episode_length, terminate_at=20),
                          make_policy=lambda: cf.FakeActor(tr), config=cfg, id_worker=1)
    finally:
        ew.Queue, time.sleep, builtins.print = saved_q, saved_sleep, saved_print
    out["actor"] = np.array(cf.compress([e for e in tr.events if e != "sleep"]))
    save("callers", **out)


def main():
    assert os.path.isdir(REF_SRC), "run in the build container (needs /root/reference)"
    t.set_num_threads(1)
    _install_stubs()
    gen_ddpg()
    gen_td3()
    gen_sac("humanoid", 1024, 300, False, 2, "sac_humanoid_b1024")
    gen_sac("walker", 256, 350, True, 3, "sac_walker_tune_b256")
    gen_tqc()
    gen_replay()
    gen_policy_io()
    gen_callers()


if __name__ == "__main__":
    main()
