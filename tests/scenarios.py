"""Scripted update scenarios shared by the oracle-vs-golden tests (CPU) and the
HIP-vs-oracle / HIP-vs-golden parity tests (GPU).  Each scenario mirrors a
``gen_*`` function of oracle/gen_golden.py but drives an *adapter* instead of
the reference, and returns the same keys the golden file holds."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch as t

from oracle import fixtures as fx
from oracle import oprl_oracle as orc

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_golden(name):
    return dict(np.load(GOLDEN / f"{name}.npz", allow_pickle=False))


def flatten(out):
    flat = {}
    for k, v in out.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat[f"{k}.{kk}"] = np.asarray(vv)
        elif isinstance(v, t.Tensor):
            flat[k] = v.detach().cpu().numpy()
        else:
            flat[k] = np.asarray(v)
    return flat


def rel_dev(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


PARAM_TOL = 1e-4   # north_star gate


def _is_param_key(k: str) -> bool:
    return any(f".{w}." in k for w in ("actor", "critic", "actor_target", "critic_target",
                                       "m_critic", "v_critic", "m_actor", "v_actor"))


def _is_moment_key(k: str) -> bool:
    return any(f".{w}." in k for w in ("m_critic", "v_critic", "m_actor", "v_actor"))


MOMENT_ELEM_TOL = 1e-4      # an Adam-moment element counts as "off" beyond this (relative to the key's largest magnitude) ...
MOMENT_OFF_FRAC = 0.15      # ... and at most this share of a moment key's elements may be off


def compare(got: dict, want: dict, tol: float, skip=(), param_tol: float | None = None, moment_tol: float | None = None,
            moment_off_frac: float | None = MOMENT_OFF_FRAC):
    """max-norm relative deviation per key; returns worst (key, dev).

    ``param_tol`` (GPU runs) applies to parameter / Adam-state digests: an
    element whose minibatch gradient cancels to ~1e-3 of its terms has its
    summation-order noise amplified by Adam's m/sqrt(v) (measured: one W3
    element of the SAC actor, step-1 gradient 2.7e-6 vs median 2.3e-3, moves by
    1.3% of one step = 6e-5 of max|W|).  Network outputs keep ``tol``.

    ``moment_tol`` (x2 runs) applies to the Adam-moment digests only: ONE ReLU unit whose pre-activation lies within
    rounding noise of zero for ONE minibatch row is masked on one side and not on the other (measured, update 10 of
    the DDPG scenario, tools/x2_probe.py: every moment agrees to 2e-7 through nine updates, then h1 unit 97 / h2 unit
    30 flip for one row and that unit's gradient differs by that row's term, 1.6e-3 of the largest moment; the
    parameters still agree to 1e-7).  Any two fp32 implementations with different summation orders do this to each
    other; the exact-fp32 mode happens not to on these seeds.

    Whatever the max-norm gate of a moment key is, the NUMBER of its elements further than ``MOMENT_ELEM_TOL`` from the
    expected value is bounded too (``MOMENT_OFF_FRAC`` of one optimizer state's sampled elements at one probe): a flipped ReLU unit of one row moves that unit's
    gradient row and what that row sends down the layers below — measured: 151 of the 1299 sampled elements of the x2
    DDPG learner's m_critic after update 10 (11.6 %; 84 of them in one layer's sample), ~ 6 % of the tuned-alpha SAC actor's
    (every mode, the oracle on another host included: tools/probe_moments.py), 0 everywhere else.  More flipped units — a wrong mask, a wrong seed for a group of rows — raise the count
    long before they raise the max-norm.  (``moment_off_frac=None``: a comparison across arithmetics — the bf16 learner
    against the fp32 vectors or against its emulation, where hidden activations sit on bf16 rounding boundaries — where
    elements are "off" by design.)"""
    worst = ("", 0.0)
    off_counts: dict = {}
    for k, w in want.items():
        if k == "meta" or any(k.startswith(s) for s in skip):
            continue
        assert k in got, f"missing key {k}"
        g = got[k]
        if w.dtype.kind in "US":
            assert list(g) == list(w), k
            continue
        assert g.shape == w.shape, (k, g.shape, w.shape)
        dev = rel_dev(g, w)
        if dev > worst[1]:
            worst = (k, dev)
        lim = param_tol if (param_tol is not None and _is_param_key(k)) else tol
        if moment_tol is not None and _is_moment_key(k):
            lim = moment_tol
        assert dev <= lim, f"{k}: rel dev {dev:.3e} > {lim:.1e}"
        if moment_off_frac is not None and _is_moment_key(k):
            # (counted over all the sampled layers of one optimizer state at one probe: "after10.m_critic")
            a, b = np.asarray(g, np.float64).ravel(), np.asarray(w, np.float64).ravel()
            grp = k.split(".")[0] + "." + k.split(".")[1]
            n_off, n_all = off_counts.get(grp, (0, 0))
            off_counts[grp] = (n_off + int((np.abs(a - b) > MOMENT_ELEM_TOL * max(np.abs(b).max(), 1e-30)).sum()), n_all + a.size)
    for grp, (n_off, n_all) in off_counts.items():
        assert n_all < 64 or n_off <= moment_off_frac * n_all, \
            f"{grp}: {n_off} of {n_all} sampled elements are more than {MOMENT_ELEM_TOL:.0e} off (limit {moment_off_frac:.0%})"
    return worst


# ------------------------------------------------------------------ adapters
class OracleDDPG:
    def __init__(self, S, A, actor, critic):
        self.o = orc.DDPGOracle(S, A, actor, critic)

    def update(self, s, a, r, d, s2):
        self.o.update(s, a, r, d, s2)

    def q(self, s, a): return orc.q_forward(self.o.critic, s, a)[-1]
    def q_target_pi(self, s2):
        return orc.q_forward(self.o.critic_target, s2, orc.det_policy_forward(self.o.actor_target, s2)[0])[-1]
    def pi(self, s): return orc.det_policy_forward(self.o.actor, s)[0]
    def params(self, which): return getattr(self.o, which)
    def adam(self, which):
        opt = self.o.opt_critic if which == "critic" else self.o.opt_actor
        return opt.m, opt.v
    def step1_grads(self): return self._g

    def hook_step1(self):
        self._g = (self.o.last["g_critic"], self.o.last["g_actor"])


def _moments(out, tag, algo, n=256):
    """m / v digests of both optimizers, as oracle/gen_golden.py moments() wrote them (an optimizer that has not
    stepped yet — none in these scripts — has no state and no keys)."""
    for w in ("critic", "actor"):
        m, v = algo.adam(w)
        if not m:
            continue
        out[f"{tag}.m_{w}"] = fx.digest_list(m, n=n)
        out[f"{tag}.v_{w}"] = fx.digest_list(v, n=n)


def ddpg_scenario(make, B=256):
    S, A = fx.ENVS["walker"]
    seed = 100
    actor = fx.make_net(seed + 1, fx.actor_dims(S, A))
    critic = fx.make_net(seed + 2, fx.critic_dims(S, A))
    algo = make(S, A, actor, critic)
    out = {}
    for step in range(10):
        s, a, r, d, s2 = fx.make_batch(seed + 10 + step, B, S, A)
        if step == 0:
            out["y0"] = r + (1.0 - d) * 0.99 * algo.q_target_pi(s2).cpu()
            out["q0"] = algo.q(s, a)
        algo.update(s, a, r, d, s2)
        if step == 0:
            algo.hook_step1()
            gc, ga = algo.step1_grads()
            out["g_critic_1"] = fx.digest_list(gc)
            out["g_actor_1"] = fx.digest_list(ga)
        if step in (0, 9):
            tag = f"after{step + 1}"
            s, a, r, d, s2 = fx.make_batch(seed + 99, B, S, A)
            out[f"{tag}.q"] = algo.q(s, a)
            out[f"{tag}.q_target"] = algo.q_target_pi(s2)
            out[f"{tag}.pi"] = algo.pi(s)
            for w in ("actor", "critic", "actor_target", "critic_target"):
                out[f"{tag}.{w}"] = fx.digest_list(algo.params(w))
            m, v = algo.adam("critic")
            out[f"{tag}.m_critic"] = fx.digest_list(m)
            out[f"{tag}.v_critic"] = fx.digest_list(v)
            m, v = algo.adam("actor")
            out[f"{tag}.m_actor"] = fx.digest_list(m)
            out[f"{tag}.v_actor"] = fx.digest_list(v)
    return flatten(out)


class OracleTD3:
    def __init__(self, S, A, actor, c1, c2):
        self.o = orc.TD3Oracle(S, A, actor, c1, c2)

    def update(self, s, a, r, d, s2, noise): self.o.update(s, a, r, d, s2, noise)
    def q(self, s, a, j, target=False):
        p = self.o.critic_target if target else self.o.critic
        return orc.q_forward(self.o._q(p, j), s, a)[-1]
    def pi(self, s, target=False):
        return orc.det_policy_forward(self.o.actor_target if target else self.o.actor, s)[0]
    def params(self, which): return getattr(self.o, which)
    def adam(self, which):
        opt = self.o.opt_critic if which == "critic" else self.o.opt_actor
        return opt.m, opt.v


def td3_scenario(make, B=256):
    S, A = fx.ENVS["cheetah"]
    seed = 200
    actor = fx.make_net(seed + 1, fx.actor_dims(S, A))
    c1 = fx.make_net(seed + 2, fx.critic_dims(S, A))
    c2 = fx.make_net(seed + 3, fx.critic_dims(S, A))
    algo = make(S, A, actor, c1, c2)
    out = {}
    for step in range(3):
        s, a, r, d, s2 = fx.make_batch(seed + 10 + step, B, S, A)
        algo.update(s, a, r, d, s2, fx.make_noise(seed + 50 + step, (B, A)))
        tag = f"after{step + 1}"
        s, a, r, d, s2 = fx.make_batch(seed + 99, B, S, A)
        out[f"{tag}.q1"], out[f"{tag}.q2"] = algo.q(s, a, 0), algo.q(s, a, 1)
        out[f"{tag}.pi"] = algo.pi(s)
        out[f"{tag}.pi_target"] = algo.pi(s, target=True)
        out[f"{tag}.tq1"], out[f"{tag}.tq2"] = algo.q(s, a, 0, True), algo.q(s, a, 1, True)
        out[f"{tag}.actor"] = fx.digest_list(algo.params("actor"))
        out[f"{tag}.critic"] = fx.digest_list(algo.params("critic"))
        _moments(out, tag, algo)
    return flatten(out)


class OracleSAC:
    def __init__(self, S, A, actor, c1, c2, tune_alpha):
        self.o = orc.SACOracle(S, A, actor, c1, c2, tune_alpha=tune_alpha)
        self.A = A

    def update(self, s, a, r, d, s2, e1, e2): self.o.update(s, a, r, d, s2, e1, e2)
    def q(self, s, a, j, target=False):
        p = self.o.critic_target if target else self.o.critic
        return orc.q_forward(self.o._q(p, j), s, a)[-1]
    def pi_logp(self, s, eps):
        a, lp, _ = orc.gaussian_forward(self.o.actor, s, eps, self.A)
        return a, lp
    def params(self, which): return getattr(self.o, which)
    def adam(self, which):
        opt = self.o.opt_critic if which == "critic" else self.o.opt_actor
        return opt.m, opt.v
    @property
    def alpha(self): return self.o.alpha


def sac_scenario(make, env, B, seed, tune_alpha, n_steps):
    S, A = fx.ENVS[env]
    actor = fx.make_net(seed + 1, fx.actor_dims(S, A, gaussian=True))
    c1 = fx.make_net(seed + 2, fx.critic_dims(S, A))
    c2 = fx.make_net(seed + 3, fx.critic_dims(S, A))
    algo = make(S, A, actor, c1, c2, tune_alpha)
    out, alphas = {}, []
    for step in range(n_steps):
        s, a, r, d, s2 = fx.make_batch(seed + 10 + step, B, S, A)
        algo.update(s, a, r, d, s2, fx.make_noise(seed + 50 + step, (B, A)),
                    fx.make_noise(seed + 70 + step, (B, A)))
        alphas.append(algo.alpha)
        tag = f"after{step + 1}"
        s, a, r, d, s2 = fx.make_batch(seed + 99, B, S, A)
        out[f"{tag}.q1"], out[f"{tag}.q2"] = algo.q(s, a, 0), algo.q(s, a, 1)
        out[f"{tag}.tq1"], out[f"{tag}.tq2"] = algo.q(s, a, 0, True), algo.q(s, a, 1, True)
        out[f"{tag}.pi"], out[f"{tag}.logp"] = algo.pi_logp(s, fx.make_noise(seed + 98, (B, A)))
        out[f"{tag}.actor"] = fx.digest_list(algo.params("actor"))
        out[f"{tag}.critic"] = fx.digest_list(algo.params("critic"))
        _moments(out, tag, algo)
    out["alphas"] = np.array(alphas, np.float64)
    return flatten(out)


class OracleTQC:
    def __init__(self, S, A, actor, critics):
        self.o = orc.TQCOracle(S, A, actor, critics)
        self.A = A

    def update(self, s, a, r, d, s2, e1, e2): self.o.update(s, a, r, d, s2, e1, e2)
    def z(self, s, a, target=False):
        nets = self.o.critics_target if target else self.o.critics
        return t.stack([orc.q_forward(c, s, a)[-1] for c in nets], dim=1)
    def pi_logp(self, s, eps):
        a, lp, _ = orc.gaussian_forward(self.o.actor, s, eps, self.A)
        return a, lp
    def params(self, which):
        if which == "critic":
            return [x for c in self.o.critics for x in c]
        return self.o.actor
    def adam(self, which):
        opt = self.o.opt_critic if which == "critic" else self.o.opt_actor
        return opt.m, opt.v
    @property
    def log_alpha(self): return float(self.o.log_alpha)


def tqc_scenario(make, n_steps=2, B=256):
    S, A = fx.ENVS["walker"]
    seed = 400
    actor = fx.make_net(seed + 1, fx.actor_dims(S, A, gaussian=True))
    critics = [fx.make_net(seed + 2 + n, fx.critic_dims(S, A, out=25, hidden=(512, 512, 512)))
               for n in range(5)]
    algo = make(S, A, actor, critics)
    out, las = {}, []
    for step in range(n_steps):
        s, a, r, d, s2 = fx.make_batch(seed + 10 + step, B, S, A)
        algo.update(s, a, r, d, s2, fx.make_noise(seed + 50 + step, (B, A)),
                    fx.make_noise(seed + 70 + step, (B, A)))
        las.append(algo.log_alpha)
        tag = f"after{step + 1}"
        s, a, r, d, s2 = fx.make_batch(seed + 99, B, S, A)
        out[f"{tag}.z"] = algo.z(s, a)
        out[f"{tag}.tz"] = algo.z(s, a, True)
        out[f"{tag}.pi"], out[f"{tag}.logp"] = algo.pi_logp(s, fx.make_noise(seed + 98, (B, A)))
        out[f"{tag}.actor"] = fx.digest_list(algo.params("actor"))
        out[f"{tag}.critic"] = fx.digest_list(algo.params("critic"), n=64)
        _moments(out, tag, algo, n=64)
    out["log_alphas"] = np.array(las, np.float64)
    return flatten(out)


def replay_scenario(buf, S, A, snapshot):
    """Runs the scripted add sequence (same script the golden generator ran on
    the reference).  ``snapshot(buf)`` -> (trace_row, dict of gathered arrays
    for flat indices 0..len-1) is taken after every scripted stage."""
    from oracle.gen_golden import replay_script  # pure python; does not import the reference
    trace, gathers = [], {}

    def record():
        i = len(trace)
        row, g = snapshot(buf)
        trace.append(row)
        for k, v in g.items():
            gathers[f"g{i}.{k}"] = v

    replay_script(buf, S, A, record)
    out = dict(trace=np.array(trace, np.int64))
    out.update(gathers)
    return out
