"""Adapters that drive the HIP algorithms (through their public Python API, i.e.
through the C-ABI) with the interface tests/scenarios.py expects."""
from __future__ import annotations

import torch as t

from oprl_amd.algos.ddpg import DDPG
from oprl_amd.algos.sac import SAC
from oprl_amd.algos.td3 import TD3
from oprl_amd.algos.tqc import TQC
from oprl_amd.logging import NullLogger

DEV = "cuda"


def load_params(module, params):
    ps = list(module.parameters())
    assert len(ps) == len(params), (len(ps), len(params))
    with t.no_grad():
        for dst, src in zip(ps, params):
            assert dst.shape == src.shape, (dst.shape, src.shape)
            dst.copy_(src)               # in place (bumps _version): parameters stay arena views


def cpu_params(module):
    return [p.detach().cpu().clone() for p in module.parameters()]


def split_like(flat, module):
    out, off = [], 0
    for p in module.parameters():
        out.append(flat[off:off + p.numel()].view(p.shape).detach().cpu().clone())
        off += p.numel()
    return out


def g(x):
    return x.to(DEV)


def hip_adam(algo, which):
    """(m, v) of one optimizer as per-parameter CPU tensors, in the module's parameters() order (the learner keeps each
    as one flat arena view)."""
    L = algo.learner
    mod = getattr(algo, which)
    m, v = (L.critic_m, L.critic_v) if which == "critic" else (L.actor_m, L.actor_v)
    return split_like(m, mod), split_like(v, mod)


class HipDDPG:
    def __init__(self, S, A, actor, critic, **kw):
        self.args = (S, A, actor, critic)
        self.kw = kw
        self.algo = DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device=DEV, **kw).create()
        for m, p in ((self.algo.actor, actor), (self.algo.actor_target, actor),
                     (self.algo.critic, critic), (self.algo.critic_target, critic)):
            load_params(m, p)
        self._first = None

    def update(self, s, a, r, d, s2):
        if self._first is None:
            self._first = (s, a, r, d, s2)
        self.algo.update(g(s), g(a), g(r), g(d), g(s2))

    def q(self, s, a): return self.algo.critic(g(s), g(a)).cpu()
    def q_target_pi(self, s2): return self.algo.critic_target(g(s2), self.algo.actor_target(g(s2))).cpu()
    def pi(self, s): return self.algo.actor(g(s)).cpu()
    def params(self, which): return cpu_params(getattr(self.algo, which))
    def adam(self, which):
        L = self.algo.learner
        mod = getattr(self.algo, which)
        m, v = (L.critic_m, L.critic_v) if which == "critic" else (L.actor_m, L.actor_v)
        return split_like(m, mod), split_like(v, mod)

    def hook_step1(self):
        """Gradients of update #1 via the export_grads (data-parallel) split of the
        same update: phase 0 -> critic grads -> apply -> phase 1 -> actor grads."""
        S, A, actor, critic = self.args
        tw = HipDDPG(S, A, actor, critic, **{**self.kw, "export_grads": True})
        s, a, r, d, s2 = (g(x) for x in self._first)
        L = tw.algo.learner
        L.update_phase(0, s, a, r, d, s2)
        gc = split_like(L.critic_grad, tw.algo.critic)
        L.apply(0, 1.0)
        L.update_phase(1, s, a, r, d, s2)
        ga = split_like(L.actor_grad, tw.algo.actor)
        L.apply(1, 1.0)
        self._g = (gc, ga)
        self.twin = tw

    def step1_grads(self): return self._g


class HipTD3:
    def __init__(self, S, A, actor, c1, c2, **kw):
        self.algo = TD3(logger=NullLogger(), state_dim=S, action_dim=A, device=DEV, log_every=10 ** 9, **kw).create()
        load_params(self.algo.actor, actor); load_params(self.algo.actor_target, actor)
        load_params(self.algo.critic, c1 + c2); load_params(self.algo.critic_target, c1 + c2)

    def update(self, s, a, r, d, s2, noise):
        self.algo.update(g(s), g(a), g(r), g(d), g(s2), noise=g(noise))

    def q(self, s, a, j, target=False):
        c = self.algo.critic_target if target else self.algo.critic
        return c(g(s), g(a))[j].cpu()

    def pi(self, s, target=False):
        return (self.algo.actor_target if target else self.algo.actor)(g(s)).cpu()

    def params(self, which): return cpu_params(getattr(self.algo, which))
    def adam(self, which): return hip_adam(self.algo, which)


class HipSAC:
    def __init__(self, S, A, actor, c1, c2, tune_alpha, **kw):
        self.algo = SAC(logger=NullLogger(), state_dim=S, action_dim=A, device=DEV,
                        tune_alpha=tune_alpha, log_every=10 ** 9, **kw).create()
        load_params(self.algo.actor, actor)
        load_params(self.algo.critic, c1 + c2); load_params(self.algo.critic_target, c1 + c2)

    def update(self, s, a, r, d, s2, e1, e2):
        self.algo.update(g(s), g(a), g(r), g(d), g(s2), noise=(g(e1), g(e2)))

    def q(self, s, a, j, target=False):
        c = self.algo.critic_target if target else self.algo.critic
        return c(g(s), g(a))[j].cpu()

    def pi_logp(self, s, eps):
        a, lp = self.algo.actor(g(s), eps=g(eps))
        return a.cpu(), lp.cpu()

    def params(self, which): return cpu_params(getattr(self.algo, which))
    def adam(self, which): return hip_adam(self.algo, which)

    @property
    def alpha(self): return self.algo.alpha


class HipTQC:
    def __init__(self, S, A, actor, critics, **kw):
        self.algo = TQC(logger=NullLogger(), state_dim=S, action_dim=A, device=DEV, log_every=10 ** 9, **kw).create()
        load_params(self.algo.actor, actor)
        flat = [x for c in critics for x in c]
        load_params(self.algo.critic, flat); load_params(self.algo.critic_target, flat)

    def update(self, s, a, r, d, s2, e1, e2):
        self.algo.update(g(s), g(a), g(r), g(d), g(s2), noise=(g(e1), g(e2)))

    def z(self, s, a, target=False):
        return (self.algo.critic_target if target else self.algo.critic)(g(s), g(a)).cpu()

    def pi_logp(self, s, eps):
        a, lp = self.algo.actor(g(s), eps=g(eps))
        return a.cpu(), lp.cpu()

    def params(self, which): return cpu_params(getattr(self.algo, which))
    def adam(self, which): return hip_adam(self.algo, which)

    @property
    def log_alpha(self): return float(self.algo.log_alpha.item())
