"""CPU stand-in for the HIP learner's export_grads interface (update_phase /
apply / flat gradient tensors), built on the oracle — lets the data-parallel
HOST logic (oprl_amd.parallel) run under gloo without a GPU.  Test-only."""
from __future__ import annotations

import torch as t

from oracle import oprl_oracle as orc


def _flat(ts):
    return t.cat([x.reshape(-1) for x in ts])


def _unflat(flat, like):
    out, off = [], 0
    for x in like:
        out.append(flat[off:off + x.numel()].view(x.shape))
        off += x.numel()
    return out


class OracleDDPGEngine:
    export_grads = True
    algo_name = "ddpg"
    policy_freq = 1
    log_alpha = None
    log_alpha_grad = None

    def __init__(self, S, A, actor, critic):
        self.o = orc.DDPGOracle(S, A, actor, critic)
        o = self.o
        # flat arenas aliasing the oracle's parameter lists (views, updated in place)
        self.actor_arena, self.critic_arena = _flat(o.actor), _flat(o.critic)
        o.actor, o.critic = _unflat(self.actor_arena, o.actor), _unflat(self.critic_arena, o.critic)
        self._at, self._ct = _flat(o.actor_target), _flat(o.critic_target)
        o.actor_target, o.critic_target = _unflat(self._at, o.actor_target), _unflat(self._ct, o.critic_target)
        self.actor_m, self.actor_v = t.zeros_like(self.actor_arena), t.zeros_like(self.actor_arena)
        self.critic_m, self.critic_v = t.zeros_like(self.critic_arena), t.zeros_like(self.critic_arena)
        o.opt_actor.m, o.opt_actor.v = _unflat(self.actor_m, o.actor), _unflat(self.actor_v, o.actor)
        o.opt_critic.m, o.opt_critic.v = _unflat(self.critic_m, o.critic), _unflat(self.critic_v, o.critic)
        self.actor_grad, self.critic_grad = t.zeros_like(self.actor_arena), t.zeros_like(self.critic_arena)
        self.update_count = 0

    def target_arenas(self):
        return [self._ct, self._at]

    def update_phase(self, phase, s, a, r, d, s2, noise0=None, noise1=None):
        o, B = self.o, s.shape[0]
        if phase == 0:
            a2, _ = orc.det_policy_forward(o.actor_target, s2)
            y = r + (1.0 - d.to(t.float32)) * o.gamma * orc.q_forward(o.critic_target, s2, a2)[-1]
            acts = orc.q_forward(o.critic, s, a)
            g, _ = orc.mlp_backward(o.critic, acts, 2.0 * (acts[-1] - y) / B)
            self.critic_grad.copy_(_flat(g))
        else:
            pi, a_acts = orc.det_policy_forward(o.actor, s)
            c_acts = orc.q_forward(o.critic, s, pi)
            _, dx = orc.mlp_backward(o.critic, c_acts, t.full_like(c_acts[-1], -1.0 / B), need_dx=True, need_dw=False)
            g, _ = orc.mlp_backward(o.actor, a_acts, dx[:, o.S:] * (1 - pi * pi))
            self.actor_grad.copy_(_flat(g))
            self.update_count += 1

    def apply(self, phase, scale):
        o = self.o
        if phase == 0:
            o.opt_critic.step(o.critic, _unflat(self.critic_grad * scale, o.critic))
            orc.polyak(o.critic_target, o.critic, o.tau)
        else:
            o.opt_actor.step(o.actor, _unflat(self.actor_grad * scale, o.actor))
            orc.polyak(o.actor_target, o.actor, o.tau)
