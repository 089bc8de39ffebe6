"""CPU oracle for the oprl off-policy learner hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (explicit forward / backward / Adam /
Polyak arithmetic on torch-CPU tensors, *no autograd*) of the reference's
``ReplayBuffer.sample -> algo.update`` path.  It exists so that the HIP path can
be checked on a GPU box where ``/root/reference`` does not exist.

Who may import this:  ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  The product package ``oprl_amd`` never
imports it and has no CPU fallback.

Parity pin: the reference's own tests hold no numeric vectors for this path
(SURVEY.md §4), so this oracle is pinned against outputs of the *reference
itself*, run in the build container by ``oracle/gen_golden.py`` and committed as
``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` checks every one).

Reference lines restated (all relative to /root/reference/src/oprl):
  MLP / Critic / DoubleCritic        algos/nn_models.py:27-107
  DeterministicPolicy                algos/nn_models.py:110-150
  GaussianActor / TanhNormal         algos/nn_models.py:153-214
  soft_update                        algos/nn_functions.py:5-10
  DDPG.update                        algos/ddpg.py:61-107
  TD3.update                         algos/td3.py:71-146
  SAC.update                         algos/sac.py:75-155
  TQC.update, quantile_huber_loss_f  algos/tqc.py:14-36,116-189
  torch.optim.Adam defaults          (third party, torch==2.2.2; betas .9/.999 eps 1e-8)
  EpisodicReplayBuffer               buffers/episodic_buffer.py:29-140
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch as t

LOG_STD_MIN, LOG_STD_MAX = -20.0, 2.0  # nn_models.py:11
F32 = t.float32


# --------------------------------------------------------------------------- #
# parameters                                                                  #
# --------------------------------------------------------------------------- #
def make_mlp_params(rs: np.random.RandomState, dims: list[int]) -> list[t.Tensor]:
    """[W0,b0,W1,b1,...] drawn U(+-1/sqrt(in)) from a frozen legacy RandomState
    stream (torch's default nn.Linear distribution; the *stream* is numpy's so
    fixtures regenerate identically everywhere)."""
    out = []
    for i in range(len(dims) - 1):
        fan_in, fan_out = dims[i], dims[i + 1]
        bound = 1.0 / math.sqrt(fan_in)
        w = rs.uniform(-bound, bound, size=(fan_out, fan_in)).astype(np.float32)
        b = rs.uniform(-bound, bound, size=(fan_out,)).astype(np.float32)
        out += [t.from_numpy(w), t.from_numpy(b)]
    return out


def clone_params(p: list[t.Tensor]) -> list[t.Tensor]:
    return [x.clone() for x in p]


# --------------------------------------------------------------------------- #
# MLP forward / backward (nn_models.py:84-107; ReLU hidden, identity output)  #
# --------------------------------------------------------------------------- #
# Emulation of the HIP library's OPRL_PREC_BF16 mode (include/oprl_amd.h): both operands of every
# forward / backward GEMM are rounded to bf16 (round-to-nearest-even, what v_cvt_pk_bf16_f32 does) and
# multiplied / accumulated in fp32; master weights, biases, activations, dW, Adam and Polyak stay fp32.
# Not part of the reference (which is fp32 throughout): it pins what the bf16 kernels claim to compute,
# the fp32 oracle measures how far that is from the reference.
GEMM_BF16 = False
GEMM_BF16_MIN_DIM = 0     # only layers whose fan-in AND fan-out reach this are rounded
# the WHOLE-UPDATE form of a bf16 DDPG learner (k_ddpg_chain<PrecBF16>, round 5): the critic pass of the actor step runs on
# clusters of EIGHT (its partial dz are rounded per eighth of the hidden columns), and the actor's own backward — the output
# layer's rows times du in the tiles, the first layer through unit-seed rows — is formed in exact fp32
GEMM_BF16_CHAIN = False


class bf16_gemm:
    """``with bf16_gemm(): ...`` — run the oracle with bf16-rounded GEMM operands.  ``min_dim=512``:
    only the 512 x 512 layers (what a bf16 TQC learner does: its layer-wise critic kernels run the hidden
    layers in bf16; the first layer, the heads and the 256-wide actor stay fp32)."""

    def __init__(self, on: bool = True, min_dim: int = 0, chain: bool = False):
        self.on, self.min_dim, self.chain = on, min_dim, chain

    def __enter__(self):
        global GEMM_BF16, GEMM_BF16_MIN_DIM, GEMM_BF16_CHAIN
        self.saved = (GEMM_BF16, GEMM_BF16_MIN_DIM, GEMM_BF16_CHAIN)
        GEMM_BF16, GEMM_BF16_MIN_DIM, GEMM_BF16_CHAIN = self.on, self.min_dim, self.chain
        return self

    def __exit__(self, *exc):
        global GEMM_BF16, GEMM_BF16_MIN_DIM, GEMM_BF16_CHAIN
        GEMM_BF16, GEMM_BF16_MIN_DIM, GEMM_BF16_CHAIN = self.saved
        return False


def _q(x: t.Tensor, w: Optional[t.Tensor] = None) -> t.Tensor:
    """x rounded to bf16 if the emulation is on (and the layer with weight ``w`` is wide enough)."""
    if not GEMM_BF16 or (w is not None and min(w.shape) < GEMM_BF16_MIN_DIM):
        return x
    return x.to(t.bfloat16).to(F32)


def mlp_forward(p: list[t.Tensor], x: t.Tensor) -> list[t.Tensor]:
    """Returns [x0, x1, ..., x_{L-1}, out]; x_l is the *input* of layer l."""
    acts = [x]
    n_layers = len(p) // 2
    for l in range(n_layers):
        z = t.addmm(p[2 * l + 1], _q(acts[-1], p[2 * l]), _q(p[2 * l], p[2 * l]).t())
        if l < n_layers - 1:
            z = t.relu(z)
        acts.append(z)
    return acts


def mlp_backward(p: list[t.Tensor], acts: list[t.Tensor], dout: t.Tensor,
                 need_dx: bool = False, need_dw: bool = True):
    """Given d(loss)/d(out) returns (grads [dW0,db0,...], dx0 or None)."""
    n_layers = len(p) // 2
    grads: list[Optional[t.Tensor]] = [None] * (2 * n_layers)
    dz = dout
    dx = None
    # bf16 emulation of a scalar-output net (a critic): the kernels run its backward with a UNIT seed
    # (the backward is linear in the per-row seed) and apply the seed per row afterwards in fp32
    # (csrc/tp4.h tp4_scalar_fb), so the seed itself is never rounded to bf16
    row_scale = None
    parts = None
    if GEMM_BF16 and GEMM_BF16_MIN_DIM == 0 and dout.shape[1] == 1 and n_layers > 1:
        row_scale = dout
        dz = t.ones_like(dout)
    for l in range(n_layers - 1, -1, -1):
        dz_s = dz if row_scale is None else dz * row_scale
        if need_dw:
            grads[2 * l] = dz_s.t() @ acts[l]
            grads[2 * l + 1] = dz_s.sum(0)
        if l > 0 or need_dx:
            if row_scale is not None and l == n_layers - 1:
                dx = dz * _q(p[2 * l])          # [B,1] x [1,W]: an elementwise product, no GEMM
            elif row_scale is not None and l == 0 and parts is not None:
                # the input gradient of a scalar net on a 4-CU cluster: every member rounds ITS partial
                # dz (contraction over its quarter of the hidden columns) to bf16 for the next GEMM and
                # the members' products are summed in fp32 (csrc/tp4.h: dact quarters + all-reduce)
                dx = sum(_q(pm) @ _q(p[0]) for pm in parts)
            elif GEMM_BF16_CHAIN and row_scale is None and GEMM_BF16_MIN_DIM == 0:
                dx = dz @ p[2 * l]               # (the chain form's actor backward: exact fp32)
            else:
                dx = _q(dz, p[2 * l]) @ _q(p[2 * l], p[2 * l])
        if l > 0:
            mask = (acts[l] > 0).to(F32)     # threshold_backward on the ReLU output
            parts = None
            if row_scale is not None and n_layers == 3 and l == 1 and dz.shape[1] % 8 == 0:
                # (members of the cluster: four — eight for the chain form's critic pass, the backward that wants dx only)
                nm = 8 if (GEMM_BF16_CHAIN and need_dx and not need_dw) else 4
                w4 = dz.shape[1] // nm
                parts = [(_q(dz[:, m * w4:(m + 1) * w4]) @ _q(p[2 * l][m * w4:(m + 1) * w4, :])) * mask
                         for m in range(nm)]
                dx = parts[0]
                for m in range(1, nm):
                    dx = dx + parts[m]           # member order, as k_dw_adam sums them
                dz = dx
            else:
                dz = dx * mask
    if row_scale is not None and dx is not None:
        dx = dx * row_scale
    return grads, (dx if need_dx else None)


# --------------------------------------------------------------------------- #
# Adam (torch.optim.Adam defaults) and Polyak                                 #
# --------------------------------------------------------------------------- #
@dataclass
class Adam:
    lr: float
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    step_count: int = 0
    m: list[t.Tensor] = field(default_factory=list)
    v: list[t.Tensor] = field(default_factory=list)

    def step(self, params: list[t.Tensor], grads: list[t.Tensor]) -> None:
        if not self.m:
            self.m = [t.zeros_like(x) for x in params]
            self.v = [t.zeros_like(x) for x in params]
        self.step_count += 1
        bc1 = 1.0 - self.beta1 ** self.step_count
        bc2 = 1.0 - self.beta2 ** self.step_count
        step_size = self.lr / bc1
        bc2_sqrt = math.sqrt(bc2)
        for prm, g, m, v in zip(params, grads, self.m, self.v):
            m.add_((g - m) * (1.0 - self.beta1))            # lerp
            v.mul_(self.beta2).add_(g * g * (1.0 - self.beta2))
            denom = v.sqrt() / bc2_sqrt + self.eps
            prm.sub_(step_size * (m / denom))


def polyak(target: list[t.Tensor], source: list[t.Tensor], tau: float) -> None:
    """ddpg.py:72-84 / nn_functions.py:5-10 (bit-identical spellings in fp32)."""
    for tg, s in zip(target, source):
        tg.mul_(1.0 - tau).add_(tau * s)


# --------------------------------------------------------------------------- #
# policy heads                                                                #
# --------------------------------------------------------------------------- #
def det_policy_forward(p, s):
    acts = mlp_forward(p, s)
    return t.tanh(acts[-1]), acts


def logsigmoid(x: t.Tensor) -> t.Tensor:
    return t.nn.functional.logsigmoid(x)


def gaussian_forward(p, s, eps, action_dim):
    """nn_models.py:168-178,197-214 train-mode forward with injected eps."""
    acts = mlp_forward(p, s)
    out = acts[-1]
    mu, log_std_raw = out[:, :action_dim], out[:, action_dim:]
    log_std = log_std_raw.clamp(LOG_STD_MIN, LOG_STD_MAX)
    std = t.exp(log_std)
    u = mu + std * eps
    a = t.tanh(u)
    normal_lp = -((u - mu) ** 2) / (2 * std * std) - log_std - math.log(math.sqrt(2 * math.pi))
    log_det = 2 * math.log(2) + logsigmoid(2 * u) + logsigmoid(-2 * u)
    logp = (normal_lp - log_det).sum(1, keepdim=True)
    cache = dict(acts=acts, mu=mu, log_std_raw=log_std_raw, std=std, u=u, a=a, eps=eps)
    return a, logp, cache


def gaussian_backward_seed(cache, d_a, d_logp, action_dim):
    """d(loss)/d(net output [B,2A]) from d/d(action) [B,A] and d/d(logp) [B,1].

    Derivation: with u = mu + sigma*eps the Gaussian density
    terms cancel for mu; d logp/d u_j = 2 tanh(u_j) through log_det only;
    d logp/d log_sigma_j = -1 + 2 tanh(u_j) sigma_j eps_j; clamp kills the
    log_sigma gradient outside [-20, 2]."""
    a, std, eps = cache["a"], cache["std"], cache["eps"]
    du = d_a * (1 - a * a) + d_logp * (2 * a)
    d_mu = du
    d_ls = du * std * eps - d_logp
    raw = cache["log_std_raw"]
    mask = ((raw >= LOG_STD_MIN) & (raw <= LOG_STD_MAX)).to(F32)
    return t.cat([d_mu, d_ls * mask], dim=1)


def q_forward(p, s, a):
    return mlp_forward(p, t.cat([s, a], dim=1))


# --------------------------------------------------------------------------- #
# DDPG (algos/ddpg.py:61-107)                                                 #
# --------------------------------------------------------------------------- #
class DDPGOracle:
    def __init__(self, state_dim, action_dim, actor, critic, gamma=0.99, tau=5e-3,
                 lr_actor=3e-4, lr_critic=3e-4):
        self.S, self.A = state_dim, action_dim
        self.gamma, self.tau = gamma, tau
        self.actor, self.critic = clone_params(actor), clone_params(critic)
        self.actor_target, self.critic_target = clone_params(actor), clone_params(critic)
        self.opt_actor, self.opt_critic = Adam(lr_actor), Adam(lr_critic)
        self.last: dict = {}

    def update(self, s, a, r, d, s2):
        B = s.shape[0]
        d = d.to(F32)
        a2, _ = det_policy_forward(self.actor_target, s2)
        q_next = q_forward(self.critic_target, s2, a2)[-1]
        y = r + (1.0 - d) * self.gamma * q_next
        acts = q_forward(self.critic, s, a)
        q = acts[-1]
        critic_loss = ((q - y) ** 2).mean()
        dq = 2.0 * (q - y) / B
        g_c, _ = mlp_backward(self.critic, acts, dq)
        self.opt_critic.step(self.critic, g_c)

        pi, a_acts = det_policy_forward(self.actor, s)
        c_acts = q_forward(self.critic, s, pi)
        actor_loss = -c_acts[-1].mean()
        dqa = t.full_like(c_acts[-1], -1.0 / B)
        _, dx = mlp_backward(self.critic, c_acts, dqa, need_dx=True, need_dw=False)
        da = dx[:, self.S:]
        du = da * (1 - pi * pi)
        g_a, _ = mlp_backward(self.actor, a_acts, du)
        self.opt_actor.step(self.actor, g_a)

        polyak(self.critic_target, self.critic, self.tau)
        polyak(self.actor_target, self.actor, self.tau)
        self.last = dict(q=q, y=y, critic_loss=critic_loss, actor_loss=actor_loss,
                         g_critic=g_c, g_actor=g_a)


# --------------------------------------------------------------------------- #
# TD3 (algos/td3.py:71-146)                                                   #
# --------------------------------------------------------------------------- #
class TD3Oracle:
    def __init__(self, state_dim, action_dim, actor, critic1, critic2, gamma=0.99,
                 tau=5e-3, lr_actor=3e-4, lr_critic=3e-4, policy_noise=0.2,
                 noise_clip=0.5, policy_freq=2, max_action=1.0):
        self.S, self.A = state_dim, action_dim
        self.gamma, self.tau = gamma, tau
        self.policy_noise, self.noise_clip = policy_noise, noise_clip
        self.policy_freq, self.max_action = policy_freq, max_action
        self.actor = clone_params(actor)
        self.actor_target = clone_params(actor)
        # one optimiser over q1 then q2 parameters (DoubleCritic.parameters() order)
        self.critic = clone_params(critic1) + clone_params(critic2)
        self.critic_target = clone_params(self.critic)
        self.n1 = len(critic1)
        self.opt_actor, self.opt_critic = Adam(lr_actor), Adam(lr_critic)
        self.update_step = 0
        self.last: dict = {}

    def _q(self, params, j):
        return params[:self.n1] if j == 0 else params[self.n1:]

    def update(self, s, a, r, d, s2, noise):
        """noise: the N(0,1) draw of td3.py:98 (randn_like(action))."""
        B = s.shape[0]
        d = d.to(F32)
        acts1 = q_forward(self._q(self.critic, 0), s, a)
        acts2 = q_forward(self._q(self.critic, 1), s, a)
        q1, q2 = acts1[-1], acts2[-1]
        n = (noise * self.policy_noise).clamp(-self.noise_clip, self.noise_clip)
        a2 = (det_policy_forward(self.actor_target, s2)[0] + n).clamp(-self.max_action, self.max_action)
        q1n = q_forward(self._q(self.critic_target, 0), s2, a2)[-1]
        q2n = q_forward(self._q(self.critic_target, 1), s2, a2)[-1]
        y = r + (1.0 - d) * self.gamma * t.min(q1n, q2n)
        loss = ((q1 - y) ** 2).mean() + ((q2 - y) ** 2).mean()
        g1, _ = mlp_backward(self._q(self.critic, 0), acts1, 2.0 * (q1 - y) / B)
        g2, _ = mlp_backward(self._q(self.critic, 1), acts2, 2.0 * (q2 - y) / B)
        self.opt_critic.step(self.critic, g1 + g2)
        self.last = dict(q1=q1, q2=q2, y=y, critic_loss=loss, g_critic=g1 + g2)

        if self.update_step % self.policy_freq == 0:
            pi, a_acts = det_policy_forward(self.actor, s)
            c_acts = q_forward(self._q(self.critic, 0), s, pi)
            self.last["actor_loss"] = -c_acts[-1].mean()
            dqa = t.full_like(c_acts[-1], -1.0 / B)
            _, dx = mlp_backward(self._q(self.critic, 0), c_acts, dqa, need_dx=True, need_dw=False)
            du = dx[:, self.S:] * (1 - pi * pi)
            g_a, _ = mlp_backward(self.actor, a_acts, du)
            self.opt_actor.step(self.actor, g_a)
            self.last["g_actor"] = g_a
            polyak(self.critic_target, self.critic, self.tau)
            polyak(self.actor_target, self.actor, self.tau)
        self.update_step += 1


# --------------------------------------------------------------------------- #
# SAC (algos/sac.py:75-155)                                                   #
# --------------------------------------------------------------------------- #
class SACOracle:
    def __init__(self, state_dim, action_dim, actor, critic1, critic2, gamma=0.99,
                 tau=5e-3, lr_actor=3e-4, lr_critic=3e-4, lr_alpha=1e-3,
                 alpha_init=0.2, tune_alpha=False):
        self.S, self.A = state_dim, action_dim
        self.gamma, self.tau = gamma, tau
        self.actor = clone_params(actor)
        self.critic = clone_params(critic1) + clone_params(critic2)
        self.critic_target = clone_params(self.critic)
        self.n1 = len(critic1)
        self.opt_actor, self.opt_critic = Adam(lr_actor), Adam(lr_critic)
        self.alpha = float(alpha_init)
        self.tune_alpha = tune_alpha
        if tune_alpha:
            # sac.py:65-70: t.tensor(np.log(alpha)) is float64
            self.log_alpha = t.tensor(np.log(self.alpha), dtype=t.float64)
            self.opt_alpha = Adam(lr_alpha)
            self.target_entropy = -float(action_dim)
        self.update_step = 0
        self.last: dict = {}

    def _q(self, params, j):
        return params[:self.n1] if j == 0 else params[self.n1:]

    def update(self, s, a, r, d, s2, eps_next, eps_cur):
        B = s.shape[0]
        d = d.to(F32)
        acts1 = q_forward(self._q(self.critic, 0), s, a)
        acts2 = q_forward(self._q(self.critic, 1), s, a)
        q1, q2 = acts1[-1], acts2[-1]
        a2, logp2, _ = gaussian_forward(self.actor, s2, eps_next, self.A)
        q1n = q_forward(self._q(self.critic_target, 0), s2, a2)[-1]
        q2n = q_forward(self._q(self.critic_target, 1), s2, a2)[-1]
        q_next = t.min(q1n, q2n) - self.alpha * logp2
        y = r + (1.0 - d) * self.gamma * q_next
        loss = ((q1 - y) ** 2).mean() + ((q2 - y) ** 2).mean()
        g1, _ = mlp_backward(self._q(self.critic, 0), acts1, 2.0 * (q1 - y) / B)
        g2, _ = mlp_backward(self._q(self.critic, 1), acts2, 2.0 * (q2 - y) / B)
        self.opt_critic.step(self.critic, g1 + g2)

        pi, logp, cache = gaussian_forward(self.actor, s, eps_cur, self.A)
        c1 = q_forward(self._q(self.critic, 0), s, pi)
        c2 = q_forward(self._q(self.critic, 1), s, pi)
        qa1, qa2 = c1[-1], c2[-1]
        actor_loss = self.alpha * logp.mean() - t.min(qa1, qa2).mean()
        w1 = (qa1 < qa2).to(F32) + 0.5 * (qa1 == qa2).to(F32)
        _, dx1 = mlp_backward(self._q(self.critic, 0), c1, -w1 / B, need_dx=True, need_dw=False)
        _, dx2 = mlp_backward(self._q(self.critic, 1), c2, -(1 - w1) / B, need_dx=True, need_dw=False)
        d_a = dx1[:, self.S:] + dx2[:, self.S:]
        d_logp = t.full_like(logp, self.alpha / B)
        dout = gaussian_backward_seed(cache, d_a, d_logp, self.A)
        g_a, _ = mlp_backward(self.actor, cache["acts"], dout)
        self.opt_actor.step(self.actor, g_a)
        self.last = dict(q1=q1, q2=q2, y=y, critic_loss=loss, actor_loss=actor_loss,
                         logp=logp, g_critic=g1 + g2, g_actor=g_a)

        if self.tune_alpha:
            g_alpha = -(self.target_entropy + logp.mean().to(t.float64))
            la = [self.log_alpha.reshape(1)]
            self.opt_alpha.step(la, [g_alpha.reshape(1)])
            self.log_alpha = la[0].reshape(())
            self.alpha = float(self.log_alpha.exp().item())

        polyak(self.critic_target, self.critic, self.tau)
        self.update_step += 1


# --------------------------------------------------------------------------- #
# TQC (algos/tqc.py)                                                          #
# --------------------------------------------------------------------------- #
def quantile_huber_loss(quantiles: t.Tensor, samples: t.Tensor):
    """tqc.py:14-36.  Returns (loss, dloss/dquantiles)."""
    B, N, Q = quantiles.shape
    M = samples.shape[1]
    delta = samples[:, None, None, :] - quantiles[:, :, :, None]
    ad = delta.abs()
    huber = t.where(ad > 1, ad - 0.5, delta * delta * 0.5)
    tau = t.arange(Q).float() / Q + 1 / 2 / Q
    wgt = (tau[None, None, :, None] - (delta < 0).float()).abs()
    loss = (wgt * huber).mean()
    dh = t.where(ad > 1, t.sign(delta), delta)
    dz = -(wgt * dh).sum(-1) / float(B * N * Q * M)
    return loss, dz


class TQCOracle:
    def __init__(self, state_dim, action_dim, actor, critics: list[list[t.Tensor]],
                 gamma=0.99, tau=0.005, lr_actor=3e-4, lr_critic=3e-4, lr_alpha=3e-4,
                 top_quantiles_to_drop=2, n_quantiles=25):
        self.S, self.A = state_dim, action_dim
        self.gamma, self.tau = gamma, tau
        self.drop, self.Q, self.N = top_quantiles_to_drop, n_quantiles, len(critics)
        self.actor = clone_params(actor)
        self.critics = [clone_params(c) for c in critics]
        self.critics_target = [clone_params(c) for c in critics]
        self.opt_actor, self.opt_critic, self.opt_alpha = Adam(lr_actor), Adam(lr_critic), Adam(lr_alpha)
        self.log_alpha = t.tensor(np.log(0.2), dtype=t.float64)  # tqc.py:105
        self.target_entropy = -float(action_dim)
        self.update_step = 0
        self.last: dict = {}

    def _flat(self, nets):
        return [x for n in nets for x in n]

    def update(self, s, a, r, d, s2, eps_next, eps_cur):
        B = s.shape[0]
        d = d.to(F32)
        alpha = self.log_alpha.exp().to(F32)
        a2, logp2, _ = gaussian_forward(self.actor, s2, eps_next, self.A)
        next_z = t.stack([q_forward(c, s2, a2)[-1] for c in self.critics_target], dim=1)
        sorted_z, _ = t.sort(next_z.reshape(B, -1))
        part = sorted_z[:, : self.N * self.Q - self.drop]
        target = r + (1 - d) * self.gamma * (part - alpha * logp2)
        c_acts = [q_forward(c, s, a) for c in self.critics]
        cur_z = t.stack([x[-1] for x in c_acts], dim=1)
        critic_loss, dz = quantile_huber_loss(cur_z, target)
        grads = []
        for n in range(self.N):
            g, _ = mlp_backward(self.critics[n], c_acts[n], dz[:, n, :].contiguous())
            grads += g
        self.opt_critic.step(self._flat(self.critics), grads)
        polyak(self._flat(self.critics_target), self._flat(self.critics), self.tau)

        pi, logp, cache = gaussian_forward(self.actor, s, eps_cur, self.A)
        alpha_grad = -(logp + self.target_entropy).mean().to(t.float64)
        d_a = t.zeros(B, self.A)
        zsum = t.zeros(B, 1)
        for n in range(self.N):
            acts = q_forward(self.critics[n], s, pi)
            zsum += acts[-1].mean(1, keepdim=True)
            dzn = t.full_like(acts[-1], -1.0 / (B * self.N * self.Q))
            _, dx = mlp_backward(self.critics[n], acts, dzn, need_dx=True, need_dw=False)
            d_a += dx[:, self.S:]
        actor_loss = (alpha * logp - zsum / self.N).mean()
        d_logp = t.full_like(logp, float(alpha) / B)
        dout = gaussian_backward_seed(cache, d_a, d_logp, self.A)
        g_a, _ = mlp_backward(self.actor, cache["acts"], dout)
        self.opt_actor.step(self.actor, g_a)
        la = [self.log_alpha.reshape(1)]
        self.opt_alpha.step(la, [alpha_grad.reshape(1)])
        self.log_alpha = la[0].reshape(())
        self.last = dict(cur_z=cur_z, target=target, critic_loss=critic_loss, dz=dz,
                         actor_loss=actor_loss, logp=logp, g_critic=grads, g_actor=g_a)
        self.update_step += 1


# --------------------------------------------------------------------------- #
# Episodic replay buffer (buffers/episodic_buffer.py:29-140)                  #
# --------------------------------------------------------------------------- #
class ReplayOracle:
    """Index / eviction semantics of EpisodicReplayBuffer with plain numpy
    storage (never-written slots hold ``fill`` so stale reads are observable)."""

    def __init__(self, buffer_size_transitions, state_dim, action_dim,
                 max_episode_lenth=1000, fill=0.0):
        self.cap = buffer_size_transitions
        self.L = max_episode_lenth
        self.E = buffer_size_transitions // max_episode_lenth
        self.states = np.full((self.E, self.L + 1, state_dim), fill, np.float32)
        self.actions = np.full((self.E, self.L, action_dim), fill, np.float32)
        self.rewards = np.full((self.E, self.L, 1), fill, np.float32)
        self.dones = np.full((self.E, self.L, 1), fill, np.float32)
        self.ep_lens = [0] * self.E
        self.ep_pointer = 0
        self.episodes_counter = 1
        self.n = 0

    def add_transition(self, state, action, reward, done, episode_done=None):
        e, l = self.ep_pointer, self.ep_lens[self.ep_pointer]
        self.states[e, l] = state
        self.actions[e, l] = action
        self.rewards[e, l] = reward
        self.dones[e, l] = float(done)
        self.ep_lens[e] += 1
        self.n = min(self.n + 1, self.cap)
        if episode_done:
            self._inc_episode()

    def _inc_episode(self):
        self.ep_pointer = (self.ep_pointer + 1) % self.E
        self.episodes_counter = min(self.episodes_counter + 1, self.E)
        self.n -= self.ep_lens[self.ep_pointer]
        self.ep_lens[self.ep_pointer] = 0

    def add_episode(self, episode):
        for s, a, r, d, _ in episode:
            self.add_transition(s, a, r, d, episode_done=d)
        self._inc_episode()

    def inds_to_episodic(self, inds):
        lens = np.asarray(self.ep_lens[: self.episodes_counter], dtype=np.int64)
        ends = np.cumsum(lens)
        starts = ends - lens
        ep = np.searchsorted(ends, inds, side="right")  # first episode with end > ind
        ep = np.where(ep >= len(ends), 0, ep)            # argmin-of-all-True quirk -> 0
        return ep, inds - starts[ep]

    def gather(self, inds):
        e, s = self.inds_to_episodic(np.asarray(inds, dtype=np.int64))
        return (self.states[e, s], self.actions[e, s], self.rewards[e, s],
                self.dones[e, s], self.states[e, s + 1])

    @property
    def last_episode_length(self):
        return self.ep_lens[self.ep_pointer]

    def __len__(self):
        return self.n
