"""DDPG on the MI355X-native learner.

Same dataclass fields, ``create()`` / ``update()`` contract and attributes as
the reference (/root/reference/src/oprl/algos/ddpg.py:16-107); ``update()`` is
one call into liboprl_amd.so — no host sync — instead of autograd + two torch Adam steps + 12 Polyak
tensor ops: ONE launch for the whole update in every arithmetic mode (k_ddpg_chain — through
``learner.step_n`` up to 32 updates per launch)."""
from __future__ import annotations

from dataclasses import dataclass, field

import torch as t
from torch import nn

from oprl_amd.algos.base_algorithm import HipLearner, OffPolicyAlgorithm, require_gpu
from oprl_amd.algos.nn_functions import disable_gradient
from oprl_amd.algos.nn_models import Critic, DeterministicPolicy, flatten_module_
from oprl_amd.algos.protocols import PolicyProtocol
from oprl_amd.logging import LoggerProtocol


@dataclass
class DDPG(OffPolicyAlgorithm):
    logger: LoggerProtocol
    state_dim: int
    action_dim: int
    expl_noise: float = 0.1
    gamma: float = 0.99
    lr_actor: float = 3e-4
    lr_critic: float = 3e-4
    tau: float = 5e-3
    batch_size: int = 256          # unused, as in the reference (ddpg.py:26)
    max_action: float = 1.
    device: str = "cuda"
    max_batch: int = 4096          # rows the HIP workspace is sized for
    export_grads: bool = False     # data-parallel learner: reduce grads between phases
    no_fuse: bool = False          # force the generic per-net launch sequence (tests / A-B)
    precision: str = "f32"        # "f32": exact-fp32 MFMA (parity mode, no input range); "x2": fp32 as fp16 hi + lo (parity mode, |obs| < 4094: leaving the range raises); "bf16": bf16 MFMA inputs, fp32 accumulate / master / Adam (include/oprl_amd.h)

    actor: PolicyProtocol = field(init=False)
    actor_target: PolicyProtocol = field(init=False)
    critic: nn.Module = field(init=False)
    critic_target: nn.Module = field(init=False)
    learner: HipLearner = field(init=False, repr=False)
    _created: bool = False

    def create(self) -> "DDPG":
        dev = require_gpu(self.device)

        def policy():
            return DeterministicPolicy(
                state_dim=self.state_dim, action_dim=self.action_dim, hidden_units=(256, 256),
                hidden_activation=nn.ReLU(inplace=True), expl_noise=self.expl_noise,
                max_action=self.max_action, device=self.device).to(dev)

        self.actor = policy()
        self.actor_target = policy()
        self.critic = Critic(self.state_dim, self.action_dim).to(dev)
        self.critic_target = Critic(self.state_dim, self.action_dim).to(dev)
        for m in (self.actor, self.actor_target, self.critic, self.critic_target):
            flatten_module_(m)
        self.actor_target._oprl_arena.copy_(self.actor._oprl_arena)
        for m in self.actor_target.modules():
            if hasattr(m, "mark_dirty"):
                m.mark_dirty()
        self.critic_target._oprl_arena.copy_(self.critic._oprl_arena)
        for m in self.critic_target.modules():
            if hasattr(m, "mark_dirty"):
                m.mark_dirty()
        disable_gradient(self.actor_target)
        disable_gradient(self.critic_target)
        hp = dict(gamma=self.gamma, tau=self.tau, lr_actor=self.lr_actor, lr_critic=self.lr_critic,
                  beta1=0.9, beta2=0.999, adam_eps=1e-8, max_action=self.max_action, policy_freq=1)
        self.learner = HipLearner(
            "ddpg", self.state_dim, self.action_dim, dev,
            actor_group=self.actor, actor_mlp=self.actor.mlp, actor_target_mlp=self.actor_target.mlp,
            actor_target_group=self.actor_target,
            critic_group=self.critic, critic_mlps=[self.critic.q1],
            critic_target_group=self.critic_target, critic_target_mlps=[self.critic_target.q1],
            hp=hp, max_batch=self.max_batch, export_grads=self.export_grads, no_fuse=self.no_fuse, precision=self.precision)
        self._created = True
        return self

    @property
    def update_step(self) -> int:
        return self.learner.update_count if self._created else 0

    def update(
        self,
        state: t.Tensor,
        action: t.Tensor,
        reward: t.Tensor,
        done: t.Tensor,
        next_state: t.Tensor,
    ) -> None:
        self.learner.update(state, action, reward, done, next_state)
