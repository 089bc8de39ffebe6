"""TD3 on the MI355X-native learner (reference:
/root/reference/src/oprl/algos/td3.py:15-146).  Twin critics share one Adam
state arena (the reference's single optimiser over DoubleCritic.parameters());
the actor step and both Polyak updates run when ``update_step % policy_freq == 0``."""
from __future__ import annotations

from dataclasses import dataclass, field

import torch as t
from torch import nn

from oprl_amd.algos.base_algorithm import HipLearner, OffPolicyAlgorithm, require_gpu
from oprl_amd.algos.nn_functions import disable_gradient
from oprl_amd.algos.nn_models import DeterministicPolicy, DoubleCritic, flatten_module_
from oprl_amd.algos.protocols import PolicyProtocol
from oprl_amd.logging import LoggerProtocol


@dataclass
class TD3(OffPolicyAlgorithm):
    logger: LoggerProtocol
    state_dim: int
    action_dim: int
    batch_size: int = 256
    policy_noise: float = 0.2
    expl_noise: float = 0.1
    noise_clip: float = 0.5
    policy_freq: int = 2
    gamma: float = 0.99
    lr_actor: float = 3e-4
    lr_critic: float = 3e-4
    max_action: float = 1.0
    tau: float = 5e-3
    log_every: int = 5000
    device: str = "cuda"
    max_batch: int = 4096
    export_grads: bool = False
    no_fuse: bool = False     # True: the generic per-net launch sequence instead of the fused kernels
    precision: str = "f32"        # "f32": exact-fp32 MFMA (parity mode); "bf16": bf16 MFMA inputs, fp32 accumulate / master / Adam (include/oprl_amd.h)

    actor: PolicyProtocol = field(init=False)
    actor_target: PolicyProtocol = field(init=False)
    critic: nn.Module = field(init=False)
    critic_target: nn.Module = field(init=False)
    learner: HipLearner = field(init=False, repr=False)
    _created: bool = False

    def create(self) -> "TD3":
        dev = require_gpu(self.device)

        def policy():
            # like the reference, max_action is NOT forwarded to the policy (td3.py:43-50)
            return DeterministicPolicy(
                state_dim=self.state_dim, action_dim=self.action_dim, hidden_units=(256, 256),
                hidden_activation=nn.ReLU(inplace=True), expl_noise=self.expl_noise,
                device=self.device).to(dev)

        def critic():
            return DoubleCritic(self.state_dim, self.action_dim, (256, 256), nn.ReLU(inplace=True)).to(dev)

        self.actor, self.actor_target = policy(), policy().eval()
        self.critic, self.critic_target = critic(), critic().eval()
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
                  beta1=0.9, beta2=0.999, adam_eps=1e-8, policy_noise=self.policy_noise,
                  noise_clip=self.noise_clip, max_action=self.max_action, policy_freq=self.policy_freq)
        self.learner = HipLearner(
            "td3", self.state_dim, self.action_dim, dev,
            actor_group=self.actor, actor_mlp=self.actor.mlp, actor_target_mlp=self.actor_target.mlp,
            actor_target_group=self.actor_target,
            critic_group=self.critic, critic_mlps=[self.critic.q1, self.critic.q2],
            critic_target_group=self.critic_target,
            critic_target_mlps=[self.critic_target.q1, self.critic_target.q2],
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
        *,
        noise: t.Tensor | None = None,
    ) -> None:
        """``noise``: optional injected N(0,1) draw [B, A] standing in for
        ``randn_like(action)`` (td3.py:98); None draws it on device (Philox)."""
        step = self.update_step
        self.learner.update(state, action, reward, done, next_state, noise0=noise)
        self._log_update(step)

    def _log_update(self, step: int) -> None:
        if step % self.log_every == 0:
            sc = self.learner.read_scalars()   # the only host sync, every log_every updates
            self.logger.log_scalar("algo/q1", sc["q1_mean"], step)
            self.logger.log_scalar("algo/q_target", sc["q_target_mean"], step)
            self.logger.log_scalar("algo/abs_q_err", sc["q1_mean"] - sc["q_target_mean"], step)
            self.logger.log_scalar("algo/critic_loss", sc["critic_loss"], step)
            if step % self.policy_freq == 0:
                self.logger.log_scalar("algo/loss_actor", sc["actor_loss"], step)
