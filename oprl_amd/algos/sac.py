"""SAC on the MI355X-native learner (reference:
/root/reference/src/oprl/algos/sac.py:16-155): tanh-Gaussian actor, twin
critics, fixed or learned temperature.  log_alpha lives on the device as a
float64 scalar (the reference's 0-dim double tensor); ``alpha`` reads it back."""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch as t
from torch import nn

from oprl_amd.algos.base_algorithm import HipLearner, OffPolicyAlgorithm, require_gpu
from oprl_amd.algos.nn_functions import disable_gradient
from oprl_amd.algos.nn_models import DoubleCritic, GaussianActor, flatten_module_
from oprl_amd.algos.protocols import PolicyProtocol
from oprl_amd.logging import LoggerProtocol


@dataclass
class SAC(OffPolicyAlgorithm):
    logger: LoggerProtocol
    state_dim: int
    action_dim: int
    batch_size: int = 256
    tune_alpha: bool = False
    gamma: float = 0.99
    lr_actor: float = 3e-4
    lr_critic: float = 3e-4
    lr_alpha: float = 1e-3
    alpha_init: float = 0.2
    target_update_coef: float = 5e-3
    device: str = "cuda"
    log_every: int = 5000
    max_batch: int = 4096
    export_grads: bool = False
    no_fuse: bool = False     # True: the generic per-net launch sequence instead of the fused kernels
    precision: str = "f32"        # "f32": exact-fp32 MFMA (parity mode); "bf16": bf16 MFMA inputs, fp32 accumulate / master / Adam (include/oprl_amd.h)

    actor: PolicyProtocol = field(init=False)
    critic: nn.Module = field(init=False)
    critic_target: nn.Module = field(init=False)
    learner: HipLearner = field(init=False, repr=False)
    _created: bool = False

    def create(self) -> "SAC":
        dev = require_gpu(self.device)
        self.actor = GaussianActor(self.state_dim, self.action_dim, (256, 256),
                                   nn.ReLU(inplace=True), device=self.device).to(dev)

        def critic():
            return DoubleCritic(self.state_dim, self.action_dim, (256, 256), nn.ReLU(inplace=True)).to(dev)

        self.critic, self.critic_target = critic(), critic().eval()
        for m in (self.actor, self.critic, self.critic_target):
            flatten_module_(m)
        self.critic_target._oprl_arena.copy_(self.critic._oprl_arena)
        for m in self.critic_target.modules():
            if hasattr(m, "mark_dirty"):
                m.mark_dirty()
        disable_gradient(self.critic_target)
        self.log_alpha = None
        if self.tune_alpha:
            self.log_alpha = t.tensor(math.log(self.alpha_init), dtype=t.float64, device=dev)
            self.target_entropy = -float(self.action_dim)
        hp = dict(gamma=self.gamma, tau=self.target_update_coef, lr_actor=self.lr_actor,
                  lr_critic=self.lr_critic, lr_alpha=self.lr_alpha, beta1=0.9, beta2=0.999,
                  adam_eps=1e-8, alpha_init=self.alpha_init, tune_alpha=int(self.tune_alpha),
                  target_entropy=-float(self.action_dim), policy_freq=1)
        self.learner = HipLearner(
            "sac", self.state_dim, self.action_dim, dev,
            actor_group=self.actor, actor_mlp=self.actor.net, actor_target_mlp=None,
            critic_group=self.critic, critic_mlps=[self.critic.q1, self.critic.q2],
            critic_target_group=self.critic_target,
            critic_target_mlps=[self.critic_target.q1, self.critic_target.q2],
            hp=hp, max_batch=self.max_batch, export_grads=self.export_grads, log_alpha=self.log_alpha,
            no_fuse=self.no_fuse, precision=self.precision)
        self._created = True
        return self

    @property
    def alpha(self) -> float:
        if self.log_alpha is not None:
            return float(self.log_alpha.exp().item())
        return self.alpha_init

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
        noise: tuple[t.Tensor, t.Tensor] | None = None,
    ) -> None:
        """``noise``: optional (eps_next [B,A], eps_current [B,A]) standing in for
        the two ``Normal(0,1).sample()`` draws (nn_models.py:213); None = Philox."""
        n0, n1 = noise if noise is not None else (None, None)
        step = self.update_step
        self.learner.update(state, action, reward, done, next_state, noise0=n0, noise1=n1)
        self._log_update(step)

    def _log_update(self, step: int) -> None:
        if step % self.log_every == 0:
            # the reference's tag set (sac.py:108-155), read from the kernels' partial sums in one host sync
            sc = self.learner.read_scalars()
            self.logger.log_scalars({
                "algo/q1": sc["q1_mean"], "algo/q_target": sc["q_target_mean"],
                "algo/abs_q_err": sc["q1_mean"] - sc["q_target_mean"],
                "algo/critic_loss": sc["critic_loss"],
            }, step)
            if self.tune_alpha:
                self.logger.log_scalar("algo/loss_alpha", sc["alpha_loss"], step)
            self.logger.log_scalars({
                "algo/loss_actor": sc["gauss_actor_loss"], "algo/alpha": sc["alpha"],
                "algo/log_pi": sc["log_pi_mean"],
            }, step)
