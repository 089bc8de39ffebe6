"""TQC on the MI355X-native learner (reference:
/root/reference/src/oprl/algos/tqc.py): 5 quantile critics (30->512->512->512->25),
row-wise sort + truncation of the 125 target quantiles, fused quantile-Huber
forward/backward, learned temperature.

One update is 17 launches (csrc/learner.hip critic_phase / actor_phase, csrc/layerwise.hip): the five critics run
layer by layer over the whole chip; the launches that leave CUs idle carry independent work as riding workgroups
(the online critics' first layers behind the actor's forward on s', the TD-target sort on the target pass's heads,
the actor step's forward on the critic step's heads, the narrow layers' dW tiles behind the wide dW launch, the
temperature step on the actor's dW launch, and — in step_n — the next update's minibatch rows on the
action-gradient launch).  DESIGN.md section 4.1; docs/history/DESIGN_r01-r05.md section 4.5."""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch as t
import torch.nn as nn

from oprl_amd.algos.base_algorithm import HipLearner, OffPolicyAlgorithm, require_gpu
from oprl_amd.algos.nn_models import MLP, GaussianActor, _forward_sa, flatten_module_
from oprl_amd.algos.protocols import PolicyProtocol
from oprl_amd.logging import LoggerProtocol


def quantile_huber_loss_f(quantiles: t.Tensor, samples: t.Tensor, device: str | None = None) -> t.Tensor:
    """Scalar quantile-Huber loss with torch ops — a *diagnostic* helper with the
    reference's signature (tqc.py:14-36).  ``update()`` does not call it: the
    loss gradient is produced inside the HIP slice kernel (SEED_QHUBER)."""
    delta = samples[:, None, None, :] - quantiles[:, :, :, None]
    ad = delta.abs()
    huber = t.where(ad > 1, ad - 0.5, 0.5 * delta * delta)
    n_q = quantiles.shape[2]
    tau = (t.arange(n_q, device=quantiles.device, dtype=t.float32) + 0.5) / n_q
    return ((tau[None, None, :, None] - (delta < 0).float()).abs() * huber).mean()


class QuantileQritic(nn.Module):
    def __init__(self, state_dim: int, action_dim: int, n_quantiles: int, n_nets: int) -> None:
        super().__init__()
        self.n_quantiles = n_quantiles
        self.n_nets = n_nets
        self.nets = []
        for i in range(n_nets):
            net = MLP(state_dim + action_dim, n_quantiles, (512, 512, 512), hidden_activation=nn.ReLU())
            self.add_module(f"qf{i}", net)
            self.nets.append(net)

    def forward(self, state: t.Tensor, action: t.Tensor) -> t.Tensor:
        return t.stack(tuple(_forward_sa(net, state, action) for net in self.nets), dim=1)


@dataclass
class TQC(OffPolicyAlgorithm):
    logger: LoggerProtocol
    state_dim: int
    action_dim: int
    gamma: float = 0.99
    lr_actor = 3e-4      # un-annotated class attributes, as in the reference (tqc.py:67-69)
    lr_critic = 3e-4
    lr_alpha = 3e-4
    tau: float = 0.005
    top_quantiles_to_drop: int = 2
    n_quantiles: int = 25
    n_nets: int = 5
    log_every: int = 5000
    device: str = "cuda"
    max_batch: int = 4096
    export_grads: bool = False
    precision: str = "f32"         # "f32": exact-fp32 MFMA (parity mode); "bf16": bf16 MFMA inputs, fp32 accumulate / master / Adam

    actor: PolicyProtocol = field(init=False)
    critic: QuantileQritic = field(init=False)
    critic_target: QuantileQritic = field(init=False)
    target_entropy: float = field(init=False)
    quantiles_total: int = field(init=False)
    learner: HipLearner = field(init=False, repr=False)
    _created: bool = False

    def create(self) -> "TQC":
        dev = require_gpu(self.device)
        self.target_entropy = -float(self.action_dim)
        self.actor = GaussianActor(self.state_dim, self.action_dim, hidden_units=(256, 256),
                                   hidden_activation=nn.ReLU(), device=self.device).to(dev)

        def critic():
            return QuantileQritic(self.state_dim, self.action_dim, self.n_quantiles, self.n_nets).to(dev)

        self.critic, self.critic_target = critic(), critic()
        for m in (self.actor, self.critic, self.critic_target):
            flatten_module_(m)
        self.critic_target._oprl_arena.copy_(self.critic._oprl_arena)
        for m in self.critic_target.modules():
            if hasattr(m, "mark_dirty"):
                m.mark_dirty()
        self.log_alpha = t.tensor(math.log(0.2), dtype=t.float64, device=dev)
        self.quantiles_total = self.n_quantiles * self.n_nets
        hp = dict(gamma=self.gamma, tau=self.tau, lr_actor=self.lr_actor, lr_critic=self.lr_critic,
                  lr_alpha=self.lr_alpha, beta1=0.9, beta2=0.999, adam_eps=1e-8, alpha_init=0.2,
                  tune_alpha=1, target_entropy=self.target_entropy, policy_freq=1,
                  n_quantiles=self.n_quantiles, top_quantiles_to_drop=self.top_quantiles_to_drop)
        self.learner = HipLearner(
            "tqc", self.state_dim, self.action_dim, dev,
            actor_group=self.actor, actor_mlp=self.actor.net, actor_target_mlp=None,
            critic_group=self.critic, critic_mlps=self.critic.nets,
            critic_target_group=self.critic_target, critic_target_mlps=self.critic_target.nets,
            hp=hp, max_batch=self.max_batch, export_grads=self.export_grads, log_alpha=self.log_alpha,
            precision=self.precision)
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
        noise: tuple[t.Tensor, t.Tensor] | None = None,
    ):
        n0, n1 = noise if noise is not None else (None, None)
        step = self.update_step
        self.learner.update(state, action, reward, done, next_state, noise0=n0, noise1=n1)
        self._log_update(step)

    def _log_update(self, step: int) -> None:
        if step % self.log_every == 0:
            sc = self.learner.read_scalars()
            self.logger.log_scalars({"algo/critic_loss": sc["critic_loss"],
                                     "algo/actor_loss": sc["actor_loss"]}, step)
