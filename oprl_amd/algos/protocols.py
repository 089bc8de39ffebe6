"""Structural interfaces of the algorithm layer — the Python half of the drop-in boundary.

The names and call shapes are the reference's (src/oprl/algos/protocols.py:11-41: ``PolicyProtocol``,
``AlgorithmProtocol``); the docstrings state what THIS implementation guarantees for each member, and
the last block lists what it offers on top (all optional for a caller written against the reference)."""
from __future__ import annotations

from typing import Any, Protocol

import numpy.typing as npt
import torch as t
import torch.nn as nn

from oprl_amd.logging import LoggerProtocol


class PolicyProtocol(Protocol):
    """What a trainer or an actor process needs from a policy network."""

    def explore(self, state: npt.NDArray) -> npt.NDArray:
        """One observation (any float dtype, shape ``[state_dim]``) -> one exploratory action ``[action_dim]``
        (float32, clipped to the action range).  On a GPU-resident policy this is one C call
        (``oprl_mlp_act``): no torch tensors on the way."""
        ...

    def exploit(self, state: npt.NDArray) -> npt.NDArray:
        """As ``explore`` without exploration noise (deterministic policy: tanh of the net output;
        Gaussian policy: tanh of the mean)."""
        ...

    def __call__(*args, **kwargs) -> t.Tensor:
        """Batched forward on tensors (``[B, state_dim]`` -> ``[B, action_dim]``; the Gaussian actor returns
        ``(action, log_prob)``)."""
        ...

    def state_dict(self) -> dict:
        """Same keys as the reference's modules (``mlp.nn.{0,2,4}.{weight,bias}`` / ``net.nn...``): weights
        travel between this learner, a plain CPU copy in an actor process and the reference unchanged."""
        ...


class AlgorithmProtocol(Protocol):
    """DDPG / TD3 / SAC / TQC as the trainers see them."""

    actor: PolicyProtocol
    critic: nn.Module
    logger: LoggerProtocol
    _created: bool

    def create(self) -> "AlgorithmProtocol":
        """Builds networks, targets, optimiser state and the native learner; raises on a non-GPU device or a
        missing ``liboprl_amd.so`` (there is no CPU path).  Returns ``self``."""
        ...

    def check_created(self) -> None:
        """Raises ``RuntimeError`` before ``create()``."""
        ...

    def update(
        self,
        state: t.Tensor,
        action: t.Tensor,
        reward: t.Tensor,
        done: t.Tensor,
        next_state: t.Tensor,
    ) -> None:
        """One gradient update on a minibatch (``[B, state_dim]``, ``[B, action_dim]``, ``[B, 1]`` or
        ``[B]`` rewards / dones of any real dtype, ``B <= max_batch``).  Asynchronous on the current
        stream; scalars are read back only at the algorithm's logging cadence."""
        ...

    def get_policy_state_dict(self) -> dict[str, Any]:
        return self.actor.state_dict()

    # ---- offered on top of the reference's interface --------------------------------------------------
    # update_from_buffer(replay_buffer, batch_size): sample + update as one C call (the kernels gather
    #     their own rows from the HBM replay); falls back to sample() / update() for a foreign buffer
    # state_dict() / load_state_dict(): the FULL learner state (parameters, targets, Adam moments,
    #     temperature, counters) — a restore resumes the update stream bit for bit
    # update_step: number of updates so far
