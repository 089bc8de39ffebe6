"""Run configuration records: field names and defaults of the reference's (src/oprl/runners/config.py),
as plain dataclasses — pydantic-settings is not a dependency of the learner — and with ``device``
defaulting to the only place this learner runs."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class CommonParameters:
    """Single-process training (runners/train.py)."""

    state_dim: int
    action_dim: int
    num_steps: int                 # environment steps = updates after the buffer holds one batch
    eval_every: int = 2500         # greedy evaluation cadence, in environment steps
    estimate_q_every: int = 5000   # Q-value probe cadence (0: off)
    log_every: int = 2500          # stdout cadence
    device: str = "cuda"           # (the reference: "cpu"; there is no CPU path here)


@dataclass
class DistribConfig:
    """Distributed training (runners/train_distrib.py): CPU actors feeding one GPU learner."""

    batch_size: int = 128
    num_env_workers: int = 4            # actor processes
    episodes_per_worker: int = 100      # = epochs: the learner takes one episode per actor per epoch
    warmup_epochs: int = 16             # epochs of data collection before the first update
    episode_length: int = 1000          # steps per episode; updates per epoch = episode_length x num_env_workers
    learner_num_waits: int = 10         # seconds without an episode after which the learner gives up
    warmup_env_steps: int = 1000        # uniform random actions in every actor for this many steps
