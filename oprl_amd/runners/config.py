"""Run configuration records (reference: /root/reference/src/oprl/runners/config.py).
Plain dataclasses: pydantic-settings is not a dependency of the learner."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class CommonParameters:
    state_dim: int
    action_dim: int
    num_steps: int
    eval_every: int = 2500
    estimate_q_every: int = 5000
    log_every: int = 2500
    device: str = "cuda"


@dataclass
class DistribConfig:
    batch_size: int = 128
    num_env_workers: int = 4
    episodes_per_worker: int = 100
    warmup_epochs: int = 16
    episode_length: int = 1000
    learner_num_waits: int = 10
    warmup_env_steps: int = 1000
