"""Shared body of the single-process training scripts (configs/ddpg.py, td3.py, sac.py, tqc.py): the
factories ``run_training`` wants — environment, algorithm, replay buffer, logger — built from the command
line, importing everything through the ``oprl`` alias package exactly as a reference config script does."""
from __future__ import annotations

import sys
from dataclasses import dataclass
from pathlib import Path
from typing import Callable

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from oprl.buffers.episodic_buffer import EpisodicReplayBuffer  # noqa: E402
from oprl.environment import make_env as build_env  # noqa: E402
from oprl.logging import make_text_logger_func  # noqa: E402
from oprl.parse_args import parse_args  # noqa: E402
from oprl.runners.config import CommonParameters  # noqa: E402
from oprl.runners.train import run_training  # noqa: E402

TRAIN_STEPS = 100_000
REPLAY_TRANSITIONS = 1_000_000


@dataclass
class TrainingScript:
    """Everything a config script exposes: the four factories, the run configuration, ``run()``."""

    algo_cls: type
    algo_name: str
    estimate_q_every: int
    log_every: int

    def __post_init__(self) -> None:
        self.args = parse_args()
        probe = self.make_env(seed=0)
        self.state_dim = int(probe.observation_space.shape[0])
        self.action_dim = int(probe.action_space.shape[0])
        self.config = CommonParameters(state_dim=self.state_dim, action_dim=self.action_dim, num_steps=TRAIN_STEPS,
                                       eval_every=2500, estimate_q_every=self.estimate_q_every,
                                       log_every=self.log_every, device=self.args.device)
        self.make_logger: Callable = make_text_logger_func(algo=self.algo_name, env=self.args.env)

    def make_env(self, seed: int):
        return build_env(self.args.env, seed=seed)

    def make_algo(self, logger):
        return self.algo_cls(logger=logger, state_dim=self.state_dim, action_dim=self.action_dim,
                             device=self.args.device, precision=self.args.precision).create()

    def make_replay_buffer(self):
        return EpisodicReplayBuffer(buffer_size_transitions=max(self.config.num_steps, REPLAY_TRANSITIONS),
                                    state_dim=self.state_dim, action_dim=self.action_dim,
                                    device=self.config.device).create()

    def run(self) -> None:
        run_training(make_algo=self.make_algo, make_env=self.make_env, make_replay_buffer=self.make_replay_buffer,
                     make_logger=self.make_logger, config=self.config, seeds=self.args.seeds,
                     start_seed=self.args.start_seed)
