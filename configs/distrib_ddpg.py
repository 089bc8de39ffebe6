"""Distributed DDPG in the reference's config-script shape
(/root/reference/configs/distrib_ddpg.py): CPU actor processes feed the GPU learner's
HBM replay; here through in-host queues (``oprl_amd/distrib/queue.py``) instead of
RabbitMQ.  The reference's own script runs against this repo unchanged.

    python configs/distrib_ddpg.py --env walker-walk --device cuda
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch.nn as nn  # noqa: E402

from oprl.algos.ddpg import DDPG  # noqa: E402
from oprl.algos.nn_models import DeterministicPolicy  # noqa: E402
from oprl.algos.protocols import AlgorithmProtocol, PolicyProtocol  # noqa: E402
from oprl.buffers.episodic_buffer import EpisodicReplayBuffer  # noqa: E402
from oprl.buffers.protocols import ReplayBufferProtocol  # noqa: E402
from oprl.distrib.env_worker import run_env_worker  # noqa: E402
from oprl.distrib.policy_update_worker import run_policy_update_worker  # noqa: E402
from oprl.environment import make_env as _make_env  # noqa: E402
from oprl.logging import FileTxtLogger, LoggerProtocol, get_logs_path  # noqa: E402
from oprl.parse_args import parse_args_distrib  # noqa: E402
from oprl.runners.config import DistribConfig  # noqa: E402
from oprl.runners.train_distrib import run_distrib_training  # noqa: E402

config = DistribConfig(batch_size=128, num_env_workers=4, episodes_per_worker=100, warmup_epochs=16,
                       episode_length=1000, learner_num_waits=10)
args = parse_args_distrib()


def make_env(seed: int):
    return _make_env(args.env, seed=seed)


_probe = make_env(seed=0)
STATE_DIM: int = _probe.observation_space.shape[0]
ACTION_DIM: int = _probe.action_space.shape[0]


def make_logger() -> LoggerProtocol:
    log_dir = get_logs_path(logdir=os.environ.get("OPRL_LOGS", "logs"), algo="DistribDDPG", env=args.env, seed=0)
    return FileTxtLogger(log_dir)


def make_policy() -> PolicyProtocol:
    # the actors run on the CPU (one process each); only the learner owns the GPU
    return DeterministicPolicy(state_dim=STATE_DIM, action_dim=ACTION_DIM, hidden_units=(256, 256),
                               hidden_activation=nn.ReLU(inplace=True), device="cpu")


def make_replay_buffer() -> ReplayBufferProtocol:
    return EpisodicReplayBuffer(buffer_size_transitions=int(1_000_000), state_dim=STATE_DIM,
                                action_dim=ACTION_DIM, device=args.device).create()


def make_algo(logger: LoggerProtocol) -> AlgorithmProtocol:
    return DDPG(logger=logger, state_dim=STATE_DIM, action_dim=ACTION_DIM, device=args.device).create()


if __name__ == "__main__":
    run_distrib_training(run_env_worker=run_env_worker, run_policy_update_worker=run_policy_update_worker,
                         make_env=make_env, make_algo=make_algo, make_policy=make_policy,
                         make_replay_buffer=make_replay_buffer, make_logger=make_logger, config=config)
