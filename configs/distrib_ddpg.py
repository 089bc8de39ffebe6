"""Distributed DDPG, the counterpart of the reference's configs/distrib_ddpg.py: a pool of CPU actor
processes steps environments with a policy snapshot and feeds the learner(s), which own the GPU(s), keep
the replay in HBM and train with the fused HIP update.

    python configs/distrib_ddpg.py --env walker-walk --device cuda                 # one learner, in-host queues
    python configs/distrib_ddpg.py --env walker-walk --learners 8 --actors 32      # BASELINE.json config 5

With ``--learners 1`` the layout is the reference's: whole episodes to ONE learner, actors and learner taking
turns, over in-host queues (``oprl_amd/distrib/queue.py``; the reference used RabbitMQ).  With ``--learners N``
the run is N data-parallel learner processes, one per MI355X, gradients all-reduced over RCCL / xGMI, actor i
writing into the HBM replay shard of rank i % N through a shared-memory ring, the policy published by rank 0 on
a shared-memory board — actors and learners overlap (``oprl_amd/distrib/dp_learner.py``).  The reference's own
script also runs against this repo unchanged (same module paths, same keywords).
"""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import torch.nn as nn  # noqa: E402
from oprl.algos.ddpg import DDPG  # noqa: E402
from oprl.algos.nn_models import DeterministicPolicy  # noqa: E402
from oprl.buffers.episodic_buffer import EpisodicReplayBuffer  # noqa: E402
from oprl.distrib import env_worker, policy_update_worker  # noqa: E402
from oprl.environment import make_env as build_env  # noqa: E402
from oprl.logging import FileTxtLogger, get_logs_path  # noqa: E402
from oprl.parse_args import parse_args_distrib  # noqa: E402
from oprl.runners.config import DistribConfig  # noqa: E402
from oprl.runners.train_distrib import run_distrib_training  # noqa: E402

cli = parse_args_distrib()
HIDDEN = (256, 256)
REPLAY_TRANSITIONS = 1_000_000

# 32 actors (BASELINE.json config 5; the reference's script starts 4) x 100 episodes of 1000 steps; the
# learner(s) start after 16 warm-up epochs
settings = DistribConfig(num_env_workers=cli.actors or 32, episodes_per_worker=100, episode_length=1000,
                         warmup_epochs=16, batch_size=128, learner_num_waits=10)


def make_env(seed: int):
    return build_env(cli.env, seed=seed)


_dims_probe = make_env(seed=0)
OBS_DIM = int(_dims_probe.observation_space.shape[0])
ACT_DIM = int(_dims_probe.action_space.shape[0])
del _dims_probe


def make_policy():
    """What an actor process holds: a CPU copy of the deterministic policy (weights arrive from the learner)."""
    return DeterministicPolicy(OBS_DIM, ACT_DIM, hidden_units=HIDDEN, hidden_activation=nn.ReLU(inplace=True),
                               device="cpu")


def make_algo(logger, **overrides):
    """``overrides``: what a data-parallel rank needs on top (device=cuda:<rank>, export_grads=True)."""
    return DDPG(logger=logger, state_dim=OBS_DIM, action_dim=ACT_DIM,
                **{"device": cli.device, "precision": cli.precision, **overrides}).create()


def make_replay_buffer(**overrides):
    """``overrides``: a data-parallel rank's device and sampler seed (its own HBM shard)."""
    return EpisodicReplayBuffer(buffer_size_transitions=REPLAY_TRANSITIONS, state_dim=OBS_DIM, action_dim=ACT_DIM,
                                **{"device": cli.device, **overrides}).create()


def make_logger():
    where = get_logs_path(logdir=os.environ.get("OPRL_LOGS", "logs"), algo="DistribDDPG", env=cli.env, seed=0)
    return FileTxtLogger(where)


if __name__ == "__main__":
    run_distrib_training(config=settings, make_env=make_env, make_policy=make_policy, make_algo=make_algo,
                         make_replay_buffer=make_replay_buffer, make_logger=make_logger,
                         run_env_worker=env_worker.run_env_worker,
                         run_policy_update_worker=policy_update_worker.run_policy_update_worker,
                         learners=cli.learners)
