"""Learner side of the distributed setup (what the reference's distrib/policy_update_worker.py:22-119
does): per epoch it takes one episode from every actor into the HBM replay, — after the warm-up epochs —
trains ``episode_length x num_env_workers`` updates, and answers every actor with the new policy weights.
The training block is ONE C call (``oprl_learner_step_n``: the update kernels gather their own rows on the
device) instead of thousands of python sample()/update() iterations."""
from __future__ import annotations

import pickle
import time
from pathlib import Path
from typing import Callable

import numpy as np
import torch as t
import torch.nn as nn

from oprl_amd.algos.protocols import AlgorithmProtocol
from oprl_amd.buffers.protocols import ReplayBufferProtocol
from oprl_amd.distrib.queue import Queue, QueueHub
from oprl_amd.environment.protocols import EnvProtocol
from oprl_amd.logging import LoggerProtocol, create_stdout_logger
from oprl_amd.runners.config import DistribConfig

logger = create_stdout_logger()
STOP = b"STOP"
EVAL_EVERY_EPOCHS = 10


class EpochLearner:
    def __init__(self, algo: AlgorithmProtocol, buffer: ReplayBufferProtocol, config: DistribConfig, hub: QueueHub) -> None:
        self.algo, self.buffer, self.config = algo, buffer, config
        n = config.num_env_workers
        self.inboxes = [Queue(f"env_{i}", hub) for i in range(n)]
        self.outboxes = [Queue(f"policy_{i}", hub) for i in range(n)]

    def gather_episodes(self) -> bool:
        """One episode from every actor into the replay; False if the actors stay silent for
        ``learner_num_waits`` seconds in total."""
        deadline = time.monotonic() + float(self.config.learner_num_waits)
        for inbox in self.inboxes:
            data = None
            while not data:
                left = deadline - time.monotonic()
                if left <= 0.0:
                    return False
                data = inbox.pop_wait(min(left, 1.0))
            self.buffer.add_episode(pickle.loads(data))
        return True

    def train(self, i_epoch: int) -> None:
        n_updates = self.config.episode_length * self.config.num_env_workers
        fused = getattr(getattr(self.algo, "learner", None), "step_n", None)
        if fused is not None and hasattr(self.buffer, "handle"):
            fused(self.buffer.handle, n_updates, self.config.batch_size, seed=i_epoch)
            return
        for _ in range(n_updates):
            self.algo.update(*self.buffer.sample(self.config.batch_size))

    def tell_actors(self, message: bytes) -> None:
        for box in self.outboxes:
            box.push(message)

    def policy_message(self) -> bytes:
        weights = {name: w.detach().cpu() for name, w in self.algo.get_policy_state_dict().items()}
        return pickle.dumps(weights)


def run_policy_update_worker(
    make_algo: Callable[[LoggerProtocol], AlgorithmProtocol],
    make_env_test: Callable[[int], EnvProtocol],
    make_buffer: Callable[[], ReplayBufferProtocol],
    make_logger: Callable[[], LoggerProtocol],
    config: DistribConfig,
    hub: QueueHub,
    max_epochs: int | None = None,
    wait_s: float = 0.05,
    on_epoch: Callable[[int, AlgorithmProtocol], None] | None = None,
) -> AlgorithmProtocol:
    algo = make_algo(make_logger())
    learner = EpochLearner(algo, make_buffer(), config, hub)
    i_epoch = 0
    while max_epochs is None or i_epoch < max_epochs:
        if not learner.gather_episodes():
            logger.info("Learner is not receiving data, exiting...")
            break
        if i_epoch > config.warmup_epochs:
            learner.train(i_epoch)
        learner.tell_actors(learner.policy_message())
        if on_epoch is not None:
            on_epoch(i_epoch, algo)
        if i_epoch > 0 and i_epoch % EVAL_EVERY_EPOCHS == 0:
            algo.logger.log_scalar("trainer/ep_reward", evaluate(algo, make_env_test), i_epoch)
            save_policy(algo.actor, algo.logger.log_dir / "weights" / f"epoch_{i_epoch}.w")
        i_epoch += 1
    learner.tell_actors(STOP)
    return algo


def save_policy(policy: nn.Module, save_path: Path) -> None:
    save_path.parent.mkdir(parents=True, exist_ok=True)
    t.save(policy, save_path)


def evaluate(algo: AlgorithmProtocol, make_env_test: Callable[[int], EnvProtocol],
             num_eval_episodes: int = 5, seed: int = 0) -> float:
    """Mean undiscounted return of the greedy policy over a few fresh environments."""
    totals = []
    for k in range(num_eval_episodes):
        env = make_env_test(seed * 100 + k)
        obs, _ = env.reset()
        ret, over = 0.0, False
        while not over:
            obs, reward, terminated, truncated, _ = env.step(algo.actor.exploit(obs))
            ret += reward
            over = bool(terminated or truncated)
        totals.append(ret)
    return float(np.mean(totals))
