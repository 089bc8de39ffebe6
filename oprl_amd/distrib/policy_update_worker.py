"""Learner process of the distributed setup (reference:
/root/reference/src/oprl/distrib/policy_update_worker.py:22-119).

Per epoch: one episode from every actor -> ``buffer.add_episode``; after
``warmup_epochs`` run ``episode_length * num_env_workers`` back-to-back
sample()+update() iterations; push the actor's state_dict to every actor.  The
update loop is one C call (``oprl_learner_step_n``: device-side sampling, no
host sync) instead of 4000 python iterations."""
from __future__ import annotations

import pickle
import time
from itertools import count
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
    scalar_logger = make_logger()
    algo = make_algo(scalar_logger)
    buffer = make_buffer()
    q_envs = [Queue(f"env_{i}", hub) for i in range(config.num_env_workers)]
    q_policies = [Queue(f"policy_{i}", hub) for i in range(config.num_env_workers)]

    for i_epoch in count(0):
        if max_epochs is not None and i_epoch >= max_epochs:
            break
        n_waits = 0
        for i_env in range(config.num_env_workers):
            while True:
                data = q_envs[i_env].pop()
                if data:
                    buffer.add_episode(pickle.loads(data))
                    break
                time.sleep(wait_s)
                n_waits += 1
                if n_waits >= config.learner_num_waits / wait_s:
                    logger.info("Learner is not receiving data, exiting...")
                    for q in q_policies:
                        q.push(b"STOP")
                    return algo

        if i_epoch > config.warmup_epochs:
            n_updates = config.episode_length * config.num_env_workers
            step_n = getattr(getattr(algo, "learner", None), "step_n", None)
            if step_n is not None and hasattr(buffer, "handle"):
                step_n(buffer.handle, n_updates, config.batch_size, seed=i_epoch)
            else:
                for _ in range(n_updates):
                    algo.update(*buffer.sample(config.batch_size))

        payload = pickle.dumps({k: v.detach().cpu() for k, v in algo.get_policy_state_dict().items()})
        for q in q_policies:
            q.push(payload)
        if on_epoch is not None:
            on_epoch(i_epoch, algo)
        if i_epoch > 0 and i_epoch % 10 == 0:
            mean_reward = evaluate(algo, make_env_test)
            algo.logger.log_scalar("trainer/ep_reward", mean_reward, i_epoch)
            save_policy(algo.actor, algo.logger.log_dir / "weights" / f"epoch_{i_epoch}.w")
    for q in q_policies:
        q.push(b"STOP")
    return algo


def save_policy(policy: nn.Module, save_path: Path) -> None:
    save_path.parent.mkdir(parents=True, exist_ok=True)
    t.save(policy, save_path)


def evaluate(algo: AlgorithmProtocol, make_env_test: Callable[[int], EnvProtocol],
             num_eval_episodes: int = 5, seed: int = 0) -> float:
    returns = []
    for i_ep in range(num_eval_episodes):
        env_test = make_env_test(seed * 100 + i_ep)
        state, _ = env_test.reset()
        total, done = 0.0, False
        while not done:
            state, reward, terminated, truncated, _ = env_test.step(algo.actor.exploit(state))
            total += reward
            done = terminated or truncated
        returns.append(total)
    return float(np.mean(returns))
