"""Actor process: roll out one episode, ship it, wait for the new policy
(reference: /root/reference/src/oprl/distrib/env_worker.py:15-64).  Actors are
CPU processes; the policy they hold is a plain CPU module (B=1 explore)."""
from __future__ import annotations

import pickle
import time
from typing import Callable

from oprl_amd.algos.protocols import PolicyProtocol
from oprl_amd.distrib.queue import Queue, QueueHub
from oprl_amd.environment.protocols import EnvProtocol
from oprl_amd.logging import create_stdout_logger
from oprl_amd.runners.config import DistribConfig

logger = create_stdout_logger()


def run_env_worker(
    make_env: Callable[[int], EnvProtocol],
    make_policy: Callable[[], PolicyProtocol],
    config: DistribConfig,
    id_worker: int,
    hub: QueueHub,
    policy_wait_s: float = 0.05,
) -> None:
    env = make_env(seed=id_worker)
    policy = make_policy()
    q_env, q_policy = Queue(f"env_{id_worker}", hub), Queue(f"policy_{id_worker}", hub)
    total_env_step = 0
    for i_ep in range(config.episodes_per_worker):
        episode = []
        state, _ = env.reset()
        for _ in range(config.episode_length):
            if total_env_step <= config.warmup_env_steps:
                action = env.sample_action()
            else:
                action = policy.explore(state)
            next_state, reward, terminated, truncated, _ = env.step(action)
            episode.append([state, action, reward, terminated, next_state])
            if terminated or truncated:
                break
            state = next_state
            total_env_step += 1
        q_env.push(pickle.dumps(episode))
        while True:                       # lock-step with the learner, as in the reference
            data = q_policy.pop()
            if data is None:
                time.sleep(policy_wait_s)
                continue
            if data == b"STOP":
                return
            policy.load_state_dict(pickle.loads(data))
            break
    logger.info(f"env worker {id_worker} done")
