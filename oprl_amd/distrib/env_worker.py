"""Actor side of the distributed setup (what the reference's distrib/env_worker.py:15-64 does over
RabbitMQ): a CPU process that owns one environment and one policy snapshot, sends the learner one whole
episode at a time and then blocks until the learner answers with fresh weights (or tells it to stop).
The policy is a plain CPU module — single-observation ``explore`` calls, no GPU in the actors."""
from __future__ import annotations

import pickle
from typing import Callable

from oprl_amd.algos.protocols import PolicyProtocol
from oprl_amd.distrib.queue import Queue, QueueHub
from oprl_amd.environment.protocols import EnvProtocol
from oprl_amd.logging import create_stdout_logger
from oprl_amd.runners.config import DistribConfig

logger = create_stdout_logger()
STOP = b"STOP"


class EpisodeActor:
    """Environment + policy snapshot; ``rollout()`` returns one episode as the list of
    ``[state, action, reward, terminated, next_state]`` rows the replay's ``add_episode`` takes."""

    def __init__(self, env: EnvProtocol, policy: PolicyProtocol, episode_length: int, warmup_env_steps: int) -> None:
        self.env, self.policy = env, policy
        self.episode_length = int(episode_length)
        self.warmup_env_steps = int(warmup_env_steps)
        self.env_steps = 0               # counted over the worker's lifetime: uniform actions until the warm-up is over

    def act(self, state):
        if self.env_steps <= self.warmup_env_steps:
            return self.env.sample_action()
        return self.policy.explore(state)

    def rollout(self) -> list:
        rows = []
        state, _ = self.env.reset()
        for _ in range(self.episode_length):
            action = self.act(state)
            nxt, reward, terminated, truncated, _ = self.env.step(action)
            rows.append([state, action, reward, terminated, nxt])
            if terminated or truncated:
                break                    # (the step that ends an episode is not counted, as in the reference)
            state = nxt
            self.env_steps += 1
        return rows


def run_env_worker(
    make_env: Callable[[int], EnvProtocol],
    make_policy: Callable[[], PolicyProtocol],
    config: DistribConfig,
    id_worker: int,
    hub: QueueHub,
    policy_wait_s: float = 0.05,
) -> None:
    actor = EpisodeActor(make_env(seed=id_worker), make_policy(), config.episode_length, config.warmup_env_steps)
    to_learner = Queue(f"env_{id_worker}", hub)
    from_learner = Queue(f"policy_{id_worker}", hub)
    for _episode in range(config.episodes_per_worker):
        to_learner.push(pickle.dumps(actor.rollout()))
        # lock step with the learner: nothing happens here until it has trained on this episode
        reply = None
        while reply is None:
            reply = from_learner.pop_wait(max(policy_wait_s, 1.0))
        if reply == STOP:
            return
        actor.policy.load_state_dict(pickle.loads(reply))
    logger.info(f"env worker {id_worker} done")
