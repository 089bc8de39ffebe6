"""Synthetic control environment with the observation/action shapes of the
dm_control tasks the benchmarks name.

The reference wraps dm_control / gymnasium / safety-gymnasium
(/root/reference/src/oprl/environment/*.py); those simulators are CPU physics,
outside the learner hot path and absent from the GPU box, so ``make_env`` here
serves a cheap linear-dynamics stand-in with the right dims (SURVEY.md §2.1 #8:
only obs/act dims matter to the metric).  Like dm_control it never terminates;
episodes are truncated at ``episode_length`` steps."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any
import zlib

import numpy as np
import numpy.typing as npt

# task -> (observation dim, action dim): dm_control suite specs
DM_CONTROL_DIMS: dict[str, tuple[int, int]] = {
    "cartpole-balance": (5, 1), "cartpole-swingup": (5, 1), "pendulum-swingup": (3, 1),
    "reacher-easy": (6, 2), "reacher-hard": (6, 2), "finger-spin": (9, 2),
    "hopper-stand": (15, 4), "hopper-hop": (15, 4), "cheetah-run": (17, 6),
    "walker-stand": (24, 6), "walker-walk": (24, 6), "walker-run": (24, 6),
    "quadruped-walk": (78, 12), "humanoid-stand": (67, 21), "humanoid-walk": (67, 21),
    "humanoid-run": (67, 21),
}


@dataclass
class _Box:
    shape: tuple[int, ...]
    low: float = -1.0
    high: float = 1.0


class SyntheticEnv:
    env_family = "synthetic"      # never passes for a simulator: runners and logs can tell

    def __init__(self, name: str, seed: int = 0, episode_length: int = 1000):
        if name not in DM_CONTROL_DIMS:
            raise ValueError(f"unknown env {name!r}; known: {sorted(DM_CONTROL_DIMS)}")
        self.name = name
        self.S, self.A = DM_CONTROL_DIMS[name]
        self.episode_length = episode_length
        self._rng = np.random.RandomState(seed)
        # (crc32, not hash(): str hashes are salted per interpreter, and actors, learner and the
        # children of --seeds N must all build the SAME dynamics for a task name)
        dyn = np.random.RandomState(zlib.crc32(name.encode()) % (2 ** 31))
        self._F = (0.95 * np.eye(self.S) + 0.02 * dyn.standard_normal((self.S, self.S))).astype(np.float32)
        self._G = (0.3 * dyn.standard_normal((self.S, self.A))).astype(np.float32)
        self._goal = dyn.standard_normal(self.S).astype(np.float32)
        self._t = 0
        self._x = np.zeros(self.S, np.float32)

    @property
    def observation_space(self) -> _Box:
        return _Box((self.S,), -np.inf, np.inf)

    @property
    def action_space(self) -> _Box:
        return _Box((self.A,))

    def reset(self) -> tuple[npt.NDArray, dict[str, Any]]:
        self._t = 0
        self._x = self._rng.standard_normal(self.S).astype(np.float32)
        return self._x.copy(), {}

    def sample_action(self) -> npt.NDArray:
        return self._rng.uniform(-1, 1, self.A).astype(np.float32)

    def step(self, action: npt.NDArray):
        a = np.clip(np.asarray(action, np.float32).reshape(self.A), -1, 1)
        self._x = np.tanh(self._F @ self._x + self._G @ a
                          + 0.01 * self._rng.standard_normal(self.S).astype(np.float32))
        self._t += 1
        reward = float(np.exp(-np.mean((self._x - np.tanh(self._goal)) ** 2)))   # in (0, 1]
        truncated = self._t >= self.episode_length
        return self._x.copy(), reward, False, truncated, {}
