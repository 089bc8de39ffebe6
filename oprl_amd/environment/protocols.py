"""What the trainers and the actor processes need from an environment (the gymnasium-style step /
reset pair the reference wraps its simulators into, src/oprl/environment/protocols.py:6-34)."""
from __future__ import annotations

from typing import Any, Protocol

import numpy.typing as npt


class EnvProtocol(Protocol):
    env_family: str            # "dm_control" / "gymnasium" / "synthetic" (this build's stand-in): the runners refuse anything else

    def reset(self) -> tuple[npt.NDArray, dict[str, Any]]:
        """Starts an episode: ``(observation [state_dim], info)``."""
        ...

    def step(self, action: npt.NDArray) -> tuple[npt.NDArray, float, bool, bool, dict[str, Any]]:
        """``(next_observation, reward, terminated, truncated, info)``; the trainer stores ``terminated`` as
        the transition's ``done`` and closes the replay episode on either flag."""
        ...

    def sample_action(self) -> npt.NDArray:
        """A uniform action in the action range (the warm-up policy)."""
        ...

    @property
    def observation_space(self):
        """Object with a ``.shape`` (``(state_dim,)``)."""
        ...

    @property
    def action_space(self):
        """Object with a ``.shape`` (``(action_dim,)``)."""
        ...
