"""Environment interface used by the trainers (reference:
/root/reference/src/oprl/environment/protocols.py:6-34)."""
from __future__ import annotations

from typing import Any, Protocol

import numpy.typing as npt


class EnvProtocol(Protocol):
    env_family: str

    def step(self, action: npt.NDArray) -> tuple[npt.NDArray, float, bool, bool, dict[str, Any]]: ...

    def reset(self) -> tuple[npt.NDArray, dict[str, Any]]: ...

    def sample_action(self) -> npt.NDArray: ...

    @property
    def observation_space(self): ...

    @property
    def action_space(self): ...
