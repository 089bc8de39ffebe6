"""Replay-buffer interface — the data side of the drop-in boundary.

Member names and call shapes are the reference's (src/oprl/buffers/protocols.py:6-26); the docstrings
state what the HBM-resident implementation (buffers/episodic_buffer.py + csrc/replay.hip) guarantees."""
from __future__ import annotations

from typing import Protocol, runtime_checkable

import torch as t


@runtime_checkable
class ReplayBufferProtocol(Protocol):
    episodes_counter: int      # episode slots in use (the ring evicts whole episodes)
    _created: bool

    def create(self) -> "ReplayBufferProtocol":
        """Allocates the four storage tensors on the device (``states[E, L+1, S]``, ``actions[E, L, A]``,
        ``rewards[E, L, 1]``, ``dones[E, L, 1]``) and the native handle; returns ``self``."""
        ...

    def check_created(self) -> None:
        """Raises ``RuntimeError`` before ``create()``."""
        ...

    def add_transition(self, state, action, reward, done, episode_done=None):
        """Appends one step to the episode being written (host rows staged in pinned memory, flushed to
        HBM in batches); ``episode_done`` closes the episode slot.  Raises ``IndexError`` when an episode
        outgrows ``max_episode_lenth``."""
        ...

    def add_episode(self, episode):
        """A whole episode as rows ``[state, action, reward, done, next_state]`` (what an actor process
        sends), then closes the slot."""
        ...

    def sample(self, batch_size) -> tuple[t.Tensor, t.Tensor, t.Tensor, t.Tensor, t.Tensor]:
        """Uniform with replacement over the live transitions, including the tail of the episode still
        being written: ``(state, action, reward, done, next_state)`` as fresh float32 device tensors,
        gathered by one kernel.  Raises on an empty buffer, like ``np.random.randint(0, 0)``."""
        ...

    def __len__(self) -> int:
        """Live transitions."""
        ...

    @property
    def last_episode_length(self) -> int:
        """Steps written so far into the current episode slot."""
        ...
