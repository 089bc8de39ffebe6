"""Shared-memory transport between the CPU actor processes and the learner rank(s) of one node (SURVEY.md
section 8f, row N2).

The reference ships every episode as a pickle through a RabbitMQ broker and every policy as a pickled
state_dict back (distrib/env_worker.py:39-62, policy_update_worker.py:45-76), actors and learner taking
turns.  Here the actors are processes on the learner's host, so the hand-off is memory:

* ``TransitionRing`` — one single-producer / single-consumer ring per actor of fixed-size float32 records
  ``[state (S) | action (A) | reward | terminated | episode_done]``; the actor appends while it steps its
  environment, the learner drains whole rings between chunks of updates (no pickle, no copy through a
  broker, no lock: the producer only writes ``head``, the consumer only writes ``tail``; a full ring makes
  the actor wait, which is the back-pressure).
* ``PolicyBoard`` — the current policy as one flat float32 vector with a sequence counter (seqlock: odd
  while the learner writes); actors pick up a newer version at their next episode boundary — nobody waits
  for anybody.

Both are plain ``multiprocessing.shared_memory`` blocks addressed by name, so they cross ``spawn``."""
from __future__ import annotations

import time
from multiprocessing import shared_memory

import numpy as np

_HDR = 256          # bytes; head, tail, geometry and flags on separate 64-byte lines


class TransitionRing:
    def __init__(self, name: str | None, capacity: int = 0, state_dim: int = 0, action_dim: int = 0,
                 create: bool = False):
        if create:
            rec = state_dim + action_dim + 3
            self.shm = shared_memory.SharedMemory(create=True, size=_HDR + 4 * rec * capacity, name=name)
            hdr = np.ndarray((_HDR // 8,), dtype=np.uint64, buffer=self.shm.buf)
            hdr[:] = 0
            hdr[16], hdr[17], hdr[18], hdr[19] = capacity, rec, state_dim, action_dim
        else:
            self.shm = shared_memory.SharedMemory(name=name)
        self._hdr = np.ndarray((_HDR // 8,), dtype=np.uint64, buffer=self.shm.buf)
        self.capacity, self.rec = int(self._hdr[16]), int(self._hdr[17])
        self.S, self.A = int(self._hdr[18]), int(self._hdr[19])
        self._data = np.ndarray((self.capacity, self.rec), dtype=np.float32, buffer=self.shm.buf, offset=_HDR)
        self.name = self.shm.name
        self._owner = create

    # head = records ever written (hdr[0]), tail = records ever consumed (hdr[8]), closed = hdr[24]
    def __len__(self) -> int:
        return int(self._hdr[0] - self._hdr[8])

    @property
    def closed(self) -> bool:
        return bool(self._hdr[24])

    def close_writer(self) -> None:
        """The producer is done (its last record is already visible)."""
        self._hdr[24] = 1

    def push(self, state, action, reward: float, terminated: bool, episode_done: bool,
             timeout_s: float | None = None) -> bool:
        """Producer side.  Waits while the ring is full (back-pressure); False on timeout."""
        t0, nap = time.monotonic(), 0.0002
        while int(self._hdr[0] - self._hdr[8]) >= self.capacity:
            if timeout_s is not None and time.monotonic() - t0 > timeout_s:
                return False
            time.sleep(nap)                    # a full ring means the learner is the slower side: back off
            nap = min(2 * nap, 0.005)          # (32 polling actors must not eat the learner's host cores)
        row = self._data[int(self._hdr[0]) % self.capacity]
        row[:self.S] = state
        row[self.S:self.S + self.A] = action
        row[self.S + self.A] = reward
        row[self.S + self.A + 1] = 1.0 if terminated else 0.0
        row[self.S + self.A + 2] = 1.0 if episode_done else 0.0
        self._hdr[0] += 1                      # publish (x86 stores are not reordered with earlier stores)
        return True

    def pop_all(self, max_records: int | None = None) -> np.ndarray:
        """Consumer side: every record written so far (oldest first), as a copy ``[n, rec]``."""
        head, tail = int(self._hdr[0]), int(self._hdr[8])
        n = head - tail
        if max_records is not None:
            n = min(n, max_records)
        if n <= 0:
            return np.empty((0, self.rec), np.float32)
        i0 = tail % self.capacity
        first = min(n, self.capacity - i0)
        out = np.concatenate([self._data[i0:i0 + first], self._data[:n - first]]) if first < n \
            else self._data[i0:i0 + n].copy()
        self._hdr[8] = tail + n                # release the slots
        return out

    def detach(self) -> None:
        self._hdr = self._data = None
        self.shm.close()
        if self._owner:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass


class PolicyBoard:
    """The newest policy parameters (flat float32, ``state_dict()`` order) + a stop flag."""

    def __init__(self, name: str | None, n_floats: int = 0, create: bool = False):
        if create:
            self.shm = shared_memory.SharedMemory(create=True, size=_HDR + 4 * n_floats, name=name)
            hdr = np.ndarray((_HDR // 8,), dtype=np.uint64, buffer=self.shm.buf)
            hdr[:] = 0
            hdr[16] = n_floats
        else:
            self.shm = shared_memory.SharedMemory(name=name)
        self._hdr = np.ndarray((_HDR // 8,), dtype=np.uint64, buffer=self.shm.buf)
        self.n = int(self._hdr[16])
        self._data = np.ndarray((self.n,), dtype=np.float32, buffer=self.shm.buf, offset=_HDR)
        self.name = self.shm.name
        self._owner = create

    @property
    def version(self) -> int:
        return int(self._hdr[0]) // 2

    @property
    def stopped(self) -> bool:
        return bool(self._hdr[8])

    def stop(self) -> None:
        self._hdr[8] = 1

    def publish(self, flat: np.ndarray) -> int:
        """Learner (rank 0) side: one writer.  Returns the new version."""
        assert flat.size == self.n, (flat.size, self.n)
        self._hdr[0] += 1                      # odd: write in progress
        self._data[:] = flat.reshape(-1)
        self._hdr[0] += 1
        return self.version

    def read_if_newer(self, have: int) -> tuple[int, np.ndarray] | None:
        """Actor side: (version, copy of the parameters) if a version newer than ``have`` is complete."""
        for _ in range(100):
            s0 = int(self._hdr[0])
            if s0 % 2 == 1:
                time.sleep(0.0001)
                continue
            if s0 // 2 <= have:
                return None
            flat = self._data.copy()
            if int(self._hdr[0]) == s0:
                return s0 // 2, flat
        return None

    def detach(self) -> None:
        self._hdr = self._data = None
        self.shm.close()
        if self._owner:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass


def flatten_state_dict(sd) -> np.ndarray:
    """One flat float32 vector, state_dict order.  Device tensors are concatenated ON the device and come over in one
    copy: a ``torch.cat`` of CPU tensors wakes torch's intra-op thread pool (one thread per host core), whose workers
    then spin for milliseconds — beside a learner's launch loop that halved the update rate (measured 29 -> 75 us per
    update in 500-update chunks, tools/probe_gap.py: the spinning workers exhaust the container's CPU quota)."""
    import torch as t
    vals = [v.detach().reshape(-1).to(dtype=t.float32) for v in sd.values()]
    if vals and all(v.is_cuda for v in vals):
        return t.cat(vals).cpu().numpy()
    out = np.empty(sum(v.numel() for v in vals), dtype=np.float32)
    off = 0
    for v in vals:
        n = v.numel()
        out[off:off + n] = v.cpu().numpy()
        off += n
    return out


def unflatten_into(policy, flat: np.ndarray) -> None:
    """Load a flat vector (``flatten_state_dict`` order) into ``policy`` through load_state_dict."""
    import torch as t
    sd, off = {}, 0
    for k, v in policy.state_dict().items():
        n = v.numel()
        sd[k] = t.from_numpy(flat[off:off + n].copy()).view(v.shape)
        off += n
    assert off == flat.size, (off, flat.size)
    policy.load_state_dict(sd)
