"""Recording fakes for the CALLERS of the hot path (fixture G7): a scripted environment,
a counting replay buffer, an algorithm and a logger that only record what is called on them, and an in-memory
queue registry.  oracle/gen_golden.py drives the REFERENCE's BaseTrainer.train / run_policy_update_worker /
run_env_worker with them and stores the traces in tests/golden/callers.npz; tests/test_callers_golden.py
drives this repo's counterparts with the same fakes and compares.  Nothing here computes anything."""
from __future__ import annotations

import queue as _queue

import numpy as np
import torch as t

S, A = 3, 2


class Trace:
    def __init__(self):
        self.events: list[str] = []

    def __call__(self, ev: str) -> None:
        self.events.append(ev)


class FakeEnv:
    """Episodes are truncated after ``length`` steps; global step ``terminate_at`` (counted over the env's
    life) ends its episode with terminated=True instead."""
    env_family = "dm_control"

    def __init__(self, trace: Trace, name: str, length: int, terminate_at: int = -1):
        self.trace, self.name, self.length, self.terminate_at = trace, name, length, terminate_at
        self.t = 0
        self.total = 0

    def reset(self):
        self.trace(f"{self.name}.reset")
        self.t = 0
        return np.full(S, float(self.total), np.float32), {}

    def sample_action(self):
        self.trace(f"{self.name}.sample_action")
        return np.zeros(A, np.float32)

    def step(self, action):
        self.t += 1
        self.total += 1
        terminated = self.total == self.terminate_at
        truncated = self.t >= self.length
        self.trace(f"{self.name}.step")
        return np.full(S, float(self.total), np.float32), 0.5, terminated, truncated, {}


class FakeActor:
    def __init__(self, trace: Trace | None = None):
        self.trace = trace

    def explore(self, state):
        if self.trace:
            self.trace("actor.explore")
        return np.ones(A, np.float32)

    def exploit(self, state):
        if self.trace:
            self.trace("actor.exploit")
        return np.ones(A, np.float32)

    def state_dict(self):
        return {"w": t.zeros(1)}

    def load_state_dict(self, sd):
        if self.trace:
            self.trace("policy.load_state_dict")

    def __getstate__(self):          # picklable by t.save without its trace
        return {}

    def __setstate__(self, st):
        self.trace = None


class FakeAlgo:
    def __init__(self, trace: Trace, logger=None):
        self.trace, self.logger = trace, logger
        self.actor = FakeActor(trace)

    def check_created(self):
        self.trace("algo.check_created")

    def update(self, *batch):
        assert len(batch) == 5
        self.trace("algo.update")

    def get_policy_state_dict(self):
        self.trace("algo.get_policy_state_dict")
        return self.actor.state_dict()


class FakeBuffer:
    episodes_counter = 1
    last_episode_length = 0

    def __init__(self, trace: Trace):
        self.trace = trace
        self.n = 0

    def check_created(self):
        self.trace("buffer.check_created")

    def add_transition(self, state, action, reward, done, episode_done=None):
        self.n += 1
        self.trace(f"buffer.add_transition done={bool(done)} episode_done={bool(episode_done)}")

    def add_episode(self, episode):
        self.n += len(episode)
        self.trace(f"buffer.add_episode len={len(episode)} last_done={bool(episode[-1][3])}")

    def __len__(self):
        return self.n

    def sample(self, batch_size):
        self.trace(f"buffer.sample {batch_size}")
        z = t.zeros
        return z(batch_size, S), z(batch_size, A), z(batch_size, 1), z(batch_size, 1), z(batch_size, S)


class FakeLogger:
    def __init__(self, trace: Trace, log_dir):
        self.trace, self.log_dir = trace, log_dir

    def log_scalar(self, tag, value, step):
        self.trace(f"log {tag} @{step}")

    def log_scalars(self, values, step):
        for tag in values:
            self.trace(f"log {tag} @{step}")


class Registry:
    """Named in-memory FIFOs.  ``as_reference_queue(name)`` has the reference's Queue surface (push / pop ->
    bytes | None); ``queues[name]`` has the multiprocessing.Queue surface this repo's Queue wraps."""

    def __init__(self, trace: Trace, names):
        self.trace = trace
        self.fifo = {n: [] for n in names}
        self.queues = {n: _MpLike(self, n) for n in names}
        self.clock = 0.0                      # advanced by blocking gets that time out

    def reference_queue_class(self):
        reg = self

        class Queue:
            def __init__(self, name, host="localhost"):
                self.name = name

            def push(self, data):
                reg.trace(f"push {self.name}")
                reg.fifo[self.name].append(data)

            def pop(self):
                if reg.fifo[self.name]:
                    reg.trace(f"pop {self.name}")
                    return reg.fifo[self.name].pop(0)
                return None
        return Queue


class _MpLike:
    def __init__(self, reg: Registry, name: str):
        self.reg, self.name = reg, name

    def put(self, data):
        self.reg.trace(f"push {self.name}")
        self.reg.fifo[self.name].append(data)

    def get_nowait(self):
        if self.reg.fifo[self.name]:
            self.reg.trace(f"pop {self.name}")
            return self.reg.fifo[self.name].pop(0)
        raise _queue.Empty

    def get(self, timeout=None):
        try:
            return self.get_nowait()
        except _queue.Empty:
            self.reg.clock += float(timeout or 0.0)
            raise


def compress(events: list[str]) -> list[str]:
    """Run-length form ("algo.update x40") — the traces hold thousands of identical events."""
    out: list[str] = []
    for e in events:
        if out and out[-1].split(" x")[0] == e and (out[-1] == e or " x" in out[-1]):
            head, _, n = out[-1].partition(" x")
            out[-1] = f"{head} x{int(n or 1) + 1}"
        else:
            out.append(e)
    return out
