"""In-host message queues for the actor <-> learner feed.

The reference talks pickles over RabbitMQ via pika (one broker queue per actor,
/root/reference/src/oprl/distrib/queue.py:5-19).  On a single MI355X node the
actors are CPU processes on the learner's host, so the broker is replaced by
``multiprocessing`` queues created by the runner and handed to the workers
(no broker process, no network hop).  Same ``push`` / non-blocking ``pop``
surface, addressed by the same names (``env_{i}``, ``policy_{i}``)."""
from __future__ import annotations

import queue as _queue
from multiprocessing import get_context


class QueueHub:
    """All named queues of one distributed run (picklable: pass it to workers)."""

    def __init__(self, names, ctx=None):
        ctx = ctx or get_context("spawn")
        self.queues = {n: ctx.Queue() for n in names}


class Queue:
    def __init__(self, name: str, hub: QueueHub) -> None:
        self._name = name
        self._q = hub.queues[name]

    def push(self, data) -> None:
        self._q.put(data)

    def pop(self) -> bytes | None:
        try:
            return self._q.get_nowait()
        except _queue.Empty:
            return None

    def pop_wait(self, timeout_s: float) -> bytes | None:
        """Blocking pop: the next message, or None after ``timeout_s`` seconds without one (the reference
        polls its broker and sleeps; a process queue can simply block)."""
        try:
            return self._q.get(timeout=max(timeout_s, 0.0))
        except _queue.Empty:
            return None
