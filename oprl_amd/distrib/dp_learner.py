"""Config 5 of BASELINE.json end to end: N data-parallel learner processes (one per MI355X, gradients
all-reduced over RCCL / xGMI) fed by CPU actor processes through shared-memory rings.

What the reference does with one learner and a broker (distrib/policy_update_worker.py:45-76,
distrib/env_worker.py:39-62), re-partitioned as SURVEY.md section 8e / 8f N2 ask:

* actor ``i`` writes its transitions into ring ``i``; ring ``i`` belongs to learner rank ``i % world``, so
  every rank fills its OWN replay shard in its own HBM (32 actors on 8 ranks: 4 each);
* the learners drain their rings between chunks of updates and run the synchronous data-parallel update
  (``DataParallelLearner``: two gradient all-reduces per update) — ``oprl_learner_dp_step_n`` in C when the
  engine is the HIP learner, ``dp.update(*buffer.sample())`` otherwise (CPU tests with gloo);
* rank 0 publishes the policy on the ``PolicyBoard`` after every chunk; actors pick it up at their next
  episode boundary.  Nobody takes turns: actors keep stepping while the learners train.

Every decision that changes what the ranks do next (start training, how many updates, stop) is taken
from all-reduced numbers, so the ranks stay in lock step without a coordinator."""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Callable

import numpy as np
import torch as t
import torch.distributed as dist

from oprl_amd.distrib.shm import PolicyBoard, TransitionRing, flatten_state_dict, unflatten_into
from oprl_amd.parallel import DataParallelLearner


@dataclass
class LearnerPlan:
    """What the learner ranks do (picklable; shared by all ranks)."""
    total_updates: int                   # optimiser steps of the whole job
    batch_size: int = 128                # per rank (global batch = world x batch_size)
    chunk: int = 200                     # updates between two ring drains / policy publications
    warmup_transitions: int = 1000       # per rank, before the first update
    updates_per_transition: float = 1.0  # optimiser steps per environment transition received (all ranks together)
    seed: int = 0
    idle_timeout_s: float = 60.0         # nothing arrives for this long: before training could start an error, after it the end of the run
    wall_timeout_s: float | None = None  # bound on the whole run (run_dp_training terminates the ranks beyond it)
    force_exchange: bool = False         # one rank: still the data-parallel step (phase / RCCL all-reduce / apply), not the fused one


def run_ring_actor(make_env: Callable, make_policy: Callable, config, id_worker: int, ring_name: str,
                   board_name: str) -> None:
    """One CPU actor: steps its environment with the newest policy on the board and appends every
    transition to its ring.  Uniform actions for the first ``warmup_env_steps`` steps, as the reference's
    actor (env_worker.py:39-43)."""
    t.set_num_threads(1)      # dozens of actors share the host: one core each (a 256-wide MLP on one observation)
    env, policy = make_env(seed=id_worker), make_policy()
    ring, board = TransitionRing(ring_name), PolicyBoard(board_name)
    have, env_steps = 0, 0
    try:
        for _episode in range(config.episodes_per_worker):
            if board.stopped:
                break
            newer = board.read_if_newer(have)
            if newer is not None:
                have = newer[0]
                unflatten_into(policy, newer[1])
            state, _ = env.reset()
            for k in range(config.episode_length):
                action = env.sample_action() if env_steps <= config.warmup_env_steps else policy.explore(state)
                nxt, reward, terminated, truncated, _ = env.step(action)
                over = bool(terminated or truncated) or k == config.episode_length - 1
                while not ring.push(state, action, reward, bool(terminated), over, timeout_s=0.5):
                    if board.stopped:
                        return
                if over:
                    break
                state = nxt
                env_steps += 1
    finally:
        ring.close_writer()
        ring.detach()
        board.detach()


class RingDrainer:
    """This rank's rings -> its replay shard.  The replay is EPISODIC (one episode being written at a time,
    s' = the next row), while several actors write concurrently: records are held per ring until their
    episode is complete and then stored as one contiguous episode, the way the reference's learner takes
    whole episodes from its actors (policy_update_worker.py:49-54)."""

    def __init__(self, rings: list[TransitionRing], buffer):
        self.rings, self.buffer = rings, buffer
        self.pending: list[np.ndarray | None] = [None for _ in rings]

    def __call__(self) -> int:
        """Drain everything waiting; returns the number of transitions stored into the replay."""
        stored = 0
        block = getattr(self.buffer, "add_transitions", None)
        for i, ring in enumerate(self.rings):
            rows = ring.pop_all()
            if len(rows) == 0:
                continue
            if self.pending[i] is not None:
                rows = np.concatenate([self.pending[i], rows])
            S, A = ring.S, ring.A
            ends = np.flatnonzero(rows[:, S + A + 2])           # last rows of complete episodes
            a = 0
            for b in ends:
                ep = rows[a:b + 1]
                if block is not None:                           # one library call per episode
                    block(ep, episode_done=True)
                else:
                    for r in ep:
                        self.buffer.add_transition(r[:S], r[S:S + A], float(r[S + A]), bool(r[S + A + 1]),
                                                   episode_done=bool(r[S + A + 2]))
                stored += len(ep)
                a = b + 1
            self.pending[i] = rows[a:].copy() if a < len(rows) else None
        return stored

    def exhausted(self) -> bool:
        return all(r.closed and len(r) == 0 for r in self.rings)


def learner_rank_loop(rank: int, world: int, algo, buffer, rings: list[TransitionRing], board: PolicyBoard,
                      plan: LearnerPlan, group=None, engine=None, native: bool | None = None,
                      on_chunk: Callable | None = None) -> dict:
    """The body of one learner rank (torch.distributed already initialised).  Returns its statistics."""
    eng = engine if engine is not None else algo.learner
    if native is None:
        native = hasattr(buffer, "handle") and t.cuda.is_available() and hasattr(eng, "handle")
    # one rank: no exchange — the plain fused update (4 launches per update instead of the data-parallel
    # path's phase / all-reduce / apply sequence), if the engine was created without export_grads
    solo = world == 1 and native and not getattr(eng, "export_grads", False)
    dp = None
    if not solo:
        dp = DataParallelLearner(algo, group, engine=eng)
        if native:
            dp.init_native_comm()
        dp.broadcast_parameters(src=0)
    if rank == 0:
        board.publish(flatten_state_dict(algo.get_policy_state_dict()))
    dev = getattr(eng, "device", None) or t.device("cpu")
    received, done, chunks = 0, 0, 0
    t_last_data, t_train = time.monotonic(), 0.0
    drain = RingDrainer(rings, buffer)
    t_begin = time.monotonic()
    t_parts = [0.0, 0.0, 0.0]          # enqueue the chunk / drain the rings / publish (= wait for the GPU)
    stopped_early = None
    sample_seed = getattr(buffer, "seed", None)      # the buffer's own (per-rank) sampler seed
    if sample_seed is None:
        sample_seed = plan.seed
    while done < plan.total_updates:
        ready = received >= plan.warmup_transitions and len(buffer) >= plan.batch_size
        dry = drain.exhausted() or time.monotonic() - t_last_data > plan.idle_timeout_s   # nothing more is coming
        starved = (not ready) and dry
        # one small all-reduce decides for everybody: [all ready?, anybody starved?, every rank dry?, transitions received]
        k = 0
        if world > 1:
            v = t.tensor([1.0 if ready else 0.0, 0.0 if starved else 1.0, 1.0 if dry else 0.0, float(received)],
                         dtype=t.float64, device=dev)
            lo = v.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
            tot = v.clone()
            dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
            all_ready, any_starved, all_dry, total_received = (lo[0].item() == 1.0, lo[1].item() == 0.0,
                                                               lo[2].item() == 1.0, tot[3].item())
        else:
            all_ready, any_starved, all_dry, total_received = ready, starved, dry, float(received)
        if any_starved:
            raise RuntimeError("a learner rank ran out of actor data before training could start")
        if all_ready:
            k = min(plan.chunk, plan.total_updates - done, int(plan.updates_per_transition * total_received) - done)
            if k <= 0 and all_dry:
                # the update quota of everything the actors ever delivered is used up and no ring will deliver more
                # (actors finished, died, or their episodes ended early): the reference's learner gives up after
                # learner_num_waits empty polls (distrib/policy_update_worker.py:55-63); this one returns what it has
                # done instead of spinning on empty rings for ever
                stopped_early = (f"actors delivered {int(total_received)} transitions = a quota of "
                                 f"{int(plan.updates_per_transition * total_received)} updates of {plan.total_updates}")
                break
        t0 = time.monotonic()
        if k > 0:      # enqueue the chunk (asynchronous on the GPU path) ...
            if solo:
                eng.step_n(buffer.handle, k, plan.batch_size, seed=sample_seed)
            elif native:
                dp.step_n(buffer.handle, k, plan.batch_size, seed=sample_seed)
            else:
                for _ in range(k):
                    dp.update(*buffer.sample(plan.batch_size))
        # ... and take in what the actors produced meanwhile: the ring drain (host work + staged H2D copies,
        # ordered behind the chunk on the stream) overlaps the GPU's updates instead of leaving it idle
        t1 = time.monotonic()
        got = drain()
        t2 = time.monotonic()
        received += got
        if got:
            t_last_data = time.monotonic()
        if k > 0:
            done += k
            chunks += 1
            if rank == 0:
                board.publish(flatten_state_dict(algo.get_policy_state_dict()))     # (a D2H copy: it also ends the chunk)
            elif dev.type == "cuda":
                t.cuda.synchronize(dev)
            t3 = time.monotonic()
            t_train += t3 - t0
            t_parts[0] += t1 - t0
            t_parts[1] += t2 - t1
            t_parts[2] += t3 - t2
            if on_chunk is not None:
                on_chunk(rank, done, dp)
        elif got == 0:
            time.sleep(0.002)
    spread = float(dp.replica_checksum().abs().max()) if dp is not None else 0.0
    if rank == 0:
        board.stop()
    return dict(rank=rank, updates=done, received=received, chunks=chunks, train_s=t_train,
                wall_s=time.monotonic() - t_begin, enqueue_s=t_parts[0], drain_s=t_parts[1], publish_wait_s=t_parts[2],
                replica_spread=spread, policy_version=board.version, stopped_early=stopped_early)
