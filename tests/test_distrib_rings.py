"""The shared-memory transport and the wired data-parallel learner path of BASELINE.json config 5, on CPU:
ring wrap-around / ordering / back-pressure with a producer PROCESS, the policy board's seqlock, and two
learner ranks under gloo (engine = the CPU oracle behind the export_grads interface) fed by ring actors —
per-rank shards, collective start, replicas bit-identical, the policy reaching the actors."""
import json
import os
import tempfile
import time
from multiprocessing import get_context

import numpy as np
import torch as t

from oprl_amd.distrib.shm import PolicyBoard, TransitionRing, flatten_state_dict, unflatten_into

S, A = 5, 2


def _producer(name, n, delay):
    ring = TransitionRing(name)
    for k in range(n):
        assert ring.push(np.full(S, k, np.float32), np.full(A, -k, np.float32), float(k), k % 7 == 0, k % 5 == 4)
        if delay and k % 50 == 0:
            time.sleep(delay)
    ring.close_writer()
    ring.detach()


def test_ring_keeps_order_across_wrap_around_and_applies_back_pressure():
    ring = TransitionRing(None, capacity=64, state_dim=S, action_dim=A, create=True)
    try:
        n = 1000                                   # 15 times around a 64-slot ring
        p = get_context("spawn").Process(target=_producer, args=(ring.name, n, 0.001))
        p.start()
        got = []
        t0 = time.monotonic()
        while len(got) < n and time.monotonic() - t0 < 60:
            assert len(ring) <= ring.capacity       # the producer never overruns the consumer
            rows = ring.pop_all(max_records=17)     # ragged drains
            got.extend(rows)
            if len(rows) == 0:
                time.sleep(0.0005)
        p.join(timeout=10)
        assert p.exitcode == 0 and len(got) == n and ring.closed and len(ring) == 0
        got = np.stack(got)
        k = np.arange(n, dtype=np.float32)
        assert np.array_equal(got[:, 0], k) and np.array_equal(got[:, S - 1], k)          # state
        assert np.array_equal(got[:, S], -k) and np.array_equal(got[:, S + A], k)         # action, reward
        assert np.array_equal(got[:, S + A + 1], (np.arange(n) % 7 == 0).astype(np.float32))
        assert np.array_equal(got[:, S + A + 2], (np.arange(n) % 5 == 4).astype(np.float32))
        # a full ring refuses within the timeout instead of overwriting
        for k in range(64):
            assert ring.push(np.zeros(S), np.zeros(A), 0.0, False, False, timeout_s=0.1)
        assert not ring.push(np.zeros(S), np.zeros(A), 0.0, False, False, timeout_s=0.05)
        assert len(ring.pop_all()) == 64
    finally:
        ring.detach()


def test_policy_board_versions_and_roundtrip():
    import torch.nn as nn
    net = nn.Sequential(nn.Linear(3, 4), nn.ReLU(), nn.Linear(4, 2))
    flat = flatten_state_dict(net.state_dict())
    board = PolicyBoard(None, n_floats=flat.size, create=True)
    try:
        other = PolicyBoard(board.name)
        assert other.read_if_newer(0) is None and other.version == 0
        assert board.publish(flat) == 1
        v, got = other.read_if_newer(0)
        assert v == 1 and np.array_equal(got, flat) and other.read_if_newer(1) is None
        twin = nn.Sequential(nn.Linear(3, 4), nn.ReLU(), nn.Linear(4, 2))
        unflatten_into(twin, got)
        x = t.randn(5, 3)
        assert t.equal(net(x), twin(x))
        board.publish(flat * 2)
        assert other.read_if_newer(1)[0] == 2 and not other.stopped
        board.stop()
        assert other.stopped
        other.detach()
    finally:
        board.detach()


# ---- two learner ranks under gloo, fed by ring actors ---------------------------------------------------
WS, WA, WB = 24, 6, 32


class _CpuReplay:
    """A host replay for the CPU ranks (the product's sampler is a GPU kernel): the oracle's container."""

    def __init__(self, seed):
        from oracle import oprl_oracle as orc
        self.o = orc.ReplayOracle(4000, WS, WA, max_episode_lenth=50)
        self.rs = np.random.RandomState(seed)

    def add_transition(self, s, a, r, d, episode_done=None):
        self.o.add_transition(s, a, r, d, episode_done=episode_done)

    def __len__(self):
        return len(self.o)

    def sample(self, B):
        return [t.from_numpy(np.ascontiguousarray(x)) for x in self.o.gather(self.rs.randint(0, len(self.o), size=B))]


class _Algo:
    def __init__(self, engine):
        self.learner = engine

    def get_policy_state_dict(self):
        return {f"p{i}": x for i, x in enumerate(self.learner.o.actor)}


class _Policy:
    """What an actor holds: it only has to accept the learner's parameters and act."""

    def __init__(self):
        from oracle import fixtures as fx
        self.p = [x.clone() for x in fx.make_net(1, fx.actor_dims(WS, WA))]
        self.loads = 0

    def state_dict(self):
        return {f"p{i}": x for i, x in enumerate(self.p)}

    def load_state_dict(self, sd):
        self.p = [sd[f"p{i}"] for i in range(len(self.p))]
        self.loads += 1

    def explore(self, state):
        from oracle import oprl_oracle as orc
        return orc.det_policy_forward(self.p, t.from_numpy(np.asarray(state, np.float32))[None])[0][0].numpy()


def _make_env(seed):
    from oprl_amd.environment.synthetic import SyntheticEnv
    return SyntheticEnv("walker-walk", seed=seed, episode_length=50)


def _rank(rank, world, init_file, ring_names, board_name, out):
    import torch.distributed as dist
    from oracle import fixtures as fx
    from oprl_amd.distrib.dp_learner import LearnerPlan, learner_rank_loop
    from tests.oracle_engine import OracleDDPGEngine
    t.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    eng = OracleDDPGEngine(WS, WA, fx.make_net(1, fx.actor_dims(WS, WA)), fx.make_net(2 + rank, fx.critic_dims(WS, WA)))
    rings = [TransitionRing(n) for n in ring_names]
    board = PolicyBoard(board_name)
    plan = LearnerPlan(total_updates=12, batch_size=WB, chunk=4, warmup_transitions=100, seed=3)
    checks = []
    stats = learner_rank_loop(rank, world, _Algo(eng), _CpuReplay(10 + rank), rings, board, plan, native=False,
                              on_chunk=lambda r, done, dp: checks.append(float(dp.replica_checksum().abs().max())))
    stats["checks"] = checks
    stats["actor_sum"] = float(eng.actor_arena.double().sum())
    json.dump(stats, open(f"{out}.{rank}", "w"))
    dist.destroy_process_group()


def test_two_learner_ranks_fed_by_ring_actors_gloo():
    from types import SimpleNamespace
    from oprl_amd.distrib.dp_learner import run_ring_actor
    ctx = get_context("spawn")
    cfg = SimpleNamespace(episodes_per_worker=40, episode_length=50, warmup_env_steps=60)
    n_actors, world = 4, 2
    rings = [TransitionRing(None, capacity=256, state_dim=WS, action_dim=WA, create=True) for _ in range(n_actors)]
    from oracle import fixtures as fx
    board = PolicyBoard(None, n_floats=sum(x.numel() for x in fx.make_net(1, fx.actor_dims(WS, WA))), create=True)
    with tempfile.TemporaryDirectory() as td:
        init_file, out = os.path.join(td, "rdv"), os.path.join(td, "stats")
        actors = [ctx.Process(target=run_ring_actor, args=(_make_env, _Policy, cfg, i, rings[i].name, board.name))
                  for i in range(n_actors)]
        ranks = [ctx.Process(target=_rank, args=(r, world, init_file, [rings[i].name for i in range(r, n_actors, world)],
                                                 board.name, out)) for r in range(world)]
        try:
            for p in actors + ranks:
                p.start()
            for p in ranks:
                p.join(timeout=240)
            assert [p.exitcode for p in ranks] == [0, 0]
            assert board.stopped                   # rank 0 told the actors to stop
            for p in actors:
                p.join(timeout=30)
            assert [p.exitcode for p in actors] == [0] * n_actors
            st = [json.load(open(f"{out}.{r}")) for r in range(world)]
        finally:
            for p in actors + ranks:
                if p.is_alive():
                    p.terminate()
            for r in rings:
                r.detach()
            board.detach()
    for s in st:
        assert s["updates"] == 12 and s["chunks"] >= 3 and s["received"] >= 100
        assert s["replica_spread"] == 0.0 and all(c == 0.0 for c in s["checks"])      # bit-identical replicas
    assert st[0]["actor_sum"] == st[1]["actor_sum"]
    assert st[0]["policy_version"] >= 4            # the initial policy + one publication per chunk


def test_learner_loop_ends_when_the_actors_stop_short_of_the_update_quota():
    """The round-2 advisor's livelock: every rank `ready`, the rings closed and drained, the update quota of what
    arrived (updates_per_transition x received) below total_updates — the loop slept 2 ms for ever.  It must return
    what it has done, and say so (the reference's learner gives up after learner_num_waits empty polls,
    distrib/policy_update_worker.py:55-63)."""
    from oracle import fixtures as fx
    from oprl_amd.distrib.dp_learner import LearnerPlan, learner_rank_loop
    from tests.oracle_engine import OracleDDPGEngine
    import torch.distributed as dist
    ring = TransitionRing(None, capacity=512, state_dim=WS, action_dim=WA, create=True)
    board = PolicyBoard(None, n_floats=sum(x.numel() for x in fx.make_net(1, fx.actor_dims(WS, WA))), create=True)
    td = tempfile.mkdtemp()
    dist.init_process_group("gloo", init_method=f"file://{os.path.join(td, 'rdv')}", rank=0, world_size=1)
    try:
        rs = np.random.RandomState(0)
        for k in range(200):                      # four whole episodes of 50, then the actor is gone
            ring.push(rs.standard_normal(WS).astype(np.float32), rs.uniform(-1, 1, WA).astype(np.float32), 0.5, False,
                      k % 50 == 49)
        ring.close_writer()
        eng = OracleDDPGEngine(WS, WA, fx.make_net(1, fx.actor_dims(WS, WA)), fx.make_net(2, fx.critic_dims(WS, WA)))
        plan = LearnerPlan(total_updates=1000, batch_size=WB, chunk=8, warmup_transitions=100, seed=3,
                           updates_per_transition=0.1)
        t0 = time.monotonic()
        stats = learner_rank_loop(0, 1, _Algo(eng), _CpuReplay(10), [ring], board, plan, native=False)
        assert time.monotonic() - t0 < 60
        assert stats["received"] == 200 and stats["updates"] == 20        # the quota of 200 transitions, not 1000
        assert stats["stopped_early"] and "quota" in stats["stopped_early"]
    finally:
        dist.destroy_process_group()
        ring.detach()
        board.detach()
