"""Config 5 end to end on this node: N data-parallel learner processes (one per GPU) + CPU actors over
shared-memory rings (runners/train_distrib.py::run_dp_training) on the synthetic walker-walk stand-in.
Reports environment steps/s taken in by the learners, optimiser steps/s, and how much of the wall time the
learners spent training (the rest: waiting for data / draining rings).

    python tools/bench_distrib.py [--learners 1] [--actors 32] [--updates 20000]
"""
import argparse
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

ap = argparse.ArgumentParser()
ap.add_argument("--learners", type=int, default=1)
ap.add_argument("--actors", type=int, default=32)
ap.add_argument("--updates", type=int, default=20000)
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--utd", type=float, default=1.0, help="optimiser steps per transition received")
ap.add_argument("--episode-length", type=int, default=200)
ap.add_argument("--precision", default="x2")
args = ap.parse_args()

S, A = 24, 6


def make_env(seed):
    import os
    import time as _t
    from oprl_amd.environment import make_env as mk
    env = mk("synthetic:walker-walk", seed=seed)
    pause = float(os.environ.get("OPRL_BENCH_ACTOR_SLEEP_US", "0")) * 1e-6     # a slower simulator (dm_control: ~1 ms per step)
    if pause > 0:
        step = env.step

        def slow_step(a):
            _t.sleep(pause)
            return step(a)
        env.step = slow_step
    return env


def make_policy():
    import torch.nn as nn
    from oprl_amd.algos.nn_models import DeterministicPolicy
    return DeterministicPolicy(S, A, hidden_units=(256, 256), hidden_activation=nn.ReLU(inplace=True), device="cpu")


def make_algo(logger, **kw):
    from oprl_amd.algos.ddpg import DDPG
    return DDPG(logger=logger, state_dim=S, action_dim=A, max_batch=args.batch, precision=args.precision,
                **{"device": "cuda", **kw}).create()


def make_replay_buffer(**kw):
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    return EpisodicReplayBuffer(buffer_size_transitions=1_000_000, state_dim=S, action_dim=A,
                                max_episode_lenth=args.episode_length, **{"device": "cuda", **kw}).create()


def make_logger():
    from oprl_amd.logging import NullLogger
    return NullLogger("/tmp/oprl_bench_distrib")


if __name__ == "__main__":
    from oprl_amd.distrib.dp_learner import LearnerPlan
    from oprl_amd.runners.config import DistribConfig
    from oprl_amd.runners.train_distrib import run_dp_training
    cfg = DistribConfig(batch_size=args.batch, num_env_workers=args.actors, episodes_per_worker=10 ** 6,
                        episode_length=args.episode_length, warmup_epochs=0, warmup_env_steps=1000)
    plan = LearnerPlan(total_updates=args.updates, batch_size=args.batch, chunk=500,
                       warmup_transitions=2 * args.batch, updates_per_transition=args.utd)
    t0 = time.perf_counter()
    stats = run_dp_training(make_env=make_env, make_algo=make_algo, make_policy=make_policy,
                            make_replay_buffer=make_replay_buffer, make_logger=make_logger, config=cfg,
                            learners=args.learners, plan=plan)
    wall = time.perf_counter() - t0
    recv = sum(s["received"] for s in stats)
    print(f"learners={args.learners} actors={args.actors} B={args.batch}/rank utd={args.utd}: {args.updates} optimiser steps, "
          f"{recv} env steps taken in, wall {wall:.1f} s (incl. process start-up) -> "
          f"{recv / wall:.0f} env steps/s, {args.updates / wall:.0f} optimiser steps/s; "
          f"learner loop {max(s['wall_s'] for s in stats):.1f} s of which chunks (updates + overlapped ring drain) "
          f"{max(s['train_s'] for s in stats):.1f} s = {args.updates / max(s['train_s'] for s in stats):.0f} steps/s; "
          f"[enqueue {stats[0]['enqueue_s']:.1f} s, ring drain {stats[0]['drain_s']:.1f} s, publish / GPU wait {stats[0]['publish_wait_s']:.1f} s]; "
          f"intake {recv / max(s['wall_s'] for s in stats):.0f} env steps/s over the learner loop; "
          f"replica spread {max(s['replica_spread'] for s in stats)}; policy versions {stats[0]['policy_version']}")
