"""``run_distrib_training`` (reference:
/root/reference/src/oprl/runners/train_distrib.py:14-40): N CPU actor processes
+ one learner, connected by in-host queues instead of RabbitMQ."""
from __future__ import annotations

from multiprocessing import get_context
from typing import Callable

from oprl_amd.distrib.queue import QueueHub
from oprl_amd.runners.config import DistribConfig


def run_distrib_training(
    run_env_worker: Callable | None = None,
    run_policy_update_worker: Callable | None = None,
    *,
    make_env: Callable,
    make_algo: Callable,
    make_policy: Callable,
    make_replay_buffer: Callable,
    make_logger: Callable,
    config: DistribConfig,
    max_epochs: int | None = None,
    learners: int = 1,
    plan=None,
):
    """The reference's config script passes the two worker entry points explicitly
    (configs/distrib_ddpg.py:80-90); through the ``oprl`` alias they are this package's
    (``distrib/env_worker.py``, ``distrib/policy_update_worker.py``), which is also the default."""
    from oprl_amd.distrib.env_worker import run_env_worker as _env_worker
    from oprl_amd.distrib.policy_update_worker import run_policy_update_worker as _learner
    if learners > 1 or plan is not None:
        # BASELINE.json config 5: data-parallel learners (one per GPU) fed over shared-memory rings
        return run_dp_training(make_env=make_env, make_algo=make_algo, make_policy=make_policy,
                               make_replay_buffer=make_replay_buffer, make_logger=make_logger, config=config,
                               learners=learners, plan=plan)
    run_env_worker = run_env_worker or _env_worker
    run_policy_update_worker = run_policy_update_worker or _learner
    ctx = get_context("spawn")
    names = [f"{kind}_{i}" for i in range(config.num_env_workers) for kind in ("env", "policy")]
    hub = QueueHub(names, ctx)
    workers = [ctx.Process(target=run_env_worker, args=(make_env, make_policy, config, i, hub))
               for i in range(config.num_env_workers)]
    for p in workers:
        p.start()
    try:
        run_policy_update_worker(make_algo, make_env, make_replay_buffer, make_logger, config, hub,
                                 max_epochs=max_epochs)
    finally:
        for p in workers:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()


# ------------------------------------------------------------------------------------------------------
# BASELINE.json config 5: N data-parallel learner processes + CPU actors over shared-memory rings
# (oprl_amd/distrib/dp_learner.py, oprl_amd/distrib/shm.py)
# ------------------------------------------------------------------------------------------------------
def _call_with_overrides(factory: Callable, what: str, *args, **overrides):
    try:
        return factory(*args, **overrides)
    except TypeError as exc:
        raise TypeError(f"{what} must accept the keyword overrides {sorted(overrides)} for a data-parallel run "
                        f"(see configs/distrib_ddpg.py): {exc}") from exc


def _dp_learner_main(rank: int, world: int, init_file: str, make_algo, make_replay_buffer, make_logger, plan,
                     ring_names, board_name: str, result_path: str, backend: str) -> None:
    import json
    import os

    import torch as t
    import torch.distributed as dist

    from oprl_amd.distrib.dp_learner import learner_rank_loop
    from oprl_amd.distrib.shm import PolicyBoard, TransitionRing
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    t.set_num_threads(1)      # a learner rank does no CPU math: no intra-op pool whose idle workers spin beside the launch loop
    device = "cpu"
    kw = {}
    if backend == "nccl":
        t.cuda.set_device(rank)
        device = f"cuda:{rank}"
        kw["device_id"] = t.device(device)
    dist.init_process_group(backend, init_method=f"file://{init_file}", rank=rank, world_size=world, **kw)
    t.manual_seed(plan.seed)                        # every replica starts from the same parameters
    algo = _call_with_overrides(make_algo, "make_algo", make_logger(), device=device,
                                export_grads=world > 1 or bool(getattr(plan, "force_exchange", False)))
    if hasattr(algo, "set_seed"):
        algo.set_seed(plan.seed, rank)              # the run seed reaches the device-side noise keys, per rank
    buffer = _call_with_overrides(make_replay_buffer, "make_replay_buffer", device=device, seed=plan.seed * 1000 + rank)
    rings = [TransitionRing(n) for n in ring_names]
    board = PolicyBoard(board_name)
    try:
        stats = learner_rank_loop(rank, world, algo, buffer, rings, board, plan)
        with open(f"{result_path}.{rank}", "w") as f:
            json.dump(stats, f)
    finally:
        board.stop() if rank == 0 else None
        for r in rings:
            r.detach()
        board.detach()
        dist.destroy_process_group()


def run_dp_training(*, make_env, make_algo, make_policy, make_replay_buffer, make_logger, config: DistribConfig,
                    learners: int, plan=None, backend: str = "nccl") -> list[dict]:
    """Spawn ``config.num_env_workers`` ring actors and ``learners`` learner ranks (rank r on cuda:r); actor i
    feeds rank ``i % learners``.  Returns the ranks' statistics."""
    import json
    import os
    import tempfile

    from oprl_amd.distrib.dp_learner import LearnerPlan, run_ring_actor
    from oprl_amd.distrib.shm import PolicyBoard, TransitionRing, flatten_state_dict
    if plan is None:
        plan = LearnerPlan(total_updates=config.episode_length * config.num_env_workers
                           * max(config.episodes_per_worker - config.warmup_epochs - 1, 1),
                           batch_size=config.batch_size,
                           warmup_transitions=(config.warmup_epochs + 1) * config.episode_length
                           * max(config.num_env_workers // learners, 1))
    probe = make_env(seed=0)
    S, A = int(probe.observation_space.shape[0]), int(probe.action_space.shape[0])
    n_policy = int(flatten_state_dict(make_policy().state_dict()).size)
    ctx = get_context("spawn")
    rings = [TransitionRing(None, capacity=4 * config.episode_length, state_dim=S, action_dim=A, create=True)
             for _ in range(config.num_env_workers)]
    board = PolicyBoard(None, n_floats=n_policy, create=True)
    tmp = tempfile.mkdtemp(prefix="oprl_dp_")
    init_file, result = os.path.join(tmp, "rendezvous"), os.path.join(tmp, "stats")
    actors = [ctx.Process(target=run_ring_actor, args=(make_env, make_policy, config, i, rings[i].name, board.name))
              for i in range(config.num_env_workers)]
    ranks = [ctx.Process(target=_dp_learner_main,
                         args=(r, learners, init_file, make_algo, make_replay_buffer, make_logger, plan,
                               [rings[i].name for i in range(r, config.num_env_workers, learners)], board.name,
                               result, backend))
             for r in range(learners)]
    try:
        # one core per actor: the BLAS / OpenMP pools of 32 actor processes would otherwise each spin up a
        # thread per host core (children inherit the environment they are started with)
        saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
        os.environ.update({k: "1" for k in saved})
        try:
            for p in actors:
                p.start()
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        for p in ranks:
            p.start()
        # poll instead of a blind join: a rank that died takes the others down with it (they would wait in a
        # collective for ever), and the whole run has a wall-clock bound
        import time
        t_end = None if getattr(plan, "wall_timeout_s", None) is None else time.monotonic() + plan.wall_timeout_s
        while any(p.is_alive() for p in ranks):
            dead = [p.exitcode for p in ranks if p.exitcode not in (None, 0)]
            if dead or (t_end is not None and time.monotonic() > t_end):
                for p in ranks:
                    if p.is_alive():
                        p.terminate()
                board.stop()
                raise RuntimeError(f"learner rank(s) failed with exit codes {dead}" if dead else
                                   f"data-parallel training exceeded its wall-clock bound of {plan.wall_timeout_s} s")
            time.sleep(0.2)
        board.stop()
        for p in actors:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
        bad = [p.exitcode for p in ranks if p.exitcode != 0]
        if bad:
            raise RuntimeError(f"learner rank(s) failed with exit codes {bad}")
        return [json.load(open(f"{result}.{r}")) for r in range(learners)]
    finally:
        for p in actors + ranks:
            if p.is_alive():
                p.terminate()
        for r in rings:
            r.detach()
        board.detach()
