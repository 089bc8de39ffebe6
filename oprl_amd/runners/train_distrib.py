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
) -> None:
    """The reference's config script passes the two worker entry points explicitly
    (configs/distrib_ddpg.py:80-90); through the ``oprl`` alias they are this package's
    (``distrib/env_worker.py``, ``distrib/policy_update_worker.py``), which is also the default."""
    from oprl_amd.distrib.env_worker import run_env_worker as _env_worker
    from oprl_amd.distrib.policy_update_worker import run_policy_update_worker as _learner
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
