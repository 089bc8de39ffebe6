"""``run_training`` (reference: /root/reference/src/oprl/runners/train.py:24-86):
builds env / buffer / logger / algo from the config script's factories and runs
the trainer; ``seeds > 1`` fans out one process per seed."""
from __future__ import annotations

import logging
import os
import random
from multiprocessing import get_context
from typing import Callable

import numpy as np
import torch as t

from oprl_amd.algos.protocols import AlgorithmProtocol
from oprl_amd.buffers.protocols import ReplayBufferProtocol
from oprl_amd.environment.protocols import EnvProtocol
from oprl_amd.logging import LoggerProtocol
from oprl_amd.runners.config import CommonParameters
from oprl_amd.trainers.base_trainer import BaseTrainer


def set_seed(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    t.manual_seed(seed)


def _run_training_func(make_algo, make_env, make_replay_buffer, make_logger,
                       config: CommonParameters, seed: int, **trainer_kwargs) -> None:
    set_seed(seed)
    env = make_env(seed)
    replay_buffer = make_replay_buffer()
    logger = make_logger(seed)
    algo = make_algo(logger)
    # the run seed reaches the DEVICE-side generators too (the reference draws minibatch indices from
    # np.random and the in-update noise from torch after set_seed(seed), runners/train.py:14-21): the
    # replay's Philox sampler key and the learner's noise key
    if hasattr(replay_buffer, "seed"):
        replay_buffer.seed = seed
    if hasattr(algo, "set_seed"):
        algo.set_seed(seed)
    if env.env_family not in ("dm_control", "gymnasium", "synthetic"):
        raise ValueError(f"Unsupported env family: {env.env_family}")
    BaseTrainer(env=env, make_env_test=make_env, algo=algo, replay_buffer=replay_buffer,
                num_steps=config.num_steps, eval_interval=config.eval_every, device=config.device,
                estimate_q_every=config.estimate_q_every, stdout_log_every=config.log_every,
                seed=seed, logger=logger, **trainer_kwargs).train()


def run_training(
    make_algo: Callable[[LoggerProtocol], AlgorithmProtocol],
    make_env: Callable[[int], EnvProtocol],
    make_replay_buffer: Callable[[], ReplayBufferProtocol],
    make_logger: Callable[[int], LoggerProtocol],
    config: CommonParameters,
    seeds: int = 1,
    start_seed: int = 0,
    **trainer_kwargs,
) -> None:
    if seeds == 1:
        _run_training_func(make_algo, make_env, make_replay_buffer, make_logger, config, 0, **trainer_kwargs)
        return
    # the seeds share ONE GPU: launches whose workgroups wait for each other inside the launch (clusters of eight, the
    # merged phase + tile launches, the whole-update launch: include/oprl_amd.h) need their workgroups co-resident, and a
    # sibling process holding the compute units turns a bounded wait into a poisoned update — not into a slowdown.
    # From two seeds on the children inherit the forms that only ever wait within a cluster of four
    os.environ.setdefault("OPRL_AMD_NO_WIDE", "1")
    os.environ.setdefault("OPRL_AMD_FORM", "plain")
    ctx = get_context("spawn")   # a forked child cannot re-initialise the GPU runtime
    procs = [ctx.Process(target=_run_training_func,
                         args=(make_algo, make_env, make_replay_buffer, make_logger, config, seed),
                         kwargs=trainer_kwargs)
             for seed in range(start_seed, start_seed + seeds)]
    for i, p in enumerate(procs):
        p.start()
        logging.info(f"Starting process {i}...")
    for p in procs:
        p.join()
    logging.info("Training finished.")
