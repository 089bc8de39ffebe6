"""Packed learners: N independent DDPG, TD3 or SAC learners stepped by ONE launch sequence.

The reference's ``--seeds N`` starts N training processes (runners/train.py:36-50), whatever the algorithm.  On an
MI355X one B = 256 learner is a chain of latency-bound launches that leaves most of the chip idle, so N seeds are
packed on ONE GPU: ``LearnerGroup([algo_0, ..., algo_{N-1}]).step_n(replay.handle, K, B, seeds)`` runs K updates of
every member with four kernel launches per update for the whole group (``oprl_group_step_n``; 32 exact-fp32 DDPG
members: 102k updates/s aggregate on one MI355X).  Members are of one algorithm, shape and precision; they keep their
own weights, optimiser state, sampler key and counters, and stay ordinary algorithms (``update()``, checkpoints,
``actor.explore``) between group calls; a member's parameters are bit-identical to the same learner stepped alone with
the group's launch form (``learner.set_cluster(1)`` for exact-fp32 DDPG, ``set_cluster(4)`` otherwise: include/oprl_amd.h)."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

from oprl_amd import _capi


class LearnerGroup:
    def __init__(self, algos: Sequence):
        if not algos:
            raise ValueError("LearnerGroup needs at least one algorithm")
        self.algos = list(algos)
        self.lib = _capi.load()
        self.device = self.algos[0].learner.device
        if any(a.learner.device != self.device for a in self.algos):
            raise ValueError("LearnerGroup: every member must live on the same GPU")
        handles = (C.c_void_p * len(self.algos))(*[a.learner.handle for a in self.algos])
        g = C.c_void_p()
        with _capi.on_device(self.device):
            _capi.check(self.lib.oprl_group_create(handles, len(self.algos), C.byref(g)), "oprl_group_create")
        self.handle = g

    def step_n(self, replay_handle, K: int, B: int, seeds: Sequence[int]) -> None:
        if len(seeds) != len(self.algos):
            raise ValueError("one sampler seed per member")
        for a in self.algos:
            a.learner.check_bound()
        arr = (C.c_uint64 * len(seeds))(*[int(s) & (2 ** 64 - 1) for s in seeds])
        with _capi.on_device(self.device):
            _capi.check(self.lib.oprl_group_step_n(self.handle, replay_handle, int(K), int(B), arr,
                                                   _capi.current_stream()), "oprl_group_step_n")

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.oprl_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
