"""Synchronous data-parallel learner: one process per GPU, gradients all-reduced
with RCCL over xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).

New functionality mandated by BASELINE.json (the reference has a single learner
fed over pika/RabbitMQ, distrib/policy_update_worker.py:45-76; no gradient
exchange exists there).  Partitioning: parameters, targets and
Adam state replicated (broadcast once from rank 0); every rank owns a disjoint
replay shard and samples its own minibatch; per update the critic gradients are
summed across ranks and scaled by 1/world *before* the critic Adam step, then
the actor gradients likewise — two reductions per update are required because
the actor loss must see the post-Adam critic (ddpg.py:69-70).  Polyak is local
(identical on all ranks).

The class only needs an *engine* with ``update_phase(phase, batch..)``,
``apply(phase, scale)`` and flat ``critic_grad`` / ``actor_grad`` tensors (the
HipLearner created with export_grads=True), so the host logic is also exercised
on CPU with the gloo backend in tests/test_parallel_gloo.py.
"""
from __future__ import annotations

import torch as t
import torch.distributed as dist


class DataParallelLearner:
    def __init__(self, algo, group=None, engine=None):
        self.algo = algo
        self.engine = engine if engine is not None else algo.learner
        if not getattr(self.engine, "export_grads", False):
            raise RuntimeError("DataParallelLearner needs a learner created with export_grads=True")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if hasattr(self.engine, "set_seed"):       # every rank draws its own in-update noise
            self.engine.set_seed(getattr(self.engine, "seed", 0), self.rank)

    # ---- native (C) data-parallel loop -------------------------------------------
    def init_native_comm(self, rccl_path: str | None = None) -> None:
        """Give the C library its own RCCL communicator so the whole synchronous
        update (phase -> all-reduce -> Adam -> phase -> all-reduce -> Adam) runs in C
        (``oprl_learner_dp_update`` / ``dp_step_n``) without a python round trip per
        collective.  The 128-byte unique id travels over torch.distributed."""
        import ctypes as C
        import os

        from oprl_amd import _capi
        lib = _capi.load()
        if rccl_path is None:
            rccl_path = os.path.join(os.path.dirname(t.__file__), "lib", "librccl.so")
        path = rccl_path.encode()
        box = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            _capi.check(lib.oprl_comm_unique_id(path, buf), "oprl_comm_unique_id")
            box[0] = buf.raw
        dist.broadcast_object_list(box, src=0, group=self.group)
        with t.cuda.device(self.engine.device):
            _capi.check(lib.oprl_comm_init(self.engine.handle, path, self.rank, self.world, box[0]),
                        "oprl_comm_init")
        self._native = True

    def init_p2p(self, level: int = 2) -> bool:
        """``level``: see set_p2p_level (2 = also inside the dW launches of fused learners).
        Set up the one-shot all-reduce over peer windows (csrc/p2p.hip): every rank allocates its
        window, the IPC handles travel over torch.distributed, every rank maps the others' windows, then
        all ranks run the self-test together and keep the path only if EVERY rank's sum was exact
        (otherwise the RCCL communicator of init_native_comm() keeps doing the exchanges).  Returns
        whether the windows are in use."""
        import ctypes as C

        from oprl_amd import _capi
        lib = _capi.load()
        e = self.engine
        ok = True
        self.p2p_error = ""

        def failed(what: str) -> None:
            self.p2p_error = f"{what}: {lib.oprl_last_error().decode('utf-8', 'replace')}"

        buf = C.create_string_buffer(64)
        with t.cuda.device(e.device):
            ok = lib.oprl_p2p_create(e.handle, self.rank, self.world, buf) == 0
        if not ok:
            failed("oprl_p2p_create")
        handles = [None] * self.world
        dist.all_gather_object(handles, buf.raw if ok else None, group=self.group)
        ok = ok and all(h is not None for h in handles)
        if ok:
            with t.cuda.device(e.device):
                ok = lib.oprl_p2p_connect(e.handle, b"".join(handles)) == 0
            if not ok:
                failed("oprl_p2p_connect")
        verdicts = [None] * self.world
        dist.all_gather_object(verdicts, bool(ok), group=self.group)
        if not all(verdicts):
            return False                     # some rank could not map a window: nobody enters the self-test
        with t.cuda.device(e.device):
            ok = lib.oprl_p2p_selftest(e.handle, _capi.current_stream()) == 0
        if not ok:
            failed("oprl_p2p_selftest")
        dist.all_gather_object(verdicts, bool(ok), group=self.group)
        good = all(verdicts)
        if good:
            _capi.check(lib.oprl_p2p_enable(e.handle, int(level)), "oprl_p2p_enable")
            self._native = True
        self.p2p = good
        self.p2p_level = int(level) if good else 0
        return good

    def set_p2p_level(self, level: int) -> None:
        """0: RCCL; 1: one window kernel per exchange; 2: fused learners exchange inside their dW launches
        (all ranks must switch together)."""
        from oprl_amd import _capi
        _capi.check(_capi.load().oprl_p2p_enable(self.engine.handle, int(level)), "oprl_p2p_enable")
        self.p2p_level = int(level)

    def healthy(self) -> bool:
        """Collective: parameters finite on this rank and replica checksums identical across ranks."""
        e = self.engine
        finite = bool(t.isfinite(e.actor_arena).all() and t.isfinite(e.critic_arena).all())
        flags = [None] * self.world
        dist.all_gather_object(flags, finite, group=self.group)
        if not all(flags):
            return False
        return float(self.replica_checksum().abs().max()) == 0.0

    def step_n(self, replay_handle, K: int, B: int, seed: int = 0) -> None:
        """K synchronous data-parallel sample()+update() iterations in one C call."""
        from oprl_amd import _capi
        if not getattr(self, "_native", False):
            raise RuntimeError("call init_native_comm() first")
        e = self.engine
        e.check_bound()
        with t.cuda.device(e.device):
            _capi.check(e.lib.oprl_learner_dp_step_n(e.handle, replay_handle, K, B, seed,
                                                     _capi.current_stream()), "oprl_learner_dp_step_n")

    # ---- replica management ---------------------------------------------------
    def _state_tensors(self):
        e = self.engine
        out = [e.actor_arena, e.critic_arena, e.actor_m, e.actor_v, e.critic_m, e.critic_v]
        for g in e.target_arenas():
            out.append(g)
        if getattr(e, "log_alpha", None) is not None:
            out += [e.log_alpha, e.log_alpha_m, e.log_alpha_v]
        return out

    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every replica bit-identical to rank ``src`` (done once).  With the library's own RCCL
        communicator (init_native_comm) this is one C call, oprl_comm_broadcast_params; otherwise
        torch.distributed broadcasts of the same arenas (any backend; the gloo tests)."""
        if getattr(self, "_native", False) and getattr(self.engine, "handle", None) is not None \
                and not getattr(self, "p2p", False):
            from oprl_amd import _capi
            with t.cuda.device(self.engine.device):
                _capi.check(self.engine.lib.oprl_comm_broadcast_params(self.engine.handle, int(src),
                                                                       _capi.current_stream()),
                            "oprl_comm_broadcast_params")
            self.engine.sync_params()
            return
        for x in self._state_tensors():
            dist.broadcast(x, src=src, group=self.group)
        if hasattr(self.engine, "sync_params"):
            self.engine.sync_params()      # the fragment packs are derived from the arenas just overwritten

    def replica_checksum(self) -> t.Tensor:
        """[max - min] over ranks of a parameter checksum: 0 iff replicas agree."""
        e = self.engine
        cs = t.stack([e.actor_arena.double().sum(), e.critic_arena.double().sum()])
        if dist.get_backend(self.group) == "gloo":
            cs = cs.cpu()                      # (gloo reduces on the host)
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        return hi - lo

    # ---- one synchronous update -------------------------------------------------
    def actor_due(self) -> bool:
        e = self.engine
        return e.algo_name != "td3" or (e.update_count % e.policy_freq == 0)

    def update(self, state, action, reward, done, next_state, noise0=None, noise1=None) -> None:
        e = self.engine
        scale = 1.0 / self.world
        e.update_phase(0, state, action, reward, done, next_state, noise0, noise1)
        dist.all_reduce(e.critic_grad, op=dist.ReduceOp.SUM, group=self.group)
        e.apply(0, scale)
        due = self.actor_due()
        e.update_phase(1, state, action, reward, done, next_state, noise0, noise1)
        if due:
            dist.all_reduce(e.actor_grad, op=dist.ReduceOp.SUM, group=self.group)
            if getattr(e, "log_alpha_grad", None) is not None:
                dist.all_reduce(e.log_alpha_grad, op=dist.ReduceOp.SUM, group=self.group)
            e.apply(1, scale)
