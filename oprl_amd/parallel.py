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
        """Make every replica bit-identical to rank ``src`` (done once)."""
        for x in self._state_tensors():
            dist.broadcast(x, src=src, group=self.group)

    def replica_checksum(self) -> t.Tensor:
        """[max - min] over ranks of a parameter checksum: 0 iff replicas agree."""
        e = self.engine
        cs = t.stack([e.actor_arena.double().sum(), e.critic_arena.double().sum()])
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
