"""Common base of the four algorithms + the binding between the torch-owned
parameter arenas and the HIP learner handle.

Reference: /root/reference/src/oprl/algos/base_algorithm.py:7-15 (guard +
``get_policy_state_dict``).  Everything below ``HipLearner`` is new: it replaces
autograd + torch.optim.Adam + the per-tensor Polyak loops of the reference's
``update()`` bodies with one call into liboprl_amd.so."""
from __future__ import annotations

import ctypes as C
from abc import ABC
from typing import Any, Sequence

import numpy as np
import torch as t
import torch.nn as nn

from oprl_amd import _capi
from oprl_amd.algos.nn_models import MLP, flatten_module_, is_flat


class OffPolicyAlgorithm(ABC):
    _created: bool = False

    def check_created(self) -> None:
        if not self._created:
            raise RuntimeError(
                f"Algorithm {type(self).__name__} has not been created with `create()`."
            )

    def get_policy_state_dict(self) -> dict[str, Any]:
        return self.actor.state_dict()

    def _log_update(self, step: int) -> None:
        """Scalar logging at the algorithm's cadence (the only host sync of an update)."""

    def update_from_buffer(self, replay_buffer, batch_size: int, act_next=None) -> None:
        """``update(*replay_buffer.sample(batch_size))`` as ONE C call (oprl_learner_step_n with
        K = 1): the slice kernels gather their own rows from the HBM replay with the sampler's
        Philox draw, so the per-step host work is one ctypes call instead of a gather launch, five
        output allocations and the update's argument checks.  A buffer without a device handle
        (or a gradient-exporting data-parallel learner) takes the two-call path.

        ``act_next``: the observation the NEXT environment step starts from (the trainer has it before the update,
        base_trainer.py:38-74).  The actor's forward for it rides behind the update in the same call
        (oprl_learner_step_act) and the next ``actor.explore(act_next)`` — with this very array — only collects the
        row: one host wait per environment step instead of update-sync, act-launch, act-sync."""
        handle = getattr(replay_buffer, "handle", None)
        if handle is None or self.learner.export_grads:
            self.update(*replay_buffer.sample(batch_size))
            return
        step = self.update_step
        seed = int(getattr(replay_buffer, "seed", 0))
        mlp = self._actor_mlp() if act_next is not None else None
        if mlp is not None:
            self.learner.step_act(handle, int(batch_size), seed, act_next)
            mlp.set_pending(act_next, self.learner)
        else:
            self.learner.step_n(handle, 1, int(batch_size), seed=seed)
        self._log_update(step)

    def _actor_mlp(self):
        """The actor's MLP when it is one the learner's policy kernel can run (the policy classes of nn_models.py)."""
        actor = getattr(self, "actor", None)
        mlp = getattr(actor, "mlp", None) or getattr(actor, "net", None)
        return mlp if (mlp is not None and hasattr(mlp, "set_pending") and mlp.on_gpu()) else None

    def debug_form(self, batch_size: int) -> dict[str, int]:
        """Which launch form this algorithm's learner takes for `batch_size` (HipLearner.debug_form)."""
        return self.learner.debug_form(batch_size)

    def set_seed(self, seed: int, rank: int = 0) -> None:
        """Key the learner's device-side noise streams with the run seed (and data-parallel rank)."""
        self.learner.set_seed(seed, rank)

    # full learner state (parameters, targets, optimiser moments, counters): exact resume
    def state_dict(self) -> dict[str, Any]:
        return self.learner.state_dict()

    def load_state_dict(self, sd: dict[str, Any]) -> None:
        self.learner.load_state_dict(sd)


def require_gpu(device: str) -> t.device:
    dev = t.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"device={device!r}: the oprl_amd learner runs only on an MI355X (device='cuda'); "
            "it has no CPU path")
    if not t.cuda.is_available():
        raise RuntimeError("no GPU visible to torch; the oprl_amd learner has no CPU path")
    _capi.load()
    return dev if dev.index is not None else t.device("cuda", t.cuda.current_device())


class HipLearner:
    """Owns the C handle.  ``actor_mlp`` / ``critic_mlps`` are the trainable MLPs,
    ``*_targets`` their target twins (or None); each group (actor, all critics)
    must already be flat so one Adam-state arena per group lines up with it."""

    def __init__(self, algo: str, state_dim: int, action_dim: int, device: t.device,
                 actor_group: nn.Module, actor_mlp: MLP, actor_target_mlp: MLP | None,
                 critic_group: nn.Module, critic_mlps: Sequence[MLP],
                 critic_target_group: nn.Module, critic_target_mlps: Sequence[MLP],
                 hp: dict, max_batch: int, export_grads: bool = False,
                 log_alpha: t.Tensor | None = None, actor_target_group: nn.Module | None = None,
                 no_fuse: bool = False, precision: str = "f32"):
        self.lib = _capi.load()
        self.device = device
        self.S, self.A = state_dim, action_dim
        self.max_batch = int(max_batch)
        self.export_grads = bool(export_grads)
        if actor_target_mlp is not None and actor_target_group is None:
            actor_target_group = actor_target_mlp
        self._groups = [actor_group, critic_group, critic_target_group]
        self._actor_target_group = actor_target_group
        for g in [*self._groups, actor_target_group]:
            if g is not None and not is_flat(g):
                flatten_module_(g)
        self.actor_arena = actor_group._oprl_arena
        self.critic_arena = critic_group._oprl_arena
        self.actor_m = t.zeros_like(self.actor_arena)
        self.actor_v = t.zeros_like(self.actor_arena)
        self.critic_m = t.zeros_like(self.critic_arena)
        self.critic_v = t.zeros_like(self.critic_arena)
        self.actor_grad = t.zeros_like(self.actor_arena) if export_grads else None
        self.critic_grad = t.zeros_like(self.critic_arena) if export_grads else None
        self.log_alpha = log_alpha
        self.log_alpha_m = t.zeros((), dtype=t.float64, device=device) if log_alpha is not None else None
        self.log_alpha_v = t.zeros((), dtype=t.float64, device=device) if log_alpha is not None else None
        self.log_alpha_grad = (t.zeros((), dtype=t.float64, device=device)
                               if (log_alpha is not None and export_grads) else None)

        cfg = _capi.OprlLearnerConfig()
        cfg.abi_version = _capi.OPRL_ABI_VERSION
        cfg.algo = _capi.ALGO[algo]
        if precision not in _capi.PRECISION:
            raise ValueError(f"precision={precision!r}: expected one of {sorted(_capi.PRECISION)}")
        cfg.precision = _capi.PRECISION[precision]
        self.precision = precision
        cfg.state_dim, cfg.action_dim = state_dim, action_dim
        cfg.max_batch = self.max_batch
        cfg.n_critics = len(critic_mlps)
        cfg.export_grads = int(export_grads)
        cfg.no_fuse = int(no_fuse)

        def off(arena: t.Tensor, mlp: MLP) -> int:
            return mlp.theta_ptr() - arena.data_ptr()

        def fill(dst: _capi.OprlNet, mlp: MLP, target: MLP | None, arena, m, v, g):
            o = off(arena, mlp)
            dst.n_layers = len(mlp.dims) - 1
            for i, x in enumerate(mlp.dims):
                dst.dims[i] = x
            dst.theta = mlp.theta_ptr()
            dst.theta_target = target.theta_ptr() if target is not None else None
            dst.pack = mlp.pack_tensor().data_ptr()
            dst.pack_target = target.pack_tensor().data_ptr() if target is not None else None
            dst.adam_m = m.data_ptr() + o
            dst.adam_v = v.data_ptr() + o
            dst.grad = (g.data_ptr() + o) if g is not None else None

        fill(cfg.actor, actor_mlp, actor_target_mlp, self.actor_arena, self.actor_m, self.actor_v,
             self.actor_grad)
        for j, (c, ct) in enumerate(zip(critic_mlps, critic_target_mlps)):
            fill(cfg.critics[j], c, ct, self.critic_arena, self.critic_m, self.critic_v, self.critic_grad)
        if log_alpha is not None:
            cfg.log_alpha = log_alpha.data_ptr()
            cfg.log_alpha_m = self.log_alpha_m.data_ptr()
            cfg.log_alpha_v = self.log_alpha_v.data_ptr()
            if self.log_alpha_grad is not None:
                cfg.log_alpha_grad = self.log_alpha_grad.data_ptr()
        for k, v in hp.items():
            setattr(cfg.hp, k, v)
        self._cfg = cfg
        self.algo_name = algo
        self.policy_freq = int(hp.get("policy_freq", 1))
        self._mlps = (actor_mlp, actor_target_mlp, list(critic_mlps), list(critic_target_mlps))
        self._ptrs = self._snapshot_ptrs()
        h = C.c_void_p()
        with _capi.on_device(device):
            _capi.check(self.lib.oprl_learner_create(C.byref(cfg), C.byref(h)), "oprl_learner_create")
        self.handle = h
        self._all_mlps = [m for m in [actor_mlp, actor_target_mlp, *critic_mlps, *critic_target_mlps]
                          if m is not None]
        for m in self._all_mlps:       # oprl_learner_create built every pack from the masters
            m.mark_packed()
        self._versions = self._snapshot_versions()

    def target_arenas(self):
        """Flat target-network arenas (critic targets, then the actor target if any)."""
        out = [self._groups[2]._oprl_arena]
        if self._actor_target_group is not None:
            out.append(self._actor_target_group._oprl_arena)
        return out

    # check_bound() runs on every update(): walking the module trees (``module.parameters()``) cost more
    # host time than the four kernel launches, so the Parameter objects are looked up once.  They stay
    # valid across ``module.to()`` / ``load_state_dict`` (which change ``.data`` / the version in place).
    def _param_cache(self):
        c = getattr(self, "_pcache", None)
        if c is None:
            a, at, cs, cts = self._mlps
            mlps = [a, *([at] if at is not None else []), *cs, *cts]
            first = [next(m.parameters()) for m in mlps]
            every = [p for m in [a, at, *cs, *cts] if m is not None for p in m.parameters()]
            c = self._pcache = (first, every)
        return c

    def _snapshot_ptrs(self):
        return tuple(p.data_ptr() for p in self._param_cache()[0])

    def _snapshot_versions(self):
        return tuple(p._version for p in self._param_cache()[1])

    def sync_params(self) -> None:
        """Rebuild the fragment-order packs from the master parameters.  Called
        automatically when torch reports an in-place change (load_state_dict,
        ``param.copy_``); call it yourself after writing through ``.data``."""
        with _capi.on_device(self.device):
            _capi.check(self.lib.oprl_learner_sync_params(self.handle, _capi.current_stream()),
                        "oprl_learner_sync_params")
        for m in self._all_mlps:
            m.mark_packed()
        self._versions = self._snapshot_versions()

    # ---- checkpoint / exact resume (the reference saves only the policy,
    # base_trainer.py:113-120) ---------------------------------------------------------------
    def state_dict(self) -> dict[str, Any]:
        """Everything a bit-exact resume needs: parameter / target / Adam arenas, the
        temperature and its Adam state, and the learner's counters."""
        cnt = (C.c_int64 * 4)()
        _capi.check(self.lib.oprl_learner_get_counters(self.handle, cnt), "oprl_learner_get_counters")
        t.cuda.synchronize(self.device)
        sd: dict[str, Any] = {
            "algo": self.algo_name,
            "counters": [int(x) for x in cnt],
            "actor": self.actor_arena.detach().cpu().clone(),
            "actor_m": self.actor_m.cpu().clone(), "actor_v": self.actor_v.cpu().clone(),
            "critic": self.critic_arena.detach().cpu().clone(),
            "critic_m": self.critic_m.cpu().clone(), "critic_v": self.critic_v.cpu().clone(),
            "targets": [a.detach().cpu().clone() for a in self.target_arenas()],
        }
        if self.log_alpha is not None:
            sd["log_alpha"] = [x.detach().cpu().clone() for x in (self.log_alpha, self.log_alpha_m, self.log_alpha_v)]
        return sd

    def load_state_dict(self, sd: dict[str, Any]) -> None:
        if sd.get("algo") != self.algo_name:
            raise ValueError(f"checkpoint is for {sd.get('algo')!r}, this learner is {self.algo_name!r}")
        pairs = [(self.actor_arena, sd["actor"]), (self.actor_m, sd["actor_m"]), (self.actor_v, sd["actor_v"]),
                 (self.critic_arena, sd["critic"]), (self.critic_m, sd["critic_m"]), (self.critic_v, sd["critic_v"]),
                 *zip(self.target_arenas(), sd["targets"])]
        if self.log_alpha is not None:
            pairs += list(zip((self.log_alpha, self.log_alpha_m, self.log_alpha_v), sd["log_alpha"]))
        for dst, src in pairs:
            if dst.shape != src.shape:
                raise ValueError(f"checkpoint tensor shape {tuple(src.shape)} != {tuple(dst.shape)}")
        with t.no_grad():
            for dst, src in pairs:
                dst.copy_(src.to(dst.device))
        cnt = (C.c_int64 * 4)(*[int(x) for x in sd["counters"]])
        _capi.check(self.lib.oprl_learner_set_counters(self.handle, cnt), "oprl_learner_set_counters")
        self.sync_params()      # the fragment-order packs are derived state

    def check_bound(self) -> None:
        """The kernels hold raw pointers into the arenas: refuse to run if a
        module was moved/re-allocated behind our back (e.g. ``actor.to(...)``)."""
        if self._snapshot_ptrs() != self._ptrs:
            raise RuntimeError("a network's parameters were re-allocated after create(); "
                               "the HIP learner is bound to the original arenas")
        if self._snapshot_versions() != self._versions or any(m._pack_key is None for m in self._all_mlps):
            self.sync_params()

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.oprl_learner_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ update
    def _prep(self, state, action, reward, done, next_state):
        dev = self.device
        B = state.shape[0]

        def f(x, cols):
            # fast path: a resident, contiguous fp32 batch (what the replay's sample() returns)
            if (isinstance(x, t.Tensor) and x.dtype == t.float32 and x.device == dev and x.is_contiguous()
                    and x.numel() == B * cols):
                return x
            x = t.as_tensor(x).to(device=dev, dtype=t.float32).reshape(B, cols)
            return x.contiguous()

        return B, f(state, self.S), f(action, self.A), f(reward, 1), f(done, 1), f(next_state, self.S)

    def update(self, state, action, reward, done, next_state, noise0=None, noise1=None) -> None:
        self.check_bound()
        B, s, a, r, d, s2 = self._prep(state, action, reward, done, next_state)
        n0 = None if noise0 is None else noise0.to(device=self.device, dtype=t.float32).reshape(B, self.A).contiguous()
        n1 = None if noise1 is None else noise1.to(device=self.device, dtype=t.float32).reshape(B, self.A).contiguous()
        with _capi.on_device(self.device):
            _capi.check(self.lib.oprl_learner_update(
                self.handle, _capi.ptr(s), _capi.ptr(a), _capi.ptr(r), _capi.ptr(d), _capi.ptr(s2), B,
                _capi.ptr(n0), _capi.ptr(n1), _capi.current_stream()), "oprl_learner_update")

    def update_phase(self, phase, state, action, reward, done, next_state, noise0=None, noise1=None):
        self.check_bound()
        B, s, a, r, d, s2 = self._prep(state, action, reward, done, next_state)
        n0 = None if noise0 is None else noise0.to(device=self.device, dtype=t.float32).reshape(B, self.A).contiguous()
        n1 = None if noise1 is None else noise1.to(device=self.device, dtype=t.float32).reshape(B, self.A).contiguous()
        with _capi.on_device(self.device):
            _capi.check(self.lib.oprl_learner_update_phase(
                self.handle, phase, _capi.ptr(s), _capi.ptr(a), _capi.ptr(r), _capi.ptr(d), _capi.ptr(s2),
                B, _capi.ptr(n0), _capi.ptr(n1), _capi.current_stream()), "oprl_learner_update_phase")

    def apply(self, phase: int, grad_scale: float) -> None:
        with _capi.on_device(self.device):
            _capi.check(self.lib.oprl_learner_apply(self.handle, phase, float(grad_scale),
                                                    _capi.current_stream()), "oprl_learner_apply")

    def step_n(self, replay_handle, K: int, B: int, seed: int) -> None:
        self.check_bound()
        with _capi.on_device(self.device):
            _capi.check(self.lib.oprl_learner_step_n(self.handle, replay_handle, K, B, seed,
                                                     _capi.current_stream()), "oprl_learner_step_n")

    def step_act(self, replay_handle, B: int, seed: int, obs) -> None:
        """One sample()+update() and, enqueued behind it, the actor's forward of ``obs`` with the updated weights
        (oprl_learner_step_act): nothing is waited for; ``act_wait`` collects the row."""
        self.check_bound()
        x = np.ascontiguousarray(obs, dtype=np.float32).reshape(-1)
        with _capi.on_device(self.device):
            _capi.check(self.lib.oprl_learner_step_act(self.handle, replay_handle, B, seed, x.ctypes.data_as(C.c_void_p),
                                                       _capi.current_stream()), "oprl_learner_step_act")

    def act_wait(self, n_out: int, timeout_us: int = 5_000_000):
        out = np.empty(n_out, dtype=np.float32)
        _capi.check(self.lib.oprl_learner_act_wait(self.handle, out.ctypes.data_as(C.c_void_p), n_out, timeout_us),
                    "oprl_learner_act_wait")
        return out

    def check(self) -> None:
        """Raise if a kernel of this learner reported an expired cross-workgroup wait (include/oprl_amd.h,
        "Device-side failures").  Every update / step_n / read_scalars call checks too."""
        _capi.check(self.lib.oprl_learner_check(self.handle), "oprl_learner_check")

    def clear_error(self) -> None:
        _capi.check(self.lib.oprl_learner_clear_error(self.handle), "oprl_learner_clear_error")

    DEBUG_FORM_FIELDS = ("fused", "lean", "form", "updates_per_chain_launch", "wide", "nc", "twin_split", "p2_pair", "arith",
                         "xcd_local", "shared_chip", "dp_inline_form")

    def debug_form(self, batch_size: int) -> dict[str, int]:
        """Which launch form this learner takes for `batch_size` (include/oprl_amd.h, oprl_learner_debug_form): the
        decision tests/golden/launch_forms.json pins, under that table's field names."""
        out = (C.c_int32 * 12)()
        _capi.check(self.lib.oprl_learner_debug_form(self.handle, int(batch_size), out), "oprl_learner_debug_form")
        return dict(zip(self.DEBUG_FORM_FIELDS, (int(x) for x in out)))

    def set_cluster(self, nc: int) -> None:
        """CUs per 16-row slice in the fused kernels (include/oprl_amd.h): 8 = the default (clusters of four, of
        eight where the kernels have them: a learner that has the GPU to itself), 4 = clusters of four only
        (learners that share a GPU with more than two others), 2 / 1 = least CU time per update."""
        _capi.check(self.lib.oprl_learner_set_cluster(self.handle, int(nc)), "oprl_learner_set_cluster")

    def set_seed(self, seed: int, rank: int = 0) -> None:
        self.seed = int(seed)
        _capi.check(self.lib.oprl_learner_set_seed(self.handle, int(seed) & (2 ** 64 - 1), int(rank)),
                    "oprl_learner_set_seed")

    def read_scalars(self) -> dict[str, float]:
        buf = (C.c_float * 10)()
        with _capi.on_device(self.device):
            _capi.check(self.lib.oprl_learner_read_scalars(self.handle, buf, 10, _capi.current_stream()),
                        "oprl_learner_read_scalars")
        keys = ("critic_loss", "actor_loss", "q_mean", "q_target_mean", "alpha", "update_step",
                "q1_mean", "log_pi_mean", "gauss_actor_loss", "alpha_loss")
        return dict(zip(keys, (float(x) for x in buf)))

    @property
    def update_count(self) -> int:
        n = C.c_int64()
        _capi.check(self.lib.oprl_learner_update_count(self.handle, C.byref(n)))
        return int(n.value)

    def debug_q_y(self, B: int) -> tuple[t.Tensor, t.Tensor]:
        """Per-row Q(s,a) and TD target of the last critic step (critic 0)."""
        q, y = C.c_void_p(), C.c_void_p()
        _capi.check(self.lib.oprl_learner_debug_ptrs(self.handle, C.byref(q), C.byref(y)))
        out = []
        for p in (q, y):
            buf = t.empty(B, dtype=t.float32, device=self.device)
            t.cuda.synchronize(self.device)
            C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(
                C.c_void_p(buf.data_ptr()), p, C.c_size_t(4 * B), C.c_int(3))
            out.append(buf)
        return out[0], out[1]
