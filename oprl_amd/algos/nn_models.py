"""Networks of the off-policy learner, arena-backed.

Same classes, constructor arguments and state_dict keys as the reference
(/root/reference/src/oprl/algos/nn_models.py: MLP :84-107, Critic :27-49,
DoubleCritic :52-81, DeterministicPolicy :110-150, GaussianActor :153-194), so
``actor.state_dict()``, ``t.save(actor)`` and ``load_state_dict`` interoperate
with a plain reference module.  What differs is underneath:

* every module's Parameters are *views into one flat fp32 arena* laid out in
  state_dict order (W0,b0,W1,b1,...) — the layout the HIP learner binds to
  (``oprl_net.theta``) and updates in place;
* on a GPU tensor ``forward`` runs the hand-written gfx950 slice kernel through
  ``oprl_mlp_forward`` (cat-free: state and action are read from two pointers).
  On CPU tensors (actor processes of the distributed setup, which only ever do
  B=1 explore/exploit) it is plain torch arithmetic; the learner itself has no
  CPU path.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Final, Sequence

import numpy as np
import numpy.typing as npt
import torch as t
import torch.nn as nn
from torch.nn.functional import logsigmoid

from oprl_amd import _capi

LOG_STD_MIN_MAX: Final[tuple[float, float]] = (-20, 2)
_INIT_LOCK = threading.Lock()


def initialize_weight_orthogonal(m: nn.Module, gain: float = nn.init.calculate_gain("relu")) -> None:
    """Orthogonal weights (gain sqrt(2)) and zero bias for Linear layers
    (reference nn_models.py:14-24; its conv branch is unused on this path)."""
    if isinstance(m, nn.Linear):
        # (the QR behind orthogonal_ on ONE host thread: on a 256-core GPU host torch's intra-op pool turned each of these
        # 256 x 256 factorisations into tens of milliseconds of thread wake-ups — 100-200 ms per DDPG / TD3 create())
        # (torch's thread count is process-global: two host threads creating learners at once — the multi-seed layout —
        # must not interleave their save / restore, or the process stays on one thread for good: ADVICE r4)
        with _INIT_LOCK:
            n = t.get_num_threads()
            t.set_num_threads(1)
            try:
                nn.init.orthogonal_(m.weight.data, gain)
            finally:
                t.set_num_threads(n)
        m.bias.data.zero_()


# --------------------------------------------------------------------------- #
# flat parameter arenas                                                       #
# --------------------------------------------------------------------------- #
def flatten_module_(module: nn.Module) -> t.Tensor:
    """Re-home all parameters of ``module`` as views of one contiguous fp32
    tensor (parameters() order == state_dict order) and return that tensor.
    Values are preserved.  Call again after ``module.to(device)``."""
    params = list(module.parameters())
    if not params:
        raise ValueError("module has no parameters")
    dev = params[0].device
    flat = t.empty(sum(p.numel() for p in params), dtype=t.float32, device=dev)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view(p.shape)
        off += n
    module._oprl_arena = flat  # keeps the storage alive with the module
    for m in module.modules():
        if isinstance(m, MLP):
            m.mark_dirty()
    return flat


def is_flat(module: nn.Module) -> bool:
    """True if the parameters are contiguous, in order, in one storage."""
    arena = getattr(module, "_oprl_arena", None)
    if arena is None:
        return False
    expect = arena.data_ptr()
    for p in module.parameters():
        if p.data_ptr() != expect or p.dtype != t.float32:
            return False
        expect += p.numel() * 4
    return True


def ensure_flat(module: nn.Module) -> t.Tensor:
    if not is_flat(module):
        flatten_module_(module)
    return module._oprl_arena


def _net_desc(dims: Sequence[int], theta_ptr: int, target_ptr: int = 0, m_ptr: int = 0,
              v_ptr: int = 0, grad_ptr: int = 0, pack_ptr: int = 0,
              pack_target_ptr: int = 0) -> _capi.OprlNet:
    d = _capi.OprlNet()
    d.n_layers = len(dims) - 1
    for i, x in enumerate(dims):
        d.dims[i] = int(x)
    d.theta, d.theta_target, d.adam_m, d.adam_v, d.grad, d.pack, d.pack_target = (
        C.c_void_p(theta_ptr), C.c_void_p(target_ptr), C.c_void_p(m_ptr), C.c_void_p(v_ptr),
        C.c_void_p(grad_ptr), C.c_void_p(pack_ptr), C.c_void_p(pack_target_ptr))
    return d


def pack_floats(dims: Sequence[int]) -> int:
    """Size of one fragment-order pack buffer (csrc/engine.h): per layer a forward
    pack of W and a backward pack of W^T, each ceil(out/16)*ceil(in/16) KB."""
    return sum(2 * (-(-o // 16)) * (-(-i // 16)) * 256 for i, o in zip(dims[:-1], dims[1:]))


class MLP(nn.Module):
    """Linear -> act -> ... -> Linear -> out_act, registered as ``self.nn``
    (an ``nn.Sequential``) so parameter keys are ``nn.{0,2,4,..}.{weight,bias}``."""

    def __init__(
        self,
        input_dim: int,
        output_dim: int,
        hidden_units: tuple[int, ...] = (64, 64),
        hidden_activation: nn.Module = nn.Tanh(),
        output_activation: nn.Module = nn.Identity(),
    ):
        super().__init__()
        self.dims = [int(input_dim), *[int(h) for h in hidden_units], int(output_dim)]
        mods: list[nn.Module] = []
        for fan_in, fan_out in zip(self.dims[:-2], self.dims[1:-1]):
            mods += [nn.Linear(fan_in, fan_out), hidden_activation]
        mods += [nn.Linear(self.dims[-2], self.dims[-1]), output_activation]
        self.nn = nn.Sequential(*mods)
        self._hip_ok = isinstance(hidden_activation, nn.ReLU) and isinstance(output_activation, nn.Identity)
        self._pack: t.Tensor | None = None   # fragment-order weight packs (device)
        self._pack_key = None
        self._plist: list[nn.Parameter] | None = None   # the Parameter objects (they outlive .data swaps)
        self._desc_key = None
        self._desc = None

    def __getstate__(self):
        # t.save(actor) pickles the whole module (base_trainer.py:113-120): the cached C descriptor
        # (ctypes, holds raw pointers) and the Parameter list are rebuilt on first use after loading
        state = self.__dict__.copy()
        state["_desc"] = None
        state["_desc_key"] = None
        state["_plist"] = None
        return state

    def _params(self) -> list[nn.Parameter]:
        # walking the module tree costs ~25 us per call; the Parameter OBJECTS of a built MLP never
        # change (flatten_module_, load_state_dict and .to() swap or fill their .data)
        if self._plist is None:
            self._plist = list(self.parameters())
        return self._plist

    # -- HIP path ------------------------------------------------------------
    def theta_ptr(self) -> int:
        """Device pointer of this MLP's W0 (its parameters must be flat; they
        are when the MLP or an ancestor went through flatten_module_)."""
        ps = self._params()
        expect = ps[0].data_ptr()
        for p in ps:
            if p.data_ptr() != expect:
                flatten_module_(self)
                return ps[0].data_ptr()
            expect += p.numel() * 4
        return ps[0].data_ptr()

    # -- fragment-order packs ---------------------------------------------------
    def _param_key(self):
        ps = self._params()
        return tuple(p.data_ptr() for p in ps) + tuple(p._version for p in ps)

    def mark_dirty(self) -> None:
        """Call after changing parameters through an alias torch cannot see
        (``p.data``, the flat arena): forces a repack before the next HIP use."""
        self._pack_key = None

    def mark_packed(self) -> None:
        """The packs were just rebuilt/updated by the library (learner kernels)."""
        self._pack_key = self._param_key()

    def pack_tensor(self) -> t.Tensor:
        """The pack buffer (allocated on first use, on the parameters' device)."""
        dev = self._params()[0].device
        if self._pack is None or self._pack.device != dev:
            self._pack = t.zeros(pack_floats(self.dims), dtype=t.float32, device=dev)
            self._pack_key = None
        return self._pack

    def ensure_packed(self) -> t.Tensor:
        return self._packed_desc()[0]

    def _packed_desc(self):
        """(pack tensor, C descriptor of this net with its packs), the packs rebuilt if the master
        parameters changed behind them."""
        pk = self.pack_tensor()
        theta = self.theta_ptr()
        key = self._param_key()
        if key != self._pack_key:
            desc = _net_desc(self.dims, theta, pack_ptr=pk.data_ptr())
            with _capi.on_device(pk.device):
                _capi.check(_capi.load().oprl_net_repack(C.byref(desc), 1, _capi.current_stream()),
                            "oprl_net_repack")
            self._pack_key = self._param_key()
        dkey = (theta, pk.data_ptr())
        if dkey != self._desc_key:
            self._desc = _net_desc(self.dims, theta, pack_ptr=pk.data_ptr())
            self._desc_key = dkey
        return pk, self._desc

    def hip_forward(self, x0: t.Tensor, x1: t.Tensor | None = None, out_act: int = _capi.ACT_NONE) -> t.Tensor:
        if not self._hip_ok:
            raise RuntimeError("the HIP MLP kernels implement ReLU hidden / identity output only")
        lib = _capi.load()
        x0 = x0.detach().to(t.float32).contiguous()
        k0 = x0.shape[-1]
        k1 = 0
        if x1 is not None:
            x1 = x1.detach().to(t.float32).contiguous()
            k1 = x1.shape[-1]
        B = x0.shape[0]
        n_out = self.dims[-1] // 2 if out_act == _capi.ACT_GAUSS_MEAN else self.dims[-1]
        out = t.empty((B, n_out), dtype=t.float32, device=x0.device)
        _, desc = self._packed_desc()
        with _capi.on_device(x0.device):
            _capi.check(lib.oprl_mlp_forward(C.byref(desc), 0, _capi.ptr(x0), k0, _capi.ptr(x1), k1,
                                             B, out_act, _capi.ptr(out), _capi.current_stream()),
                        "oprl_mlp_forward")
        return out

    def hip_act(self, obs: npt.NDArray, out_act: int = _capi.ACT_NONE) -> npt.NDArray:
        """One observation (host) -> one output row (host) through oprl_mlp_act: the per-env-step
        policy call without torch tensors on the way."""
        if not self._hip_ok:
            raise RuntimeError("the HIP MLP kernels implement ReLU hidden / identity output only")
        pending = self.__dict__.get("_pending")
        if pending is not None:
            self.__dict__["_pending"] = None
            # the row for exactly this observation was requested with the last update (set_pending): collect it
            p_obs, learner = pending
            raw = learner.act_wait(self.dims[-1])
            if p_obs is obs and out_act == _capi.ACT_NONE:
                return raw
        x = np.ascontiguousarray(obs, dtype=np.float32).reshape(-1)
        n_out = self.dims[-1] // 2 if out_act == _capi.ACT_GAUSS_MEAN else self.dims[-1]
        out = np.empty(n_out, dtype=np.float32)
        _, desc = self._packed_desc()
        with _capi.on_device(self._params()[0].device):
            _capi.check(_capi.load().oprl_mlp_act(C.byref(desc), x.ctypes.data_as(C.c_void_p), x.shape[0], out_act,
                                                  out.ctypes.data_as(C.c_void_p), n_out, _capi.current_stream()),
                        "oprl_mlp_act")
        return out

    def on_gpu(self) -> bool:
        return self._params()[0].is_cuda

    def set_pending(self, obs, learner) -> None:
        """The learner has this net's forward of ``obs`` in flight behind its last update (oprl_learner_step_act):
        the next ``hip_act`` with the same array collects it instead of launching."""
        self.__dict__["_pending"] = (obs, learner)        # (not nn.Module.__setattr__: microseconds per env step)

    def forward(self, x: t.Tensor) -> t.Tensor:
        if x.is_cuda:
            return self.hip_forward(x)
        return self.nn(x)


def _forward_sa(mlp: MLP, states: t.Tensor, actions: t.Tensor) -> t.Tensor:
    if states.is_cuda:
        return mlp.hip_forward(states, actions)
    return mlp.nn(t.cat([states, actions], dim=-1))


class Critic(nn.Module):
    def __init__(
        self,
        state_dim: int,
        action_dim: int,
        hidden_units: tuple[int, ...] = (256, 256),
        hidden_activation: nn.Module = nn.ReLU(inplace=True),
    ) -> None:
        super().__init__()
        self.q1 = MLP(state_dim + action_dim, 1, hidden_units, hidden_activation)

    def forward(self, states: t.Tensor, actions: t.Tensor) -> t.Tensor:
        return _forward_sa(self.q1, states, actions)

    def Q1(self, states: t.Tensor, actions: t.Tensor) -> t.Tensor:
        return _forward_sa(self.q1, states, actions)


class DoubleCritic(nn.Module):
    def __init__(
        self,
        state_dim: int,
        action_dim: int,
        hidden_units: tuple[int, ...] = (256, 256),
        hidden_activation: nn.Module = nn.ReLU(inplace=True),
    ):
        super().__init__()
        self.q1 = MLP(state_dim + action_dim, 1, hidden_units, hidden_activation)
        self.q2 = MLP(state_dim + action_dim, 1, hidden_units, hidden_activation)

    def forward(self, states: t.Tensor, actions: t.Tensor) -> tuple[t.Tensor, t.Tensor]:
        return _forward_sa(self.q1, states, actions), _forward_sa(self.q2, states, actions)

    def Q1(self, states: t.Tensor, actions: t.Tensor) -> t.Tensor:
        return _forward_sa(self.q1, states, actions)


class DeterministicPolicy(nn.Module):
    def __init__(
        self,
        state_dim: int,
        action_dim: int,
        hidden_units: tuple[int, ...] = (256, 256),
        hidden_activation: nn.Module = nn.ReLU(inplace=True),
        max_action: float = 1.0,
        expl_noise: float = 0.1,
        device: str = "cpu",
    ):
        super().__init__()
        self.mlp = MLP(state_dim, action_dim, hidden_units, hidden_activation)
        self.mlp.apply(initialize_weight_orthogonal)
        self._device = device
        self._action_shape = action_dim
        self._max_action = max_action
        self._expl_noise = expl_noise

    def forward(self, states: t.Tensor) -> t.Tensor:
        if states.is_cuda:
            return self.mlp.hip_forward(states, out_act=_capi.ACT_TANH)
        return t.tanh(self.mlp.nn(states))

    def exploit(self, state: npt.NDArray) -> npt.NDArray:
        if self.mlp.on_gpu():
            return self.mlp.hip_act(state, _capi.ACT_TANH)
        s = t.as_tensor(state, dtype=t.float32, device=self._device).unsqueeze(0)
        with t.no_grad():
            return self.forward(s).cpu().numpy().reshape(-1)

    def explore(self, state: npt.NDArray) -> npt.NDArray:
        # reference quirk kept: NO tanh on the exploration path (nn_models.py:144-150)
        noise = (t.randn(self._action_shape) * self._expl_noise).numpy()
        if self.mlp.on_gpu():
            raw = self.mlp.hip_act(state) + noise
        else:
            s = t.as_tensor(state, dtype=t.float32, device=self._device).unsqueeze(0)
            with t.no_grad():
                raw = self.mlp(s)[0].numpy() + noise
        return np.clip(raw, -self._max_action, self._max_action)


class TanhNormal:
    """tanh-squashed diagonal Gaussian (reference nn_models.py:197-214)."""

    def __init__(self, normal_mean: t.Tensor, normal_std: t.Tensor, device: str | None = None) -> None:
        self.normal_mean = normal_mean
        self.normal_std = normal_std

    def rsample(self, eps: t.Tensor | None = None) -> tuple[t.Tensor, t.Tensor]:
        if eps is None:
            eps = t.randn_like(self.normal_mean)
        pre = self.normal_mean + self.normal_std * eps
        return t.tanh(pre), pre

    def log_prob(self, pre_tanh: t.Tensor) -> t.Tensor:
        var = self.normal_std ** 2
        normal_lp = (-((pre_tanh - self.normal_mean) ** 2) / (2 * var) - self.normal_std.log()
                     - float(np.log(np.sqrt(2 * np.pi))))
        log_det = 2 * float(np.log(2)) + logsigmoid(2 * pre_tanh) + logsigmoid(-2 * pre_tanh)
        return normal_lp - log_det


class GaussianActor(nn.Module):
    def __init__(
        self,
        state_dim: int,
        action_dim: int,
        hidden_units: tuple[int, ...],
        hidden_activation: nn.Module,
        device: str,
    ):
        super().__init__()
        self.action_dim = action_dim
        self.net = MLP(state_dim, 2 * action_dim, hidden_units, hidden_activation=hidden_activation)
        self.device = device

    def forward(self, obs: t.Tensor, eps: t.Tensor | None = None) -> tuple[t.Tensor, t.Tensor | None]:
        """Train mode: (tanh(mu + sigma*eps), log pi); eval mode: (tanh(mu), None).
        ``eps`` lets a caller inject the N(0,1) draw (parity tests)."""
        raw = self.net(obs)
        mean, log_std = raw[:, :self.action_dim], raw[:, self.action_dim:]
        if not self.training:
            return t.tanh(mean), None
        std = t.exp(log_std.clamp(*LOG_STD_MIN_MAX))
        dist = TanhNormal(mean, std)
        action, pre = dist.rsample(eps)
        return action, dist.log_prob(pre).sum(dim=1, keepdim=True)

    def _act(self, state: npt.NDArray, sample: bool) -> npt.NDArray:
        """One observation -> one action: the MLP on the GPU, the 2A-float head on the host (the
        log-density the training forward also returns is not needed to act, and a handful of
        torch ops on a [1, A] device tensor cost more than the network)."""
        if not self.net.on_gpu():
            s = t.as_tensor(state, dtype=t.float32, device=self.device).unsqueeze(0)
            with t.no_grad():
                was = self.training
                self.train(sample)
                try:
                    return self.forward(s)[0].numpy()[0]
                finally:
                    self.train(was)
        if not sample:
            return self.net.hip_act(state, _capi.ACT_GAUSS_MEAN)
        raw = self.net.hip_act(state)
        mean, log_std = raw[:self.action_dim], raw[self.action_dim:]
        std = np.exp(np.clip(log_std, *LOG_STD_MIN_MAX))
        return np.tanh(mean + std * t.randn(self.action_dim).numpy()).astype(np.float32)

    def explore(self, state: npt.NDArray) -> npt.NDArray:
        # reference semantics (nn_models.py:180-195): a sample in train mode, tanh(mean) in eval mode
        return self._act(state, sample=self.training)

    def exploit(self, state: npt.NDArray) -> npt.NDArray:
        return self._act(state, sample=False)
