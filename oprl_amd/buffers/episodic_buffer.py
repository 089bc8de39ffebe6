"""HBM-resident episodic replay buffer.

API, storage shapes and ring/eviction semantics of the reference
(/root/reference/src/oprl/buffers/episodic_buffer.py:13-140); the work is split
differently:

* index bookkeeping (episode ring pointer, per-episode lengths, live-transition
  count) is host logic and stays in Python ints, exactly as in the reference
  (:81-112), including its quirks: ``episodes_counter`` starts at 1, the slot
  the pointer advances onto is evicted immediately, ``add_episode`` with a
  terminal last row advances twice;
* data movement goes through liboprl_amd.so: ``add_transition`` stages the row
  in pinned host memory (``oprl_replay_write``) and rows reach HBM in one
  batched copy + scatter kernel at the next ``sample``/tensor access;
  ``sample`` is the gather kernel (``oprl_replay_sample``): uniform flat
  indices -> (episode, step) by binary search over cumulative episode ends ->
  one contiguous [s | s'] run + action + reward + done per sample, staged
  through LDS and written coalesced.

Storage is zero-filled instead of ``t.empty`` so the reference's reads of
never-written ``t+1`` slots are at least deterministic.
On ``device='cpu'`` the container and its bookkeeping work (host logic is
testable without a GPU) but ``sample`` raises: there is no CPU sampler."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np
import numpy.typing as npt
import torch as t

from oprl_amd import _capi
from oprl_amd.buffers.protocols import ReplayBufferProtocol

Transition = tuple[npt.NDArray, npt.NDArray, float, bool, npt.NDArray]


@dataclass
class EpisodicReplayBuffer(ReplayBufferProtocol):
    buffer_size_transitions: int
    state_dim: int
    action_dim: int
    gamma: float = 0.99
    max_episode_lenth: int = 1000   # [sic] — the reference's spelling is the API
    episodes_counter: int = 1
    device: str = "cpu"
    seed: int = 0                   # device sampler key (Philox); extension

    _tensors: dict[str, t.Tensor] = field(init=False)
    _max_episodes: int = field(init=False)
    _ep_pointer: int = 0
    _number_transitions = 0
    _created: bool = False

    def create(self) -> "EpisodicReplayBuffer":
        E = self._max_episodes = self.buffer_size_transitions // self.max_episode_lenth
        L, S, A = self.max_episode_lenth, self.state_dim, self.action_dim
        dev = t.device(self.device)
        self._tensors = {
            "actions": t.zeros((E, L, A), dtype=t.float32, device=dev),
            "rewards": t.zeros((E, L, 1), dtype=t.float32, device=dev),
            "dones": t.zeros((E, L, 1), dtype=t.float32, device=dev),
            "states": t.zeros((E, L + 1, S), dtype=t.float32, device=dev),
        }
        self.ep_lens = [0] * E
        self._handle = None
        self._lens_dirty = True
        self._sample_counter = 0
        self._on_gpu = dev.type == "cuda"
        if self._on_gpu:
            self._dev = dev if dev.index is not None else t.device("cuda", t.cuda.current_device())
            lib = _capi.load()
            h = C.c_void_p()
            with _capi.on_device(self._dev):
                _capi.check(lib.oprl_replay_create(
                    E, L, S, A, _capi.ptr(self._tensors["states"]), _capi.ptr(self._tensors["actions"]),
                    _capi.ptr(self._tensors["rewards"]), _capi.ptr(self._tensors["dones"]), C.byref(h)),
                    "oprl_replay_create")
            self._handle = h
            self._lib = lib
        self._created = True
        return self

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            try:
                self._lib.oprl_replay_destroy(h)
            except Exception:
                pass
            self._handle = None

    def check_created(self) -> None:
        if not self._created:
            raise RuntimeError("Replay buffer has to be created with `.create()`.")

    # storage views (flush staged rows first so they are observable) -----------
    def _flush(self) -> None:
        if self._handle is not None:
            with _capi.on_device(self._dev):
                _capi.check(self._lib.oprl_replay_flush(self._handle, _capi.current_stream()),
                            "oprl_replay_flush")

    def _tensor(self, name: str) -> t.Tensor:
        self.check_created()
        self._flush()
        return self._tensors[name]

    @property
    def states(self) -> t.Tensor:
        return self._tensor("states")

    @property
    def actions(self) -> t.Tensor:
        return self._tensor("actions")

    @property
    def rewards(self) -> t.Tensor:
        return self._tensor("rewards")

    @property
    def dones(self) -> t.Tensor:
        return self._tensor("dones")

    # write path -----------------------------------------------------------------
    def add_transition(
        self,
        state: npt.NDArray,
        action: npt.NDArray,
        reward: float,
        done: bool,
        episode_done: bool | None = None,
    ) -> None:
        e, l = self._ep_pointer, self.ep_lens[self._ep_pointer]
        if l >= self.max_episode_lenth:
            raise IndexError(f"episode slot {e} is full ({l} steps, max_episode_lenth={self.max_episode_lenth})")
        s = np.ascontiguousarray(state, dtype=np.float32).reshape(self.state_dim)
        a = np.ascontiguousarray(action, dtype=np.float32).reshape(self.action_dim)  # f64 actions are cast
        eager = self._handle is not None and getattr(self, "eager_flush", False)
        if eager:
            pass        # (one library call below, once the episode table has this step in it)
        elif self._handle is not None:
            # (staging that fills up is flushed by the library: this buffer's device must be current then)
            with _capi.on_device(self._dev):
                _capi.check(self._lib.oprl_replay_write(
                    self._handle, e, l, s.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p),
                    float(reward), float(done)), "oprl_replay_write")
        else:  # host container only (no GPU): plain row stores
            self._tensors["states"][e, l] = t.from_numpy(s)
            self._tensors["actions"][e, l] = t.from_numpy(a)
            self._tensors["rewards"][e, l] = float(reward)
            self._tensors["dones"][e, l] = float(done)
        self.ep_lens[e] += 1
        self._number_transitions = min(self._number_transitions + 1, self.buffer_size_transitions)
        self._touch_len(e)
        if episode_done:
            self._inc_episode()
        if eager:
            # the per-env-step caller (trainers/base_trainer.py): the row and the episode table go to HBM NOW — one
            # library call, one small launch that runs while the host walks on to the update call
            if self._lens_dirty or getattr(self, "_lens_np", None) is None:
                self._lens_np = np.asarray(self.ep_lens, dtype=np.int32).copy()
                self._lens_dirty = False
            with _capi.on_device(self._dev):
                _capi.check(self._lib.oprl_replay_write_flush(
                    self._handle, e, l, s.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), float(reward),
                    float(done), self._lens_np.ctypes.data_as(C.POINTER(C.c_int32)), self.episodes_counter,
                    _capi.current_stream()), "oprl_replay_write_flush")
            self._lens_touched = False

    def add_transitions(self, rows: npt.NDArray, episode_done: bool = False) -> None:
        """``len(rows)`` consecutive transitions of the episode being written, as float32 records
        ``[state (S) | action (A) | reward | done | ...]`` (extra columns ignored) — the bookkeeping of that many
        ``add_transition`` calls and ONE library call for the data (oprl_replay_write_block); with
        ``episode_done`` the episode is closed after the last one.  What the learner ranks of the distributed
        setup use to take in a whole actor episode (extension; the reference adds transitions one by one)."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        n, S, A = len(rows), self.state_dim, self.action_dim
        if rows.ndim != 2 or rows.shape[1] < S + A + 2:
            raise ValueError(f"rows must be [n, >= {S + A + 2}] float32 records")
        e, l = self._ep_pointer, self.ep_lens[self._ep_pointer]
        if l + n > self.max_episode_lenth:
            raise IndexError(f"episode slot {e} would overflow ({l} + {n} steps, max_episode_lenth={self.max_episode_lenth})")
        if n:
            if self._handle is not None:
                with _capi.on_device(self._dev):
                    _capi.check(self._lib.oprl_replay_write_block(
                        self._handle, e, l, n, rows.ctypes.data_as(C.c_void_p), rows.shape[1],
                        _capi.current_stream()), "oprl_replay_write_block")
            else:
                self._tensors["states"][e, l:l + n] = t.from_numpy(rows[:, :S])
                self._tensors["actions"][e, l:l + n] = t.from_numpy(rows[:, S:S + A])
                self._tensors["rewards"][e, l:l + n, 0] = t.from_numpy(rows[:, S + A])
                self._tensors["dones"][e, l:l + n, 0] = t.from_numpy(rows[:, S + A + 1])
            self.ep_lens[e] += n
            self._number_transitions = min(self._number_transitions + n, self.buffer_size_transitions)
            self._touch_len(e)
        if episode_done:
            self._inc_episode()

    def _inc_episode(self) -> None:
        nxt = (self._ep_pointer + 1) % self._max_episodes
        if self.ep_lens[nxt] > 0:
            # the ring wraps onto a slot that holds data: rows staged for that slot and rows about to be
            # staged for it must not meet in one scatter launch (its blocks run in no particular order)
            self._flush()
        self._ep_pointer = (self._ep_pointer + 1) % self._max_episodes
        self.episodes_counter = min(self.episodes_counter + 1, self._max_episodes)
        self._number_transitions -= self.ep_lens[self._ep_pointer]
        self.ep_lens[self._ep_pointer] = 0
        self._touch_len(self._ep_pointer)

    def add_episode(self, episode: list[Transition]) -> None:
        for s, a, r, d, _ in episode:
            self.add_transition(s, a, r, d, episode_done=d)
        self._inc_episode()

    # read path ------------------------------------------------------------------
    def _touch_len(self, e: int) -> None:
        """ep_lens[e] changed through this class: keep the int32 mirror the library reads in step (one element — a
        mirror rebuilt from the list costs O(episodes) of python per env step, 20 us at a thousand episodes)."""
        mirror = getattr(self, "_lens_np", None)
        if mirror is None or len(mirror) != len(self.ep_lens):
            self._lens_dirty = True            # no mirror yet (or ep_lens was replaced): rebuild at the next sync
        else:
            mirror[e] = self.ep_lens[e]
            self._lens_touched = True

    def _sync_lens(self) -> None:
        if self._lens_dirty:                   # set by create / load_state_dict / whoever assigns ep_lens directly
            self._lens_np = np.asarray(self.ep_lens, dtype=np.int32).copy()
            self._lens_dirty = False
            self._lens_touched = True
        if getattr(self, "_lens_touched", False):
            n = self.episodes_counter
            with _capi.on_device(self._dev):
                _capi.check(self._lib.oprl_replay_set_lens(self._handle, self._lens_np.ctypes.data_as(C.POINTER(C.c_int32)), n,
                                                           _capi.current_stream()),
                            "oprl_replay_set_lens")
            self._lens_touched = False

    @property
    def handle(self):
        """C handle with the device-side episode table up to date (for step_n)."""
        self.check_created()
        if self._handle is None:
            raise RuntimeError("replay buffer is not on a GPU")
        self._sync_lens()
        return self._handle

    def sample(self, batch_size: int, inds: t.Tensor | npt.NDArray | None = None,
               return_indices: bool = False):
        """Uniform (with replacement) minibatch of live transitions, including the
        in-progress episode's tail — reference semantics (:123-133).  ``inds``:
        optional injected flat indices (what ``np.random.randint(0, len, B)``
        would return) for parity tests; None draws them on device."""
        self.check_created()
        if self._handle is None:
            raise RuntimeError("sample() runs on the MI355X gather kernel; create the buffer with "
                               "device='cuda' (there is no CPU sampler)")
        if self._number_transitions <= 0:
            raise ValueError("cannot sample from an empty replay buffer")
        self._sync_lens()
        B, S, A, dev = int(batch_size), self.state_dim, self.action_dim, self._dev
        out_s = t.empty((B, S), dtype=t.float32, device=dev)
        out_a = t.empty((B, A), dtype=t.float32, device=dev)
        out_r = t.empty((B, 1), dtype=t.float32, device=dev)
        out_d = t.empty((B, 1), dtype=t.float32, device=dev)
        out_s2 = t.empty((B, S), dtype=t.float32, device=dev)
        idx = None
        if inds is not None:
            idx = t.as_tensor(inds).to(device=dev, dtype=t.int64).contiguous()
            if idx.numel() != B:
                raise ValueError("inds must hold batch_size indices")
        ep = st = None
        if return_indices:
            ep = t.empty(B, dtype=t.int32, device=dev)
            st = t.empty(B, dtype=t.int32, device=dev)
        with _capi.on_device(dev):
            _capi.check(self._lib.oprl_replay_sample(
                self._handle, B, _capi.ptr(idx), self.seed, self._sample_counter, _capi.ptr(out_s),
                _capi.ptr(out_a), _capi.ptr(out_r), _capi.ptr(out_d), _capi.ptr(out_s2),
                _capi.ptr(ep), _capi.ptr(st), _capi.current_stream()), "oprl_replay_sample")
        self._sample_counter += 1
        if return_indices:
            return (out_s, out_a, out_r, out_d, out_s2), (ep, st)
        return out_s, out_a, out_r, out_d, out_s2

    # ---- checkpoint ------------------------------------------------------
    def state_dict(self) -> dict:
        """Storage tensors (host copies) and the bookkeeping that decides where the next
        transition lands and which Philox counter the next sample() uses."""
        self._flush()
        return {
            "dims": (self.buffer_size_transitions, self.max_episode_lenth, self.state_dim, self.action_dim),
            "tensors": {k: v.detach().cpu().clone() for k, v in self._tensors.items()},
            "ep_lens": list(self.ep_lens), "episodes_counter": self.episodes_counter,
            "ep_pointer": self._ep_pointer, "number_transitions": self._number_transitions,
            "sample_counter": self._sample_counter, "seed": self.seed,
        }

    def load_state_dict(self, sd: dict) -> None:
        self.check_created()
        dims = (self.buffer_size_transitions, self.max_episode_lenth, self.state_dim, self.action_dim)
        if tuple(sd["dims"]) != dims:
            raise ValueError(f"replay checkpoint dims {tuple(sd['dims'])} != {dims}")
        self._flush()
        for k, v in sd["tensors"].items():
            self._tensors[k].copy_(v.to(self._tensors[k].device))
        self.ep_lens = list(sd["ep_lens"])
        self.episodes_counter = int(sd["episodes_counter"])
        self._ep_pointer = int(sd["ep_pointer"])
        self._number_transitions = int(sd["number_transitions"])
        self._sample_counter = int(sd["sample_counter"])
        self.seed = int(sd["seed"])
        self._lens_dirty = True

    @property
    def last_episode_length(self) -> int:
        return self.ep_lens[self._ep_pointer]

    def __len__(self) -> int:
        return self._number_transitions
