"""ctypes binding of include/oprl_amd.h (liboprl_amd.so).

There is NO fallback: if the library is missing or a call fails, a RuntimeError
carrying ``oprl_last_error()`` is raised.  Build it with ``oprl_amd.build.build()``
(or ``python -m oprl_amd.build``), which drives ``hipcc --offload-arch=gfx950``.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

OPRL_ABI_VERSION = 1
OPRL_MAX_LAYERS = 4
OPRL_MAX_CRITICS = 5
ALGO = {"ddpg": 0, "td3": 1, "sac": 2, "tqc": 3}
PRECISION = {"f32": 0, "bf16": 1, "x2": 2}
ACT_NONE, ACT_TANH, ACT_GAUSS_MEAN = 0, 1, 4

LIB_PATH = Path(__file__).resolve().parent / "lib" / "liboprl_amd.so"


class OprlNet(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32),
        ("dims", C.c_int32 * (OPRL_MAX_LAYERS + 1)),
        ("theta", C.c_void_p),
        ("theta_target", C.c_void_p),
        ("adam_m", C.c_void_p),
        ("adam_v", C.c_void_p),
        ("grad", C.c_void_p),
        ("pack", C.c_void_p),
        ("pack_target", C.c_void_p),
    ]


class OprlHparams(C.Structure):
    _fields_ = [
        ("gamma", C.c_double), ("tau", C.c_double),
        ("lr_actor", C.c_double), ("lr_critic", C.c_double), ("lr_alpha", C.c_double),
        ("beta1", C.c_double), ("beta2", C.c_double), ("adam_eps", C.c_double),
        ("policy_noise", C.c_double), ("noise_clip", C.c_double), ("max_action", C.c_double),
        ("alpha_init", C.c_double),
        ("target_entropy", C.c_double),
        ("policy_freq", C.c_int32),
        ("tune_alpha", C.c_int32),
        ("n_quantiles", C.c_int32), ("top_quantiles_to_drop", C.c_int32),
    ]


class OprlLearnerConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("algo", C.c_int32),
        ("precision", C.c_int32),
        ("state_dim", C.c_int32), ("action_dim", C.c_int32),
        ("max_batch", C.c_int32),
        ("n_critics", C.c_int32),
        ("no_fuse", C.c_int32),
        ("export_grads", C.c_int32),
        ("actor", OprlNet),
        ("critics", OprlNet * OPRL_MAX_CRITICS),
        ("log_alpha", C.c_void_p),
        ("log_alpha_m", C.c_void_p),
        ("log_alpha_v", C.c_void_p),
        ("log_alpha_grad", C.c_void_p),
        ("hp", OprlHparams),
    ]


# every symbol the header declares: name -> (restype, argtypes)
_P, _I32, _I64, _U64, _F, _D = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double
SIGNATURES = {
    "oprl_last_error": (C.c_char_p, []),
    "oprl_abi_version": (C.c_int, []),
    "oprl_learner_create": (C.c_int, [C.POINTER(OprlLearnerConfig), C.POINTER(_P)]),
    "oprl_learner_destroy": (C.c_int, [_P]),
    "oprl_learner_update": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _P, _P, _P]),
    "oprl_learner_apply": (C.c_int, [_P, _I32, _D, _P]),
    "oprl_learner_update_phase": (C.c_int, [_P, _I32, _P, _P, _P, _P, _P, _I32, _P, _P, _P]),
    "oprl_learner_step_n": (C.c_int, [_P, _P, _I32, _I32, _U64, _P]),
    "oprl_learner_step_act": (C.c_int, [_P, _P, _I32, _U64, _P, _P]),
    "oprl_learner_act_wait": (C.c_int, [_P, _P, _I32, C.c_int64]),
    "oprl_group_create": (C.c_int, [C.POINTER(_P), _I32, C.POINTER(_P)]),
    "oprl_group_destroy": (C.c_int, [_P]),
    "oprl_group_step_n": (C.c_int, [_P, _P, _I32, _I32, C.POINTER(_U64), _P]),
    "oprl_learner_set_cluster": (C.c_int, [_P, _I32]),
    "oprl_learner_read_scalars": (C.c_int, [_P, C.POINTER(C.c_float), _I32, _P]),
    "oprl_learner_update_count": (C.c_int, [_P, C.POINTER(_I64)]),
    "oprl_learner_set_update_count": (C.c_int, [_P, _I64]),
    "oprl_learner_check": (C.c_int, [_P]),
    "oprl_learner_clear_error": (C.c_int, [_P]),
    "oprl_learner_debug_expire": (C.c_int, [_P, _I32]),
    "oprl_learner_debug_form": (C.c_int, [_P, _I32, _P]),
    "oprl_debug_noise": (C.c_int, [_P, _I32, _U64, _I32, _I32, _P, _P]),
    "oprl_learner_set_seed": (C.c_int, [_P, _U64, _I32]),
    "oprl_learner_debug_ptrs": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P)]),
    "oprl_learner_debug_view": (C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "oprl_net_pack_floats": (_I64, [C.POINTER(OprlNet)]),
    "oprl_net_repack": (C.c_int, [C.POINTER(OprlNet), _I32, _P]),
    "oprl_learner_sync_params": (C.c_int, [_P, _P]),
    "oprl_mlp_forward": (C.c_int, [C.POINTER(OprlNet), _I32, _P, _I32, _P, _I32, _I32, _I32, _P, _P]),
    "oprl_mlp_act": (C.c_int, [C.POINTER(OprlNet), _P, _I32, _I32, _P, _I32, _P]),
    "oprl_mlp_backward": (C.c_int, [C.POINTER(OprlNet), _P, _I32, _P, _I32, _I32, _P, _P, _P]),
    "oprl_adam_step": (C.c_int, [_P, _P, _P, _P, _I64, _I32, _D, _D, _D, _D, _D, _P]),
    "oprl_polyak": (C.c_int, [_P, _P, _I64, _D, _P]),
    "oprl_learner_set_trace": (C.c_int, [_P, _P]),
    "oprl_comm_unique_id": (C.c_int, [C.c_char_p, C.c_char_p]),
    "oprl_comm_init": (C.c_int, [_P, C.c_char_p, _I32, _I32, C.c_char_p]),
    "oprl_comm_broadcast_params": (C.c_int, [_P, _I32, _P]),
    "oprl_p2p_create": (C.c_int, [_P, _I32, _I32, _P]),
    "oprl_p2p_connect": (C.c_int, [_P, _P]),
    "oprl_p2p_selftest": (C.c_int, [_P, _P]),
    "oprl_p2p_enable": (C.c_int, [_P, _I32]),
    "oprl_learner_dp_update": (C.c_int, [_P, _P, _P, _P, _P, _P, _I32, _P, _P, _P]),
    "oprl_learner_dp_step_n": (C.c_int, [_P, _P, _I32, _I32, _U64, _P]),
    "oprl_learner_get_counters": (C.c_int, [_P, C.POINTER(_I64)]),
    "oprl_learner_set_counters": (C.c_int, [_P, C.POINTER(_I64)]),
    "oprl_profile_enable": (C.c_int, [_I32]),
    "oprl_profile_read": (C.c_int, [C.POINTER(_I64), C.POINTER(C.c_double), _I32]),
    "oprl_replay_create": (C.c_int, [_I32, _I32, _I32, _I32, _P, _P, _P, _P, C.POINTER(_P)]),
    "oprl_replay_destroy": (C.c_int, [_P]),
    "oprl_replay_write": (C.c_int, [_P, _I32, _I32, _P, _P, _F, _F]),
    "oprl_replay_write_block": (C.c_int, [_P, _I32, _I32, _I32, _P, _I32, _P]),
    "oprl_replay_flush": (C.c_int, [_P, _P]),
    "oprl_replay_write_flush": (C.c_int, [_P, _I32, _I32, _P, _P, _F, _F, C.POINTER(C.c_int32), _I32, _P]),
    "oprl_replay_set_lens": (C.c_int, [_P, C.POINTER(C.c_int32), _I32, _P]),
    "oprl_replay_sample": (C.c_int, [_P, _I32, _P, _U64, _U64, _P, _P, _P, _P, _P, _P, _P, _P]),
}

_lib = None


def lib_path() -> Path:
    """liboprl_amd.so; with OPRL_AMD_TRACE=1 the build with the in-kernel stage stamps compiled in
    (liboprl_amd_trace.so, `python -m oprl_amd.build --trace`: tools/trace_slice.py)."""
    if os.environ.get("OPRL_AMD_TRACE", "0") not in ("", "0") and "OPRL_AMD_LIB" not in os.environ:
        return LIB_PATH.with_name("liboprl_amd_trace.so")
    return Path(os.environ.get("OPRL_AMD_LIB", str(LIB_PATH)))


def load() -> C.CDLL:
    """Load the library and bind every declared symbol (AttributeError if one is
    missing).  Does not touch the GPU."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not p.exists():
        raise RuntimeError(
            f"{p} not found: the MI355X HIP extension is not built. Run "
            "`python -m oprl_amd.build` (needs hipcc). There is no CPU fallback.")
    # torch first: the process must run on ONE HIP runtime, the one torch bundles (device
    # memory and streams are torch's).  Loading this library before torch pulled in
    # /opt/rocm's libamdhip64 as well, and kernels then failed with "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.oprl_abi_version() != OPRL_ABI_VERSION:
        raise RuntimeError("liboprl_amd.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().oprl_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"liboprl_amd {what} failed (status {rc}): {msg}")


_raw_stream = None
_cur_dev = None


def current_stream():
    """The caller's current HIP stream as a void* (torch.cuda.current_stream() builds a Stream
    object per call, ~10 us; the raw accessor is what torch's own compiled code uses)."""
    global _raw_stream, _cur_dev
    if _raw_stream is None:
        import torch
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
        _cur_dev = torch._C._cuda_getDevice
    if _raw_stream:
        return C.c_void_p(_raw_stream(_cur_dev()))
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(dev):
    """``with on_device(dev):`` — torch.cuda.device(dev) only when dev is not already current
    (entering the real guard costs several microseconds on the per-env-step path)."""
    import torch
    cur = torch._C._cuda_getDevice()
    if dev.index is None or dev.index == cur:
        return _NO_GUARD
    return torch.cuda.device(dev.index)


def ptr(t) -> C.c_void_p:
    """Device pointer of a torch tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())
