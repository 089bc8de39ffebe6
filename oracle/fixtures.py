"""Deterministic inputs shared by the golden generator and the tests.  TEST
INFRASTRUCTURE ONLY (same import rule as oprl_oracle.py).

Everything is drawn from numpy's *legacy* ``RandomState`` (stream frozen by
numpy policy), so a fixture only has to store seeds + expected outputs."""
from __future__ import annotations

import numpy as np
import torch as t

from .oprl_oracle import make_mlp_params

# name -> (state_dim, action_dim, batch)   (dm_control dims)
ENVS = {
    "walker": (24, 6),
    "cheetah": (17, 6),
    "humanoid": (67, 21),
}


def make_batch(seed: int, B: int, S: int, A: int, p_done: float = 0.01):
    rs = np.random.RandomState(seed)
    s = rs.standard_normal((B, S)).astype(np.float32)
    a = rs.uniform(-1, 1, (B, A)).astype(np.float32)
    r = rs.uniform(0, 1, (B, 1)).astype(np.float32)
    d = (rs.uniform(0, 1, (B, 1)) < p_done).astype(np.float32)
    d[min(3, B - 1), 0] = 1.0  # always exercise the (1-d) branch
    s2 = rs.standard_normal((B, S)).astype(np.float32)
    return tuple(t.from_numpy(x) for x in (s, a, r, d, s2))


def make_noise(seed: int, shape) -> t.Tensor:
    return t.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


def actor_dims(S, A, gaussian=False, hidden=(256, 256)):
    return [S, *hidden, 2 * A if gaussian else A]


def critic_dims(S, A, out=1, hidden=(256, 256)):
    return [S + A, *hidden, out]


def make_net(seed: int, dims):
    return make_mlp_params(np.random.RandomState(seed), dims)


def digest(x: t.Tensor, n: int = 256) -> dict:
    """Small summary of a tensor: strided sample + sums (float64)."""
    f = x.detach().to(t.float64).reshape(-1)
    stride = max(1, f.numel() // n)
    return dict(sample=f[::stride].numpy().astype(np.float64), sum=float(f.sum()),
                abssum=float(f.abs().sum()), numel=int(f.numel()))


def digest_list(xs, n: int = 256) -> dict:
    out = {}
    for i, x in enumerate(xs):
        dg = digest(x, n)
        out[f"{i}.sample"] = dg["sample"]
        out[f"{i}.stats"] = np.array([dg["sum"], dg["abssum"], dg["numel"]], np.float64)
    return out
