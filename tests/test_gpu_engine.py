"""GPU parity of the HIP building blocks (through the C-ABI) against the CPU
oracle: whole-MLP forward, backward (dW, db, dX), Adam, Polyak.  fp32 MFMA vs
torch-CPU fp32: only summation order differs, gate 2e-5 (measured ~1e-6)."""
import ctypes as C

import numpy as np
import pytest
import torch as t

from oracle import fixtures as fx
from oracle import oprl_oracle as orc
from tests.scenarios import rel_dev

pytestmark = pytest.mark.gpu
TOL = 2e-5

NETS = [  # dims, batch
    ([24, 256, 256, 6], 256),          # walker actor
    ([30, 256, 256, 1], 256),          # walker critic
    ([30, 256, 256, 1], 8),            # reference test batch
    ([17, 256, 256, 6], 100),          # cheetah actor, ragged batch
    ([88, 256, 256, 1], 1024),         # humanoid critic
    ([67, 256, 256, 42], 1000),        # humanoid gaussian actor (3 narrow tiles), ragged
    ([30, 512, 512, 512, 25], 256),    # TQC quantile critic
    ([5, 256, 3], 33),                 # single hidden layer
]


def _mods():
    from oprl_amd import _capi
    from oprl_amd.algos import nn_models
    return _capi, nn_models


def _mk(dims, seed):
    _capi, nnm = _mods()
    import torch.nn as nn
    p = fx.make_net(seed, dims)
    mlp = nnm.MLP(dims[0], dims[-1], tuple(dims[1:-1]), hidden_activation=nn.ReLU()).cuda()
    nnm.flatten_module_(mlp)
    with t.no_grad():
        for dst, src in zip(mlp.parameters(), p):
            dst.copy_(src)
    return p, mlp


@pytest.mark.parametrize("dims,B", NETS)
def test_mlp_forward_matches_oracle(dims, B):
    p, mlp = _mk(dims, 7)
    x = fx.make_noise(11, (B, dims[0]))
    want = orc.mlp_forward(p, x)[-1]
    got = mlp(x.cuda()).cpu()
    assert got.shape == want.shape
    assert rel_dev(got.numpy(), want.numpy()) < TOL


def test_forward_two_pointer_concat_and_tanh():
    _capi, nnm = _mods()
    S, A = 24, 6
    p, mlp = _mk([S + A, 256, 256, 1], 3)
    s, a = fx.make_noise(1, (77, S)), fx.make_noise(2, (77, A))
    want = orc.q_forward(p, s, a)[-1]
    got = mlp.hip_forward(s.cuda(), a.cuda()).cpu()
    assert rel_dev(got.numpy(), want.numpy()) < TOL
    p2, pol = _mk([S, 256, 256, A], 4)
    got = pol.hip_forward(s.cuda(), out_act=_capi.ACT_TANH).cpu()
    assert rel_dev(got.numpy(), t.tanh(orc.mlp_forward(p2, s)[-1]).numpy()) < TOL


@pytest.mark.parametrize("dims,B", NETS)
def test_mlp_backward_matches_oracle(dims, B):
    _capi, nnm = _mods()
    lib = _capi.load()
    p, mlp = _mk(dims, 9)
    x = fx.make_noise(13, (B, dims[0]))
    dout = fx.make_noise(17, (B, dims[-1])) / B
    acts = orc.mlp_forward(p, x)
    want_dx = dims[0] <= 48
    grads, dx = orc.mlp_backward(p, acts, dout, need_dx=want_dx)
    arena = mlp._oprl_arena
    grad = t.zeros_like(arena)
    desc = nnm._net_desc(dims, arena.data_ptr(), grad_ptr=grad.data_ptr(),
                         pack_ptr=mlp.ensure_packed().data_ptr())
    xg, dg = x.cuda(), dout.cuda().contiguous()
    dxg = t.zeros((B, dims[0]), device="cuda") if want_dx else None
    _capi.check(lib.oprl_mlp_backward(C.byref(desc), _capi.ptr(xg), dims[0], None, 0, B, _capi.ptr(dg),
                                      _capi.ptr(dxg), _capi.current_stream()), "oprl_mlp_backward")
    t.cuda.synchronize()
    off = 0
    for i, gw in enumerate(grads):
        n = gw.numel()
        got = grad[off:off + n].view(gw.shape).cpu()
        assert rel_dev(got.numpy(), gw.numpy()) < TOL, f"grad tensor {i}"
        off += n
    if want_dx:
        assert rel_dev(dxg.cpu().numpy(), dx.numpy()) < TOL


def test_adam_and_polyak_match_oracle():
    _capi, _ = _mods()
    lib = _capi.load()
    rs = np.random.RandomState(5)
    n = 100_003
    th = t.from_numpy(rs.standard_normal(n).astype(np.float32))
    tgt = t.from_numpy(rs.standard_normal(n).astype(np.float32))
    opt = orc.Adam(3e-4)
    ref = [th.clone()]
    thg, mg, vg = th.cuda(), t.zeros(n, device="cuda"), t.zeros(n, device="cuda")
    for step in range(1, 6):
        gr = t.from_numpy((rs.standard_normal(n) * 10 ** rs.uniform(-6, 0, n)).astype(np.float32))
        opt.step(ref, [gr])
        grg = gr.cuda()
        _capi.check(lib.oprl_adam_step(_capi.ptr(thg), _capi.ptr(mg), _capi.ptr(vg), _capi.ptr(grg), n,
                                       step, 3e-4, 0.9, 0.999, 1e-8, 1.0, _capi.current_stream()))
    t.cuda.synchronize()
    # 1 ulp of fp32 at |theta| ~ 4 is 4.8e-7: fma contraction may differ from ATen by an ulp
    assert (thg.cpu() - ref[0]).abs().max().item() < 5e-7
    assert rel_dev(mg.cpu().numpy(), opt.m[0].numpy()) < 1e-6
    assert rel_dev(vg.cpu().numpy(), opt.v[0].numpy()) < 1e-6
    tg = tgt.cuda()
    _capi.check(lib.oprl_polyak(_capi.ptr(tg), _capi.ptr(thg), n, 5e-3, _capi.current_stream()))
    want = [tgt.clone()]
    orc.polyak(want, [thg.cpu()], 5e-3)
    assert (tg.cpu() - want[0]).abs().max().item() < 5e-7


def test_errors_are_loud():
    _capi, nnm = _mods()
    import torch.nn as nn
    mlp = nnm.MLP(10, 2, (128, 128), hidden_activation=nn.ReLU()).cuda()
    with pytest.raises(RuntimeError, match="hidden width"):
        mlp(t.zeros(4, 10, device="cuda"))
