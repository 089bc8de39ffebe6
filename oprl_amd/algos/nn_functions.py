"""Polyak averaging / gradient freezing helpers
(reference: /root/reference/src/oprl/algos/nn_functions.py:5-16).  On GPU
modules the update is the fused HIP kernel behind ``oprl_polyak``; inside
``update()`` it is not called at all — the Polyak step is fused into the
dW+Adam kernel."""
from __future__ import annotations

import torch as t
import torch.nn as nn

from oprl_amd import _capi
from oprl_amd.algos.nn_models import ensure_flat


def soft_update(target: nn.Module, source: nn.Module, tau: float) -> None:
    """target <- (1 - tau) * target + tau * source, parameter by parameter."""
    tp = next(target.parameters())
    with t.no_grad():
        if tp.is_cuda:
            ta, sa = ensure_flat(target), ensure_flat(source)
            if ta.numel() != sa.numel():
                raise ValueError("soft_update: target and source differ in size")
            with t.cuda.device(tp.device):
                _capi.check(_capi.load().oprl_polyak(_capi.ptr(ta), _capi.ptr(sa), ta.numel(),
                                                     float(tau), _capi.current_stream()), "oprl_polyak")
            for m in target.modules():       # the master changed behind torch's back
                if hasattr(m, "mark_dirty"):
                    m.mark_dirty()
        else:
            for tgt, src in zip(target.parameters(), source.parameters()):
                tgt.data.mul_(1.0 - tau).add_(tau * src.data)


def disable_gradient(network: nn.Module) -> None:
    for param in network.parameters():
        param.requires_grad = False
