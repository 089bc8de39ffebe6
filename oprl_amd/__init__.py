"""oprl_amd — MI355X-native off-policy learner with the oprl API surface.

The hot path (replay sample -> TD target -> critic/actor forward-backward ->
Adam -> Polyak) runs in hand-written gfx950 HIP kernels behind the C-ABI of
include/oprl_amd.h; this package is the Python host side that mirrors the
reference's Protocols (AlgorithmProtocol, ReplayBufferProtocol)."""
__version__ = "0.1.0"
