"""The fused DDPG path (csrc/fused_ddpg.hip: 2 slice kernels, in-kernel gather, three
concurrent roles with granule hand-off, each role a tensor-parallel cluster of CUs —
csrc/tp3.h) against the generic per-net launch sequence.  With clusters the output
layer and the first-layer gradient are sums of per-member partials, so fp32 rounding
differs from the single-CU order: compared at 1e-5 relative after 4 updates (the 1e-4
gate against the reference is checked by test_gpu_algos.py, which runs this path)."""
import pytest
import torch as t

from oracle import fixtures as fx

pytestmark = pytest.mark.gpu


def _ddpg(state_dim=24, action_dim=6, **kw):
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    t.manual_seed(0)
    return DDPG(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=state_dim, action_dim=action_dim,
                device="cuda", **kw).create()


def _close(a, b, tol=1e-5):
    return (a - b).abs().max().item() <= tol * max(b.abs().max().item(), 1e-12)


# cluster 8: the default — clusters of 4, of EIGHT for role A and phase 2's critic pass (tp4_forward /
# tp4_scalar_fb <P, 8>, narrow exchanges); 4: clusters of four only; "4g": clusters of 4 through the generic
# tp3.h passes instead of the lean tp4.h ones
# (8 and 4 run the critic's dW + Adam tiles inside phase 1's launch, csrc/dw_body.h GATED; "8s": role A on eight CUs and the
# dW launches on their own, OPRL_AMD_FORM=plain)
# 8 and 4 also run the ACTOR's dW + Adam tiles inside phase 2's launch, its backward as role U with unit seeds (GATE == 2);
# "8p": phase 2 runs the actor's backward itself and the actor's dW is a launch of its own, OPRL_AMD_FORM=p2)
@pytest.mark.parametrize("cluster", [8, "8s", "8p", 4, "4g", 2, 1])
@pytest.mark.parametrize("B", [256, 8, 100])
def test_fused_equals_generic(B, cluster, monkeypatch):
    monkeypatch.setenv("OPRL_AMD_NO_LEAN", "1" if cluster == "4g" else "0")
    monkeypatch.setenv("OPRL_AMD_NO_WIDE", "0" if cluster in (8, "8s", "8p") else "1")
    monkeypatch.setenv("OPRL_AMD_FORM", "plain" if cluster == "8s" else ("p2" if cluster == "8p" else "chain"))
    cluster = 4 if cluster in ("4g", 8, "8s", "8p") else cluster
    monkeypatch.setenv("OPRL_AMD_CLUSTER", str(cluster))
    fused, generic = _ddpg(), _ddpg(no_fuse=True)
    for step in range(4):
        batch = [x.cuda() for x in fx.make_batch(70 + step, B, 24, 6)]
        fused.update(*batch)
        generic.update(*batch)
    t.cuda.synchronize()
    assert t.isfinite(fused.actor._oprl_arena).all()     # a timed-out exchange would surface as NaN
    # parameters: 1e-4 (Adam's m/sqrt(v) amplifies summation-order noise on elements whose
    # minibatch gradient nearly cancels — tests/scenarios.py::compare); outputs below: 1e-5
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m
    # Adam's first moments: 1e-4 of the largest — for all but a thousandth of the elements, which may be off by up to 1e-3:
    # two summation orders can leave a hidden pre-activation on either side of zero, and that ONE ReLU flip moves the
    # gradient of that unit's row of weights (measured with these batches: 36 of 73734 elements above 1e-4, the largest
    # 2.0e-4, in three of the 21 cases; every other case 3e-7 and below — tools/probe_fused_ratio.py)
    for which in ("actor_m", "critic_m"):
        a, b = getattr(fused.learner, which), getattr(generic.learner, which)
        d = (a - b).abs() / max(b.abs().max().item(), 1e-12)
        assert int((d > 1e-4).sum()) <= d.numel() // 1000 and float(d.max()) <= 1e-3, (which, int((d > 1e-4).sum()), float(d.max()))
    qf, yf = fused.learner.debug_q_y(B)
    qg, yg = generic.learner.debug_q_y(B)
    assert _close(qf, qg) and _close(yf, yg)
    sf, sg = fused.learner.read_scalars(), generic.learner.read_scalars()
    for k in sf:
        assert abs(sf[k] - sg[k]) <= 1e-5 * max(abs(sg[k]), 1e-12), k


@pytest.mark.parametrize("algo", ["ddpg", "sac"])
@pytest.mark.parametrize("B", [256, 100])
def test_generic_cluster_launches_equal_single_cu_launches(B, algo, monkeypatch):
    """The GENERIC launch sequence (no_fuse: one launch per net pass) on tensor-parallel clusters of four (csrc/slice_tp.hip,
    tp3.h / tp4.h through k_slice_tp) against the same sequence on one compute unit per slice (k_mlp_slice): the cluster
    path — exchanges, partial dz1 buffers, XCD-local publishes — has a witness of its own outside the fused kernels."""
    def make(cluster):
        monkeypatch.setenv("OPRL_AMD_CLUSTER", str(cluster))
        if algo == "ddpg":
            return _ddpg(max_batch=B, no_fuse=True)
        return _sac(max_batch=B, no_fuse=True, tune_alpha=True)
    tp, one = make(4), make(1)
    for step in range(4):
        batch = [x.cuda() for x in fx.make_batch(170 + step, B, 24, 6)]
        tp.update(*batch)
        one.update(*batch)
    t.cuda.synchronize()
    tp.learner.check()
    one.learner.check()
    nets = ("actor", "critic", "actor_target", "critic_target") if algo == "ddpg" else ("actor", "critic", "critic_target")
    for m in nets:
        a, b = getattr(tp, m)._oprl_arena, getattr(one, m)._oprl_arena
        assert t.isfinite(a).all() and _close(a, b, 1e-4), m


@pytest.mark.parametrize("B", [512, 1024, 4096])
def test_fused_large_batches(B):
    """Clusters of 4 (the lean passes) are kept while ONE role's clusters fit the chip,
    4 x ceil(B/16) <= CUs: at B=512 / 1024 phase 1 has 384 / 768 workgroups for 256 CUs and the later
    roles start as earlier workgroups retire (block order = A, B, C; a B workgroup only ever waits for
    an A workgroup dispatched before it).  B=4096 = the default max_batch: 1 CU per slice, generic
    passes, 768 workgroups."""
    fused, generic = _ddpg(max_batch=B), _ddpg(max_batch=B, no_fuse=True)
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(75 + step, B, 24, 6)]
        fused.update(*batch)
        generic.update(*batch)
    t.cuda.synchronize()
    assert t.isfinite(fused.critic._oprl_arena).all()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m


@pytest.mark.parametrize("S,A", [(67, 21), (75, 21), (5, 1)])
def test_fused_wide_and_narrow_dims(S, A):
    """Humanoid dims: 16 x 67 state elements per slice are more than one element per thread of the
    lean phase-2 prologue; S + A = 96 is the widest layer-0 input the lean passes take; pendulum-
    sized dims at the other end."""
    B = 256
    fused, generic = _ddpg(S, A), _ddpg(S, A, no_fuse=True)
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(40 + step, B, S, A)]
        fused.update(*batch)
        generic.update(*batch)
    t.cuda.synchronize()
    assert t.isfinite(fused.actor._oprl_arena).all()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m


def test_fused_step_n_equals_generic_step_n():
    from tests.test_gpu_callers import _filled_buffer
    fused, generic = _ddpg(max_batch=64), _ddpg(max_batch=64, no_fuse=True)
    buf = _filled_buffer()
    fused.learner.step_n(buf.handle, 10, 64, seed=11)
    generic.learner.step_n(buf.handle, 10, 64, seed=11)
    t.cuda.synchronize()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m


# ---- TD3: the same two kernels with twin critics (roles A | B1 | B2 | C) -------------------------
def _td3(state_dim=17, **kw):
    from oprl_amd.algos.td3 import TD3
    from oprl_amd.logging import NullLogger
    t.manual_seed(0)
    return TD3(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=state_dim, action_dim=6, device="cuda", **kw).create()


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("inject", [True, False])
def test_fused_td3_equals_generic(inject, split, monkeypatch):
    """6 updates = 3 critic-only + 3 actor steps (policy_freq 2); target-policy smoothing noise
    injected (as the golden tests do) or drawn on device (same Philox stream in both paths); the twin
    target critics side by side (role A and the role-C cluster) or back to back in role A."""
    monkeypatch.setenv("OPRL_AMD_NO_SIDE_BY_SIDE", "0" if split else "1")
    fused, generic = _td3(), _td3(no_fuse=True)
    for step in range(6):
        batch = [x.cuda() for x in fx.make_batch(90 + step, 256, 17, 6)]
        noise = fx.make_noise(190 + step, (256, 6)).cuda() if inject else None
        fused.update(*batch, noise=noise)
        generic.update(*batch, noise=noise)
    t.cuda.synchronize()
    assert t.isfinite(fused.actor._oprl_arena).all() and t.isfinite(fused.critic._oprl_arena).all()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m
    q_f, y_f = fused.learner.debug_q_y(256)
    q_g, y_g = generic.learner.debug_q_y(256)
    assert _close(q_f, q_g, 2e-5) and _close(y_f, y_g, 2e-5)
    sf, sg = fused.learner.read_scalars(), generic.learner.read_scalars()
    for k in ("critic_loss", "q_mean", "q_target_mean"):
        assert abs(sf[k] - sg[k]) <= 1e-4 * max(abs(sg[k]), 1e-6), k


@pytest.mark.parametrize("B", [512, 1024])
def test_fused_td3_oversubscribed_grid(B):
    """TD3 at B=512 / 1024: phase 1 is 4 roles x 4 CUs x 32 / 64 slices = 512 / 1024 workgroups."""
    fused, generic = _td3(max_batch=B), _td3(max_batch=B, no_fuse=True)
    for step in range(4):
        batch = [x.cuda() for x in fx.make_batch(60 + step, B, 17, 6)]
        fused.update(*batch)
        generic.update(*batch)
    t.cuda.synchronize()
    assert t.isfinite(fused.critic._oprl_arena).all()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m


def test_fused_td3_falls_back_without_lean_passes(monkeypatch):
    """TD3's fused kernels exist in the lean (tp4.h) form only: with the lean passes switched off the
    learner must run the generic launch sequence, i.e. be bit-identical to a no_fuse learner."""
    monkeypatch.setenv("OPRL_AMD_NO_LEAN", "1")
    fused, generic = _td3(), _td3(no_fuse=True)
    for step in range(2):
        batch = [x.cuda() for x in fx.make_batch(95 + step, 256, 17, 6)]
        noise = fx.make_noise(195 + step, (256, 6)).cuda()
        fused.update(*batch, noise=noise)
        generic.update(*batch, noise=noise)
    t.cuda.synchronize()
    for m in ("actor", "critic"):
        assert t.equal(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena), m


@pytest.mark.parametrize("B", [100, 8])
def test_fused_td3_ragged_batches(B):
    fused, generic = _td3(), _td3(no_fuse=True)
    for step in range(4):
        batch = [x.cuda() for x in fx.make_batch(97 + step, B, 17, 6)]
        noise = fx.make_noise(197 + step, (B, 6)).cuda()
        fused.update(*batch, noise=noise)
        generic.update(*batch, noise=noise)
    t.cuda.synchronize()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m


def test_fused_td3_step_n_equals_generic_step_n():
    from tests.test_gpu_callers import _filled_buffer
    fused, generic = _td3(24, max_batch=256), _td3(24, max_batch=256, no_fuse=True)
    buf = _filled_buffer()
    fused.learner.step_n(buf.handle, 6, 256, seed=11)
    generic.learner.step_n(buf.handle, 6, 256, seed=11)
    t.cuda.synchronize()
    for m in ("actor", "critic", "actor_target", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m


@pytest.mark.parametrize("algo", ["ddpg", "td3"])
def test_fused_runs_are_deterministic(algo):
    """The cluster exchanges sum in member order and the hand-offs are data-tagged, so two runs of the
    same update stream agree bit for bit (a timed-out exchange would also show up here as NaN)."""
    from tests.test_gpu_callers import _filled_buffer
    outs = []
    for _ in range(2):
        a = _ddpg(max_batch=256) if algo == "ddpg" else _td3(24, max_batch=256)
        buf = _filled_buffer()
        for _ in range(4):
            a.learner.step_n(buf.handle, 500, 256, seed=21)
        t.cuda.synchronize()
        assert t.isfinite(a.actor._oprl_arena).all() and t.isfinite(a.critic._oprl_arena).all()
        outs.append((a.actor._oprl_arena.clone(), a.critic._oprl_arena.clone()))
    assert t.equal(outs[0][0], outs[1][0]) and t.equal(outs[0][1], outs[1][1])


# ---- SAC: the same two kernels with the tanh-Gaussian head (roles A | B1 | B2 | C) -----------------
def _sac(state_dim=24, action_dim=6, **kw):
    from oprl_amd.algos.sac import SAC
    from oprl_amd.logging import NullLogger
    t.manual_seed(0)
    return SAC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=state_dim, action_dim=action_dim,
               device="cuda", **kw).create()


@pytest.mark.parametrize("pair", [True, False])
@pytest.mark.parametrize("tune_alpha", [False, True])
@pytest.mark.parametrize("inject", [True, False])
def test_fused_sac_equals_generic(inject, tune_alpha, pair, monkeypatch):
    """Both Normal(0,1) draws injected (as the golden tests do) or drawn on device (same Philox streams
    in both paths); fixed or learned temperature; phase 2's twin critics on two clusters side by side
    or back to back on one."""
    monkeypatch.setenv("OPRL_AMD_NO_SIDE_BY_SIDE", "0" if pair else "1")   # (phase 2's critic pair and phase 1's twin targets)
    B, S, A = 256, 24, 6
    fused, generic = _sac(tune_alpha=tune_alpha), _sac(tune_alpha=tune_alpha, no_fuse=True)
    for step in range(5):
        batch = [x.cuda() for x in fx.make_batch(300 + step, B, S, A)]
        noise = (fx.make_noise(400 + step, (B, A)).cuda(), fx.make_noise(500 + step, (B, A)).cuda()) if inject else None
        fused.update(*batch, noise=noise)
        generic.update(*batch, noise=noise)
    t.cuda.synchronize()
    assert t.isfinite(fused.actor._oprl_arena).all() and t.isfinite(fused.critic._oprl_arena).all()
    for m in ("actor", "critic", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m
    q_f, y_f = fused.learner.debug_q_y(B)
    q_g, y_g = generic.learner.debug_q_y(B)
    assert _close(q_f, q_g, 2e-5) and _close(y_f, y_g, 2e-5)
    sf, sg = fused.learner.read_scalars(), generic.learner.read_scalars()
    for k in ("critic_loss", "q_mean", "q_target_mean", "actor_loss", "alpha"):
        assert abs(sf[k] - sg[k]) <= 1e-4 * max(abs(sg[k]), 1e-6), k
    if tune_alpha:
        assert abs(fused.alpha - generic.alpha) <= 1e-6 * generic.alpha


@pytest.mark.parametrize("B,S,A", [(1024, 67, 21), (100, 24, 6), (8, 17, 6)])
def test_fused_sac_shapes(B, S, A):
    """Humanoid dims at the reference's batch (over-subscribed grid: 4 roles x 4 x 64 workgroups),
    ragged and tiny batches."""
    fused, generic = _sac(S, A, max_batch=B, tune_alpha=True), _sac(S, A, max_batch=B, tune_alpha=True, no_fuse=True)
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(310 + step, B, S, A)]
        fused.update(*batch)
        generic.update(*batch)
    t.cuda.synchronize()
    assert t.isfinite(fused.actor._oprl_arena).all() and t.isfinite(fused.critic._oprl_arena).all()
    for m in ("actor", "critic", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m


def test_fused_sac_step_n_equals_generic_step_n():
    from tests.test_gpu_callers import _filled_buffer
    fused, generic = _sac(max_batch=64, tune_alpha=True), _sac(max_batch=64, tune_alpha=True, no_fuse=True)
    buf = _filled_buffer()
    fused.learner.step_n(buf.handle, 10, 64, seed=11)
    generic.learner.step_n(buf.handle, 10, 64, seed=11)
    t.cuda.synchronize()
    for m in ("actor", "critic", "critic_target"):
        assert _close(getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena, 1e-4), m
    assert abs(fused.alpha - generic.alpha) <= 1e-6 * generic.alpha


# ---- TQC: the five quantile critics of a phase in one launch ----------------------------------------
@pytest.mark.parametrize("B", [256, 100])
def test_tqc_layerwise_equals_slice_kernel(B, monkeypatch):
    """csrc/layerwise.hip (one launch per layer, workgroup per slice x 64 columns x net) against the
    single-CU slice kernel: same packs and GEMM routine, contraction split over four waves."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger

    def make():
        t.manual_seed(0)
        return TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=256).create()

    lw = make()
    monkeypatch.setenv("OPRL_AMD_NO_LAYERWISE", "1")
    ref = make()
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(25 + step, B, 24, 6)]
        lw.update(*batch)
        ref.update(*batch)
    t.cuda.synchronize()
    assert t.isfinite(lw.critic._oprl_arena).all() and t.isfinite(lw.actor._oprl_arena).all()
    # Adam's first steps move a parameter by ~lr whatever the size of its gradient, so the few
    # elements whose minibatch gradient is pure rounding noise (more of them at B=100) may step the
    # other way in the two summation orders: bound their number and their distance, hold the rest to
    # 1e-5 (both paths sit within 2e-6 of the oracle on z, pi and the parameter digests at B=100)
    for m in ("actor", "critic", "critic_target"):
        a, b = getattr(lw, m)._oprl_arena, getattr(ref, m)._oprl_arena
        d = (a - b).abs()
        scale = b.abs().max().item()
        assert d.max().item() <= 3 * 3 * 3e-4, m                 # 3 steps x 2 lr apart at most (+ margin)
        assert (d > 1e-4 * scale).float().mean().item() < 2e-3, m     # the 1e-4 gate, but for the sign-flip elements
        assert (d > 1e-5 * scale).float().mean().item() < 3e-2, m     # (eps-regime elements: |g| ~ Adam's eps)
    sl, sr = lw.learner.read_scalars(), ref.learner.read_scalars()
    for k in ("critic_loss", "actor_loss", "alpha"):
        assert abs(sl[k] - sr[k]) <= 1e-4 * max(abs(sr[k]), 1e-6), k


@pytest.mark.parametrize("B", [256, 100])
def test_tqc_target_on_head_launch_equals_target_launch(B, monkeypatch):
    """The TD target as the tail of the target critics' head launch (the last head workgroup of a slice to arrive
    sorts the slice's rows; csrc/layerwise.hip lw_tqc_target) against k_tqc_target as a launch of its own: the same
    sorting network on the same values — bit-identical, at a full and a ragged batch."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger

    def make():
        t.manual_seed(0)
        return TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=256).create()

    ride = make()
    monkeypatch.setenv("OPRL_AMD_NO_RIDE", "1")      # (read when the learner is created)
    ref = make()
    for step in range(4):
        batch = [x.cuda() for x in fx.make_batch(40 + step, B, 24, 6)]
        ride.update(*batch)
        ref.update(*batch)
    t.cuda.synchronize()
    assert t.isfinite(ride.critic._oprl_arena).all()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(ride, m)._oprl_arena, getattr(ref, m)._oprl_arena), m


@pytest.mark.parametrize("B", [256, 100])
def test_tqc_actor_forward_riding_on_heads_equals_own_launch(B, monkeypatch):
    """TQC's actor forward on s as riding workgroups of the critic step's head launch (k_lw_head's `R`; the same
    slice_tp_body on the same inputs, counter and noise) against the forward as a launch of the actor phase:
    bit-identical, through update() and through step_n."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer

    def make():
        t.manual_seed(0)
        return TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=256).create()

    ride = make()
    monkeypatch.setenv("OPRL_AMD_NO_RIDE", "2")       # (read when the learner is created)
    own = make()
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(60 + step, B, 24, 6)]
        ride.update(*batch)
        own.update(*batch)
    buf = _filled_buffer()
    ride.learner.step_n(buf.handle, 12, 64, seed=5)
    own.learner.step_n(buf.handle, 12, 64, seed=5)
    t.cuda.synchronize()
    ride.learner.check()
    assert t.isfinite(ride.critic._oprl_arena).all()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(ride, m)._oprl_arena, getattr(own, m)._oprl_arena), m
    sr, so = ride.learner.read_scalars(), own.learner.read_scalars()
    for k in ("critic_loss", "actor_loss", "alpha"):
        assert sr[k] == so[k], k


@pytest.mark.parametrize("B,prec", [(256, "f32"), (100, "f32"), (256, "bf16")])
def test_tqc_early_first_launch_equals_in_place(B, prec, monkeypatch):
    """TQC's online critics' first hidden launch riding behind the actor's forward on s' (k_slice_tp_fin) against the
    same launch in its place at the head of step 3: the same workgroup code on the same inputs, the target pass on
    scratch activations instead of the nets' dW buffers — bit-identical, through update() and through step_n."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer

    def make():
        t.manual_seed(0)
        return TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=256,
                   precision=prec).create()

    early = make()
    monkeypatch.setenv("OPRL_AMD_NO_RIDE", "4")      # (read when the learner is created)
    inplace = make()
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(90 + step, B, 24, 6)]
        early.update(*batch)
        inplace.update(*batch)
    buf = _filled_buffer()
    early.learner.step_n(buf.handle, 12, 64, seed=9)
    inplace.learner.step_n(buf.handle, 12, 64, seed=9)
    t.cuda.synchronize()
    early.learner.check()
    assert t.isfinite(early.critic._oprl_arena).all()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(early, m)._oprl_arena, getattr(inplace, m)._oprl_arena), m
    se, si = early.learner.read_scalars(), inplace.learner.read_scalars()
    for k in ("critic_loss", "actor_loss", "alpha"):
        assert se[k] == si[k], k


@pytest.mark.parametrize("B,prec", [(256, "f32"), (100, "x2"), (256, "x2"), (256, "bf16"), (40, "bf16")])
def test_tqc_hidden_layer_pairs_equal_separate_launches(B, prec, monkeypatch):
    """TQC's two hidden layers per direction as ONE launch (k_lw_mid_pair: the second layer's workgroups wait for the
    flags of the first layer's, rows handed over written through) against a launch per layer: the same workgroup
    arithmetic on the same rows — bit-identical, through update() at full and ragged batches and through step_n."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer

    def make():
        t.manual_seed(0)
        return TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=256,
                   precision=prec).create()

    paired = make()
    monkeypatch.setenv("OPRL_AMD_NO_RIDE", "32")          # (read when the learner is created)
    single = make()
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(90 + step, B, 24, 6)]
        paired.update(*batch)
        single.update(*batch)
    buf = _filled_buffer()
    paired.learner.step_n(buf.handle, 10, 64, seed=9)
    single.learner.step_n(buf.handle, 10, 64, seed=9)
    t.cuda.synchronize()
    paired.learner.check()
    assert t.isfinite(paired.critic._oprl_arena).all()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(paired, m)._oprl_arena, getattr(single, m)._oprl_arena), m
    sp, ss = paired.learner.read_scalars(), single.learner.read_scalars()
    for k in ("critic_loss", "actor_loss", "alpha"):
        assert sp[k] == ss[k], k


@pytest.mark.parametrize("B,prec", [(256, "f32"), (100, "x2"), (256, "x2"), (256, "bf16"), (40, "bf16")])
def test_tqc_second_hidden_layer_riding_on_the_target_heads_equals_its_own_launch(B, prec, monkeypatch):
    """TQC's online critics' second hidden layer on (s, a) as riders of the target pass's head launch (behind the tail of
    the early first launch, whose workgroups publish their rows with a flag per run: k_lw_head / LwFinTail::z1) against
    the k_lw_mid_run2 launch it replaces: the same workgroup arithmetic on the same rows — bit-identical, through
    update() at full and ragged batches and through step_n."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer

    def make():
        t.manual_seed(0)
        return TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=256,
                   precision=prec).create()

    riding = make()
    monkeypatch.setenv("OPRL_AMD_NO_RIDE", "128")         # (read when the learner is created)
    own = make()
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(90 + step, B, 24, 6)]
        riding.update(*batch)
        own.update(*batch)
    buf = _filled_buffer()
    riding.learner.step_n(buf.handle, 10, 64, seed=9)
    own.learner.step_n(buf.handle, 10, 64, seed=9)
    t.cuda.synchronize()
    riding.learner.check()
    assert t.isfinite(riding.critic._oprl_arena).all()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(riding, m)._oprl_arena, getattr(own, m)._oprl_arena), m
    sr, so = riding.learner.read_scalars(), own.learner.read_scalars()
    for k in ("critic_loss", "actor_loss", "alpha"):
        assert sr[k] == so[k], k


@pytest.mark.parametrize("bits", ["256", "512"])
@pytest.mark.parametrize("B,prec", [(256, "f32"), (100, "x2"), (256, "x2"), (256, "bf16"), (40, "bf16")])
def test_tqc_actor_backward_riding_on_the_action_gradient_launch_equals_its_own_launch(B, prec, bits, monkeypatch):
    """TQC's actor backward (k_mlp_slice_tp, tanh-Gaussian seed from the five critics' action gradients) as riders of the
    k_lw_dact launch that PRODUCES those gradients (the dact workgroups write their rows through and raise a flag per
    (net, slice); the riders request their fragments, then wait: SeedArgs::da_flags, r06-16) against the launch it
    replaces: the same sums in the same order — bit-identical, through update() at full and ragged batches and step_n.
    bits = 512: the backward rides, but its dW + Adam tiles and the temperature's step — gated on the riding members' flags
    behind it (r06-18) — are the launch of their own they were."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer

    def make():
        t.manual_seed(0)
        return TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=256,
                   precision=prec).create()

    riding = make()
    monkeypatch.setenv("OPRL_AMD_NO_RIDE", bits)          # (read when the learner is created)
    own = make()
    for step in range(3):
        batch = [x.cuda() for x in fx.make_batch(90 + step, B, 24, 6)]
        riding.update(*batch)
        own.update(*batch)
    buf = _filled_buffer()
    riding.learner.step_n(buf.handle, 10, 64, seed=9)
    own.learner.step_n(buf.handle, 10, 64, seed=9)
    t.cuda.synchronize()
    riding.learner.check()
    assert t.isfinite(riding.actor._oprl_arena).all()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(riding, m)._oprl_arena, getattr(own, m)._oprl_arena), m
    sr, so = riding.learner.read_scalars(), own.learner.read_scalars()
    for k in ("critic_loss", "actor_loss", "alpha"):
        assert sr[k] == so[k], k


@pytest.mark.parametrize("B", [64, 100])
def test_tqc_step_n_rows_gathered_by_riders_equal_gather_launches(B, monkeypatch):
    """TQC's step_n: the next update's minibatch rows gathered by riding workgroups of the k_lw_dact launch (same
    Philox draw and index map as k_replay_gather, into the other of two row sets) against a gather launch per
    update, and against the Python loop sample() + update(): bit-identical."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger
    from tests.test_gpu_callers import _filled_buffer

    def make():
        t.manual_seed(0)
        return TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=256).create()

    riders = make()
    monkeypatch.setenv("OPRL_AMD_NO_RIDE", "8")   # (read when the learner is created)
    launches = make()
    buf = _filled_buffer()
    riders.learner.step_n(buf.handle, 9, B, seed=4)
    riders.learner.step_n(buf.handle, 1, B, seed=4)       # (K = 1: nothing to prefetch)
    riders.learner.step_n(buf.handle, 6, B, seed=4)
    launches.learner.step_n(buf.handle, 16, B, seed=4)
    t.cuda.synchronize()
    riders.learner.check()
    assert t.isfinite(riders.critic._oprl_arena).all()
    for m in ("actor", "critic", "critic_target"):
        assert t.equal(getattr(riders, m)._oprl_arena, getattr(launches, m)._oprl_arena), m


def test_tqc_wide_dw_equals_small_tiles(monkeypatch):
    """csrc/dw_wide.hip (64x64 tiles for the 512x512 layers) against k_dw_adam's 16x32 tiles: same
    gradient up to the summation order over the minibatch, same Adam / Polyak / pack epilogue."""
    from oprl_amd.algos.tqc import TQC
    from oprl_amd.logging import NullLogger
    import subprocess, sys, os
    # the switch is read once per process: run the small-tile learner in a child
    code = (
        "import torch as t, sys; sys.path.insert(0, %r)\n"
        "from oracle import fixtures as fx\n"
        "from oprl_amd.algos.tqc import TQC\n"
        "from oprl_amd.logging import NullLogger\n"
        "t.manual_seed(0)\n"
        "a = TQC(logger=NullLogger('/tmp/oprl_amd_test'), state_dim=24, action_dim=6, device='cuda', max_batch=256).create()\n"
        "for step in range(3):\n"
        "    a.update(*[x.cuda() for x in fx.make_batch(25 + step, 256, 24, 6)])\n"
        "t.cuda.synchronize()\n"
        "t.save({m: getattr(a, m)._oprl_arena.cpu() for m in ('actor', 'critic', 'critic_target')}, sys.argv[1])\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = "/tmp/oprl_amd_test_dw_small.pt"
    env = dict(os.environ, OPRL_AMD_NO_RIDE="16")
    subprocess.run([sys.executable, "-c", code, out], check=True, env=env, timeout=300)
    ref = t.load(out)
    t.manual_seed(0)
    wide = TQC(logger=NullLogger("/tmp/oprl_amd_test"), state_dim=24, action_dim=6, device="cuda", max_batch=256).create()
    for step in range(3):
        wide.update(*[x.cuda() for x in fx.make_batch(25 + step, 256, 24, 6)])
    t.cuda.synchronize()
    for m in ("actor", "critic", "critic_target"):
        a, b = getattr(wide, m)._oprl_arena.cpu(), ref[m]
        d = (a - b).abs()
        assert t.isfinite(a).all(), m
        assert d.max().item() <= 3 * 3 * 3e-4, m
        assert (d > 1e-4 * b.abs().max().item()).float().mean().item() < 2e-3, m
        assert (d > 1e-5 * b.abs().max().item()).float().mean().item() < 3e-2, m
    # packs in step with the masters: a forward through the packs the kernel wrote equals one through
    # packs rebuilt from the master parameters
    sb, ab = [x.cuda() for x in fx.make_batch(31, 256, 24, 6)[:2]]
    z0, zt0 = wide.critic(sb, ab).clone(), wide.critic_target(sb, ab).clone()
    wide.learner.sync_params()
    assert t.equal(z0, wide.critic(sb, ab)) and t.equal(zt0, wide.critic_target(sb, ab))


# ---- phase 1's B roles on 32-row slices (tp4.h tp4_scalar_fb2, fused_ddpg.hip role_b2): the over-subscribed launches
@pytest.mark.parametrize("prec", ["f32", "x2", "bf16"])
@pytest.mark.parametrize("algo,B", [("sac", 1024), ("ddpg", 512), ("td3", 512), ("ddpg", 1024)])
def test_two_row_tiles_per_cluster_equal_one_bitwise(algo, B, prec, monkeypatch):
    """At B >= 512 the critics' forward + unit-seed backward (role B) runs on clusters that carry TWO 16-row tiles — the
    fragments of the pass fetched once per 32 rows, half the workgroups of that role.  Per tile the arithmetic is the one-tile
    pass's (same fragments, same accumulator chains, same member order in the exchange): the parameters after several
    updates — through update() on caller-supplied rows and through step_n (in-kernel gather, then staged rows) — are
    bit-identical to OPRL_AMD_NO_RT2=1, and the form really is taken (the launch grid differs: checked through the rate of
    a wrong answer, i.e. the switch must matter to debug state — here: both learners finite and equal)."""
    from tests.test_gpu_callers import _filled_buffer

    def make():
        if algo == "sac":
            return _sac(67, 21, max_batch=B, tune_alpha=True, precision=prec)
        if algo == "td3":
            return _td3(24, max_batch=B, precision=prec)
        return _ddpg(max_batch=B, precision=prec)
    S, A = (67, 21) if algo == "sac" else (24, 6)
    outs = []
    # ("2", SAC at B = 1024: the B roles' two tiles, but role C a role of its own — against "0", where role A's first pass
    # carries role C's pass as a second row tile, tp4_forward2: r06-13)
    for no_rt2 in (("1", "2", "0") if algo == "sac" else ("1", "0")):
        monkeypatch.setenv("OPRL_AMD_NO_RT2", no_rt2)
        a = make()
        for step in range(3):
            batch = [x.cuda() for x in fx.make_batch(610 + step, B, S, A)]
            a.update(*batch)
        if algo != "sac":                      # (the shared test replay has walker dims)
            buf = _filled_buffer()
            a.learner.step_n(buf.handle, 5, B, seed=13)
        t.cuda.synchronize()
        a.learner.check()
        outs.append({m: getattr(a, m)._oprl_arena.clone() for m in ("actor", "critic", "critic_target")})
    for m in outs[0]:
        for o in outs[1:]:
            assert t.isfinite(o[m]).all(), m
            assert t.equal(outs[0][m], o[m]), f"{algo} B={B} {prec} {m}: max |d| = {(outs[0][m] - o[m]).abs().max().item():.3e}"
