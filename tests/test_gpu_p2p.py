"""The one-shot all-reduce over peer windows (csrc/p2p.hip) with TWO processes: the data-parallel step
through the windows must leave both replicas bit-identical and equal to a single-process emulation
that sums the two ranks' gradients in rank order."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch as t

pytestmark = pytest.mark.gpu


def make_algo(name, B, **kw):
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.algos.sac import SAC
    from oprl_amd.logging import NullLogger
    t.manual_seed(0)
    if name == "sac":
        return SAC(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=B, tune_alpha=True,
                   log_every=10 ** 9, **kw).create()
    if name == "td3":
        from oprl_amd.algos.td3 import TD3
        return TD3(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=B, log_every=10 ** 9,
                   **kw).create()
    return DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=B, **kw).create()


def make_shard(rank):
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    buf = EpisodicReplayBuffer(buffer_size_transitions=600, state_dim=24, action_dim=6, max_episode_lenth=50,
                               device="cuda", seed=3).create()
    rs = np.random.RandomState(100 + rank)
    for e in range(5):
        for i in range(50 - e):
            buf.add_transition(rs.standard_normal(24).astype(np.float32), rs.uniform(-1, 1, 6),
                               float(rs.uniform()), False, episode_done=(i == 50 - e - 1))
    return buf


# level 1: one window kernel per exchange; level 2: the exchange inside the dW + Adam launches.  Two ranks
# share this GPU, so a rank's waiting dW workgroups (level 2) and the other rank's one-workgroup-per-CU
# phase kernels compete for the same CUs: DDPG at a small batch leaves room for both, SAC's twice as many
# dW tiles do not — SAC is exercised at level 1 (which also covers the 64-bit temperature exchange).
# ("ddpg", 2, "x2"): the split-fp16 learner's data-parallel update is its SINGLE-GPU launch — k_ddpg_chain, the whole
# update — whose 16 x 64 dW tiles all-reduce their gradients with the other rank's before Adam (dw_tile_x2.h): no
# all-reduce launches, no apply launches.  Since round 6 the same holds for ("ddpg", 2, "f32") and ("ddpg", 2, "bf16"):
# k_ddpg_chain<PrecF32 / PrecBF16> with the exchange in its tiles (the worker reports the form the rank ran).
# world 4 / 3 / 8 (level 1 only: the in-tile exchange needs every rank's tile workgroups resident together, which one GPU
# offers to two ranks): the window kernels' slot arithmetic, flag counts and rank-ordered sums beyond a pair of ranks.
@pytest.mark.parametrize("algo_name,level,prec,world", [("ddpg", 2, "f32", 2), ("ddpg", 1, "f32", 2), ("sac", 1, "f32", 2), ("td3", 1, "f32", 2),
                                                        ("ddpg", 2, "x2", 2), ("ddpg", 2, "bf16", 2),
                                                        ("ddpg", 1, "f32", 4), ("sac", 1, "f32", 3), ("ddpg", 1, "f32", 8)])
def test_two_process_p2p_data_parallel_step(algo_name, level, prec, world):
    K, B = 6, 32
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        rdv, out = os.path.join(td, "rdv"), os.path.join(td, "out")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "p2p_worker.py"), str(r), str(world),
                                   rdv, out, algo_name, str(K), str(B), str(level), prec], env=env, cwd=root)
                 for r in range(world)]
        for p in procs:
            assert p.wait(timeout=300) == 0
        res = [t.load(f"{out}.{r}", weights_only=False) for r in range(world)]
    assert all(r["ok"] for r in res), f"the peer windows are not in use: {[r['why'] for r in res]}"
    if algo_name == "ddpg" and level == 2:     # every arithmetic: the rank's update is the whole-update launch
        assert all(r["form"]["dp_inline_form"] == 4 for r in res), [r["form"] for r in res]
    for m in ("critic", "actor"):
        d = (res[0]["arenas"][m] - res[1]["arenas"][m]).abs().max().item()
        for r in range(1, world):
            assert t.equal(res[0]["arenas"][m], res[r]["arenas"][m]), f"replicas diverged: {m}, rank {r} (ranks 0 / 1: max |d| = {d:.3e})"
        assert t.isfinite(res[0]["arenas"][m]).all()
    # single-process emulation: two export_grads learners, gradients summed in rank order
    L = [make_algo(algo_name, B, export_grads=True, precision=prec) for _ in range(world)]
    for r in range(world):
        L[r].learner.set_seed(0, r)      # as the ranks of the job: every rank draws its own noise
    shards = [make_shard(r) for r in range(world)]
    for r in range(world):
        shards[r].seed = (5 * 0x9E3779B97F4A7C15 + r) & (2 ** 64 - 1)
    for k in range(K):
        batches = []
        for r in range(world):
            shards[r]._sample_counter = k
            batches.append(shards[r].sample(B))
        for phase, which in ((0, "critic_grad"), (1, "actor_grad")):
            for r in range(world):
                L[r].learner.update_phase(phase, *batches[r])
            total = getattr(L[0].learner, which) + getattr(L[1].learner, which)
            for r in range(2, world):                      # (rank order, left to right: the windows' sum)
                total = total + getattr(L[r].learner, which)
            for r in range(world):
                getattr(L[r].learner, which).copy_(total)
            if phase == 1 and L[0].learner.log_alpha_grad is not None:
                ta = L[0].learner.log_alpha_grad + L[1].learner.log_alpha_grad
                for r in range(2, world):
                    ta = ta + L[r].learner.log_alpha_grad
                for r in range(world):
                    L[r].learner.log_alpha_grad.copy_(ta)
            for r in range(world):
                L[r].learner.apply(phase, 1.0 / world)
    t.cuda.synchronize()
    for m in ("critic", "actor"):
        d = (getattr(L[0], m)._oprl_arena.cpu() - res[0]["arenas"][m]).abs().max().item()
        if level == 1:     # same kernels as the emulation: bit for bit
            assert t.equal(getattr(L[0], m)._oprl_arena.cpu(), res[0]["arenas"][m]), f"{m}: max |d| vs emulation = {d:.3e}"
        elif prec == "bf16":
            # bf16 is not a parity mode: the emulation's phase launches and the ranks' whole-update launches round different
            # partial sums to bf16 (oracle.bf16_gemm(chain=True) vs (chain=False)); what bounds the distance after K updates
            # is Adam itself — an element whose gradient changes sign moves by 2 lr per update
            print(f"bf16 inline exchange, {m}: max |d| vs emulation = {d:.3e}")
            assert d <= K * 2 * 3e-4 * 1.1, f"{m}: max |d| vs emulation = {d:.3e}"
        else:
            # the emulation's phases leave dW from the 16 x 32 exact-fp32 tiles (k_dw_adam), the ranks' whole-update launches
            # from the 16 x 64 tiles (split-fp16 or exact fp32: another order of the batch sum): two parity arithmetics
            # (tests/test_gpu_x2.py::test_x2_launch_forms_agree: 2e-6)
            scale = getattr(L[0], m)._oprl_arena.abs().max().item()
            print(f"{prec} inline exchange, {m}: max |d| vs emulation = {d:.3e} (max |theta| = {scale:.3e})")
            assert d <= 2e-6 * max(scale, 1.0), f"{m}: max |d| vs emulation = {d:.3e}"


@pytest.mark.parametrize("extra,fail,expect", [
    (["--no-p2p"], "0", "rccl"),                              # RCCL only: the windows are not even probed
    (["--precision", "f32"], "0", "p2p-inline"),             # the default: all three probed, the fastest healthy one measured
    (["--precision", "f32"], "1", "rccl"),                   # ... and back to RCCL when the windows' self-test fails
    (["--precision", "x2"], "0", None),                      # the x2 learner: whichever is picked must be healthy
])
def test_bench_picks_the_exchange_and_falls_back(extra, fail, expect):
    """bench.py --force-dp (one rank): RCCL first, then — by default since round 5 — the peer-window exchanges are probed
    as well and the fastest exchange whose self-test and replica health check pass is measured; with the self-test
    forced to fail it must stay on RCCL and still report healthy replicas; --no-p2p probes RCCL alone.  The JSON line is
    the LAST line of stdout."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OPRL_AMD_P2P_SELFTEST_FAIL=fail, MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dp", "--no-cpu-baseline",
                          "--learners", "0", "--steps", "300", "--warmup", "30", *extra], env=env, cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [x for x in out.stdout.splitlines() if x.strip()][-1]
    assert line.startswith("{"), out.stdout[-500:]
    d = json.loads(line)
    chk = d["data_parallel_check"]
    assert chk["finite"] and chk["replicas_identical"], chk
    if expect is not None:
        assert chk["exchange"] == expect, chk
    if extra == ["--no-p2p"]:
        assert list(chk["probe_us_per_step"]) == ["rccl"], chk
    elif fail == "0":
        assert set(chk["probe_us_per_step"]) == {"rccl", "p2p", "p2p-inline"}, chk
    assert d["n_gpus"] == 1 and d["value"] > 1000


@pytest.mark.parametrize("prec", ["x2", "f32"])
def test_bench_two_rank_rehearsal_on_one_gpu(prec):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, two ranks), both ranks on THIS GPU
    (OPRL_BENCH_REHEARSAL=1: gloo rendezvous, peer windows only — RCCL refuses two ranks on one device — B = 32, one update
    per launch): the N > 1 path of the file itself — probe of both window levels on fresh replicas, the rebuild of the
    fastest, warm-up, the barrier-bracketed timed region with the max over ranks, the replica check, ONE line from rank 0 —
    with the x2 learner's whole-update launches exchanging inside their tiles.  Not a measurement (the line says so)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OPRL_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MASTER_PORT", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2",
                          "--steps", "40", "--warmup", "5", "--pre-warm", "40", "--profile-steps", "40", "--precision", prec],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [x for x in out.stdout.splitlines() if x.strip().startswith("{")][-1]
    d = json.loads(line)
    chk = d["data_parallel_check"]
    assert d["n_gpus"] == 2 and d["steps"] == 40 and d["value"] > 0 and d["rehearsal"], d
    assert chk["finite"] and chk["replicas_identical"], chk
    assert set(chk["probe_us_per_step"]) == {"p2p", "p2p-inline"} and chk["exchange"] in ("p2p", "p2p-inline"), chk
    # in either parity mode a rank that exchanges inside its tiles runs the single-GPU whole-update launch
    assert chk["inline_form"] == 4, chk
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 64, d["config"]


@pytest.mark.parametrize("n", [4, 8])
def test_bench_many_rank_rehearsal_on_one_gpu(n):
    """The same rehearsal with 4 and 8 ranks on this GPU — world sizes the driver's scaling run uses.  The in-tile exchange
    cannot come up here (every rank's tile workgroups would have to be resident together on ONE chip: its probe fails or
    is dropped, by design), so the line must come from the window kernels (level 1) with bit-identical replicas: rank
    bookkeeping, window slots, flag counts, the max over ranks and the one line from rank 0 at dp4 / dp8."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OPRL_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.pop("MASTER_PORT", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                          "--master-addr", "127.0.0.1", "--master-port", str(29560 + n), os.path.join(root, "bench.py"), "--gpus", str(n),
                          "--steps", "40", "--warmup", "5", "--pre-warm", "40", "--profile-steps", "40", "--precision", "f32"],
                         env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [x for x in out.stdout.splitlines() if x.strip().startswith("{")][-1]
    d = json.loads(line)
    chk = d["data_parallel_check"]
    assert d["n_gpus"] == n and d["steps"] == 40 and d["value"] > 0 and d["rehearsal"], d
    assert chk["finite"] and chk["replicas_identical"], chk
    assert chk["exchange"] in ("p2p", "p2p-inline") and "p2p" in chk["probe_us_per_step"], chk
    assert d["config"]["parallelism"] == f"dp{n}" and d["config"]["global_batch"] == 32 * n, d["config"]


def test_bench_prints_its_line_when_the_run_fails():
    """Whatever goes wrong after start-up — here: a watchdog of a fraction of a second — rank 0 still prints ONE
    JSON line (value null, the reason in `error`) and the exit code says so."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--learners", "0",
                          "--no-configs", "--steps", "200000", "--warmup", "30", "--watchdog", "0.5"], cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    line = [x for x in out.stdout.splitlines() if x.strip()][-1]
    d = json.loads(line)
    assert d["value"] is None and "watchdog" in d["error"] and d["n_gpus"] == 1, d
