"""Headline benchmark: learner gradient steps/sec, DDPG batch=256 at walker-walk
dims (S=24, A=6), on N MI355X GPUs of one node (BASELINE.json metric).

A "step" = one sample() + update(): device-side uniform sampling from a 1e6-
transition HBM-resident replay (gather kernel) -> TD target -> critic
forward/backward + Adam -> actor forward/backward + Adam -> Polyak, all in the
hand-written HIP path (oprl_learner_step_n, or update_phase/apply + RCCL
all-reduce of the gradients for N > 1).  Inputs are resident in HBM before the
timed region.

    python bench.py --gpus 1 --steps 20000 --warmup 1000
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The timed learner runs EXACT fp32 on the matrix cores (OPRL_PREC_F32,
v_mfma_f32_16x16x4_f32: the arithmetic of the reference and the default of every class
and script of the package).  `--precision x2` times the other parity mode (OPRL_PREC_X2:
every fp32 operand as the sum of two fp16 numbers, three v_mfma_f32_16x16x32_f16 per
product, fp32 accumulate / master weights / Adam — held to the reference's golden vectors
at the exact-fp32 mode's gates, tests/test_gpu_x2.py), `--precision bf16` the
reduced-precision one.  The default run reports all three (blocks `x2`, `bf16`).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     dominant kernel (k_ddpg_chain: up to 32 whole updates per launch):
               algorithmic bytes / FLOP per launch over its average duration (HIP
               events on the launch stream, measured live in a separate instrumented
               pass) against the binding peak
  cpu_baseline the CPU oracle (port of the reference's torch-CPU update path)
               timed on this box's host cores on a bounded sample of the same
               workload.
Whatever goes wrong after start-up, rank 0 still prints one JSON line (`value` null and
an `error` field): a watchdog bounds every phase of the run.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch as t  # noqa: E402

S, A, B = 24, 6, 256
HID = 256
E, L = 1000, 1000                      # 1e6 transitions (configs/ddpg.py:49)
PEAK_F32_MATRIX_TFLOPS = 157.3         # MI355X_MICROARCH.md (v_mfma_f32_16x16x4_f32)
# algorithmic MACs per sample: F = sum(in*out) per net
F_ACTOR = S * HID + HID * HID + HID * A
F_CRITIC = (S + A) * HID + HID * HID + HID * 1
# slice kernels: 2 actor fwd + 3 critic fwd + critic dX (hidden only) +
# critic dX incl. action columns + actor dX (hidden only)
MACS_SLICE = (2 * F_ACTOR + 3 * F_CRITIC + (HID + HID * HID) + (HID + HID * HID + HID * A)
              + (A * HID + HID * HID))
MACS_DW = F_ACTOR + F_CRITIC           # one dW pass per trained net
# fused path (csrc/fused_ddpg.hip): phase 1 = actor_t, critic_t, critic, actor forwards +
# critic hidden backward; phase 2 = critic forward, critic backward incl. action columns,
# actor hidden backward
MACS_P1 = 2 * F_ACTOR + 2 * F_CRITIC + (HID + HID * HID)
MACS_P2 = F_CRITIC + (HID + HID * HID + HID * A) + (A * HID + HID * HID)
assert MACS_P1 + MACS_P2 == MACS_SLICE
STATE_BYTES = 32 * (F_ACTOR + HID * 2 + A + F_CRITIC + HID * 2 + 1)  # 32 B per trainable param


def make_replay(device, seed, S=S, A=A):
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    buf = EpisodicReplayBuffer(buffer_size_transitions=E * L, state_dim=S, action_dim=A,
                               device=str(device), seed=seed).create()
    g = t.Generator(device=device).manual_seed(1234 + seed)
    buf._tensors["states"].copy_(t.randn((E, L + 1, S), device=device, generator=g))
    buf._tensors["actions"].copy_(t.rand((E, L, A), device=device, generator=g) * 2 - 1)
    buf._tensors["rewards"].copy_(t.rand((E, L, 1), device=device, generator=g))
    buf._tensors["dones"].zero_()            # dm_control never terminates (dm_control.py:31)
    buf.ep_lens = [L] * E
    buf.episodes_counter = E
    buf._number_transitions = E * L
    buf._lens_dirty = True
    return buf


def cpu_baseline(budget_s: float = 14.0):
    """The CPU oracle (oracle/oprl_oracle.py, validated against the reference by
    tests/test_oracle_golden.py) on the same workload: numpy-index sampling from
    a host replay + one DDPG update per step.  Bounded by wall time.  torch's
    default of one thread per host core is pathological for 256-wide GEMMs (the
    128-thread box ran 5 steps/s), so it is timed at 1 and at 8 threads and the
    faster is reported with the thread count actually used."""
    from oracle import fixtures as fx
    from oracle import oprl_oracle as orc
    rs = np.random.RandomState(0)
    n_ep = 50                              # 50k host-resident transitions are enough to time
    rep = orc.ReplayOracle(n_ep * L, S, A, max_episode_lenth=L)
    rep.states[:] = rs.standard_normal(rep.states.shape).astype(np.float32)
    rep.actions[:] = rs.uniform(-1, 1, rep.actions.shape).astype(np.float32)
    rep.rewards[:] = rs.uniform(0, 1, rep.rewards.shape).astype(np.float32)
    rep.dones[:] = 0
    rep.ep_lens = [L] * n_ep
    rep.episodes_counter = n_ep
    rep.n = n_ep * L
    algo = orc.DDPGOracle(S, A, fx.make_net(1, fx.actor_dims(S, A)), fx.make_net(2, fx.critic_dims(S, A)))

    def step():
        inds = np.random.randint(0, rep.n, size=B)
        batch = [t.from_numpy(np.ascontiguousarray(x)) for x in rep.gather(inds)]
        algo.update(*batch)

    saved = t.get_num_threads()
    results = {}
    try:
        for threads in (1, min(8, os.cpu_count() or 1)):
            t.set_num_threads(threads)
            for _ in range(5):
                step()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s / 2:
                step()
                n += 1
            results[threads] = (n / (time.perf_counter() - t0), n)
    finally:
        t.set_num_threads(saved)
    best = max(results, key=lambda k: results[k][0])
    detail = ", ".join(f"{k} thread(s): {v[0]:.1f} steps/s over {v[1]} steps" for k, v in results.items())
    return dict(value=round(results[best][0], 2), unit="steps/s", cores=best, kind="port",
                sample=f"DDPG sample+update (B={B}, walker dims) with the torch-CPU oracle, "
                       f"{budget_s / 2:.0f} s per setting: {detail}; host has {os.cpu_count()} cores, "
                       f"torch {t.__version__}")


# ---- the other BASELINE.json configurations, the bf16 mode and the through-the-API rate --------
PEAK_BF16_MATRIX_TFLOPS = 2500.0       # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F16_MATRIX_TFLOPS = 2500.0        # dense fp16 MFMA; the x2 mode spends three products per algorithmic one
PMC_FILE = {"f32": "r06_pmc_traffic_f32.json", "bf16": "r06_pmc_traffic_bf16.json", "x2": "r06_pmc_traffic_x2.json"}
PEAK_OF = {"f32": PEAK_F32_MATRIX_TFLOPS, "bf16": PEAK_BF16_MATRIX_TFLOPS, "x2": PEAK_F16_MATRIX_TFLOPS / 3.0}
DTYPE_OF = {"f32": "f32 (exact-fp32 MFMA)", "bf16": "bf16 (fp32 accumulate / master / Adam)",
            "x2": "f32 as split fp16x2 (hi + lo, 3 fp16 MFMAs per product, fp32 accumulate / master / Adam)"}
PEAK_HBM_TBS = 8.0
# name -> (class name, S, A, B, constructor extras, algorithmic GFLOP / update, state MB / update): BASELINE.md section 4
BASELINE_CONFIGS = {
    "DDPG walker-walk B=256": ("DDPG", 24, 6, 256, {}, 0.365, 4.73),
    # (the batch configs/ddpg.py really trains at: BaseTrainer.batch_size = 128, trainers/base_trainer.py:28)
    "DDPG walker-walk B=128 (the reference scripts' batch)": ("DDPG", 24, 6, 128, {}, 0.1825, 4.73),
    "TD3 cheetah-run B=256": ("TD3", 17, 6, 256, {"log_every": 10 ** 9}, 0.413, 5.19),
    "SAC humanoid-walk B=1024": ("SAC", 67, 21, 1024, {"log_every": 10 ** 9}, 2.74, 7.94),
    "TQC walker-walk B=256 5x25": ("TQC", 24, 6, 256, {"log_every": 10 ** 9}, 8.57, 90.4),
}


def _make_algo(cls_name, S_, A_, B_, extras, dev, precision="f32", seed=0):
    import importlib
    from oprl_amd.logging import NullLogger
    cls = getattr(importlib.import_module(f"oprl_amd.algos.{cls_name.lower()}"), cls_name)
    t.manual_seed(seed)
    return cls(logger=NullLogger(), state_dim=S_, action_dim=A_, device=str(dev), max_batch=B_,
               precision=precision, **extras).create()


def _time_step_n(algo, replay, B_, n, dev):
    L_ = algo.learner
    L_.step_n(replay.handle, max(n // 10, 50), B_, seed=1)
    best = 1e30
    for _rep in range(2):
        t.cuda.synchronize(dev)
        t0 = time.perf_counter()
        L_.step_n(replay.handle, n, B_, seed=2)
        t.cuda.synchronize(dev)
        best = min(best, time.perf_counter() - t0)
    return best / n


def config_table(dev, replays, n_steps=2000):
    """Every BASELINE.json single-GPU configuration through the fused step_n path, the two parity modes (x2,
    exact fp32) and bf16: updates/s, us/update and the fraction of the binding roof for the WHOLE update — the larger
    of (algorithmic FLOP / matrix peak of the mode) and (state bytes / HBM peak) over the measured time."""
    rows = []
    for name, (cls_name, S_, A_, B_, extras, gflop, mbytes) in BASELINE_CONFIGS.items():
        replay = replays(S_, A_)
        for prec in ("x2", "f32", "bf16"):
            peak = PEAK_OF[prec]
            try:
                algo = _make_algo(cls_name, S_, A_, B_, extras, dev, prec)
            except Exception as exc:  # noqa: BLE001  (a mode an algorithm does not have: said, not hidden)
                rows.append(dict(name=name, dtype=prec, steps_per_s=None, unsupported=str(exc)[:120]))
                continue
            n = n_steps if cls_name != "TQC" else max(n_steps // 4, 200)
            sec = _time_step_n(algo, replay, B_, n, dev)
            t_mfma, t_hbm = gflop * 1e9 / (peak * 1e12), mbytes * 1e6 / (PEAK_HBM_TBS * 1e12)
            rows.append(dict(name=name, dtype=prec, steps_per_s=round(1.0 / sec, 1), us_per_step=round(sec * 1e6, 2),
                             roof="mfma" if t_mfma >= t_hbm else "hbm",
                             roofline_frac=round(max(t_mfma, t_hbm) / sec, 5), path="oprl_learner_step_n",
                             steps=n))
            algo.learner.check()
            del algo
    return rows


def bf16_q_deviation(dev, replay, updates=10):
    """Q(s, a) of an fp32 and a bf16 DDPG learner after the same `updates` updates (same initial weights,
    same minibatches): max-norm relative deviation on a fixed probe batch — the accuracy price of the mode."""
    a32 = _make_algo("DDPG", S, A, B, {}, dev, "f32")
    a16 = _make_algo("DDPG", S, A, B, {}, dev, "bf16")
    g = t.Generator(device=dev).manual_seed(7)
    ps = t.randn((B, S), device=dev, generator=g)
    pa = t.rand((B, A), device=dev, generator=g) * 2 - 1
    for a in (a32, a16):
        a.learner.step_n(replay.handle, updates, B, seed=11)
    q32, q16 = a32.critic(ps, pa), a16.critic(ps, pa)
    return float((q16 - q32).abs().max() / q32.abs().max())


def api_rate(dev, replay, precision="x2", n=3000):
    """The reference's call pattern from Python, one call pair per step: replay_buffer.sample(B) then
    algo.update(*batch) (trainers/base_trainer.py:63-70)."""
    algo = _make_algo("DDPG", S, A, B, {}, dev, precision)
    for _ in range(200):
        algo.update(*replay.sample(B))
    t.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        algo.update(*replay.sample(B))
    t.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return dict(value=round(n / dt, 1), unit="steps/s", us_per_step=round(dt / n * 1e6, 2), steps=n,
                dtype=precision,
                path="EpisodicReplayBuffer.sample() + DDPG.update() from Python (two C calls: the gather kernel, the update's launch(es))")


def multi_learner(n, dev, local_rank, steps):
    """Aggregate steps/s of n independent DDPG learners (own weights, own replay seed)
    driven from one host thread on n streams — the reference's ``--seeds`` fan-out
    (runners/train.py:36-50) without one process per seed.  Each learner's update is
    4 asynchronous launches; the launches of different learners overlap on the chip.
    Reported beside the headline, never as it.  The fused kernels contain bounded
    cross-workgroup waits, so the packed run is VERIFIED: every learner's parameters must be
    finite and learner 0 must equal, bit for bit, a solo run with the same seeds."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    algos, streams, replays = [], [], []
    for i in range(n):
        t.manual_seed(100 + i)
        algos.append(DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device=f"cuda:{local_rank}",
                          max_batch=B).create())
        algos[-1].learner.set_cluster(4)       # learners that share the chip: clusters of four only (include/oprl_amd.h)
        streams.append(t.cuda.Stream(device=dev))
    shared = make_replay(dev, seed=7)          # one HBM replay, n sampler keys
    chunk = 50

    handle = shared.handle                     # (flushes the replay once, on this thread)

    def run_one(i, k):
        with t.cuda.stream(streams[i]):
            for _ in range(k // chunk):
                algos[i].learner.step_n(handle, chunk, B, seed=1000 + i)

    def run(k):
        # one host thread per learner: ctypes drops the GIL for the duration of step_n, so the
        # launches of different learners are issued concurrently (a single thread tops out at
        # ~3.3 us per launch)
        import threading
        ths = [threading.Thread(target=run_one, args=(i, k)) for i in range(n)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
    run(chunk * 2)
    t.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(steps)
    t.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    done = (steps // chunk) * chunk * n
    finite = all(bool(t.isfinite(a.actor._oprl_arena).all()) and bool(t.isfinite(a.critic._oprl_arena).all())
                 for a in algos)
    # the same update stream, alone on the GPU
    t.manual_seed(100)
    solo = DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device=f"cuda:{local_rank}", max_batch=B).create()
    solo.learner.set_cluster(4)
    total = chunk * 2 + (steps // chunk) * chunk
    for _ in range(total // chunk):
        solo.learner.step_n(handle, chunk, B, seed=1000)
    t.cuda.synchronize(dev)
    same = bool(t.equal(solo.actor._oprl_arena, algos[0].actor._oprl_arena)) and \
        bool(t.equal(solo.critic._oprl_arena, algos[0].critic._oprl_arena))
    return dict(learners=n, value=round(done / dt, 1), unit="steps/s (aggregate)",
                per_learner=round(done / dt / n, 1), steps_each=(steps // chunk) * chunk,
                verified=dict(all_finite=finite, learner0_equals_solo_run=same))


def self_launch(n_gpus: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start the N ranks here (one process per GPU,
    ``torch.distributed.run`` on 127.0.0.1) and hand their output through; rank 0 prints the JSON line.
    With fewer than N GPUs on the node nothing can be measured: one JSON line saying so, exit code 0."""
    import socket
    import subprocess
    have = t.cuda.device_count() if t.cuda.is_available() else 0
    if have < n_gpus:
        why = f"--gpus {n_gpus} needs {n_gpus} GPUs on this node, found {have}: nothing measured"
        print(f"bench.py: {why}", file=sys.stderr)
        print(json.dumps({"metric": METRIC, "value": None, "unit": "steps/s", "n_gpus": n_gpus, "skipped": why}), flush=True)
        return 0
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd).returncode


def packed_group(n, dev, local_rank, steps):
    """N3 as a batched step: n independent DDPG learners in a LearnerGroup (oprl_group_step_n: four launches per
    update for ALL members, grid.z = learner, single-CU slices).  Verified: every member finite, member 0
    bit-identical to the same learner stepped alone at cluster size 1."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.group import LearnerGroup
    from oprl_amd.logging import NullLogger

    def member(i):
        t.manual_seed(100 + i)
        return DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device=f"cuda:{local_rank}", max_batch=B).create()
    algos = [member(i) for i in range(n)]
    shared = make_replay(dev, seed=7)
    handle = shared.handle
    seeds = [1000 + i for i in range(n)]
    g = LearnerGroup(algos)
    warm = 100
    g.step_n(handle, warm, B, seeds)
    t.cuda.synchronize(dev)
    t0 = time.perf_counter()
    g.step_n(handle, steps, B, seeds)
    t.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    finite = all(bool(t.isfinite(a.actor._oprl_arena).all()) and bool(t.isfinite(a.critic._oprl_arena).all()) for a in algos)
    solo = member(0)
    _capi_check = __import__("oprl_amd._capi", fromlist=["check"]).check
    _capi_check(solo.learner.lib.oprl_learner_set_cluster(solo.learner.handle, 1))
    solo.learner.step_n(handle, warm, B, seed=seeds[0])
    solo.learner.step_n(handle, steps, B, seed=seeds[0])
    t.cuda.synchronize(dev)
    same = bool(t.equal(solo.actor._oprl_arena, algos[0].actor._oprl_arena)) and \
        bool(t.equal(solo.critic._oprl_arena, algos[0].critic._oprl_arena))
    for a in algos:
        a.learner.check()
    g.close()
    del algos, g, solo
    return dict(learners=n, value=round(n * steps / dt, 1), unit="steps/s (aggregate)",
                us_per_group_update=round(dt / steps * 1e6, 1), steps_each=steps,
                path="oprl_group_step_n: 4 launches per update for the whole group (grid.z = learner, member l on XCD l % 8; one argument copy per 4 updates), exact fp32, cluster size 1",
                verified=dict(all_finite=finite, member0_equals_solo_run_at_cluster_1=same),
                other_members=packed_group_variants(n, dev, local_rank, handle, max(100, steps // 3)))


def packed_group_variants(n, dev, local_rank, handle, steps):
    """The same group call with members of the other kinds it takes (TD3 / SAC: the reference's --seeds fan-out is
    algorithm-agnostic; the x2 parity mode): lean passes on clusters of four.  Short runs; verified finite and clean."""
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.algos.sac import SAC
    from oprl_amd.algos.td3 import TD3
    from oprl_amd.group import LearnerGroup
    from oprl_amd.logging import NullLogger
    out = []
    for name, cls, kw, prec in (("DDPG", DDPG, {}, "x2"), ("TD3", TD3, dict(log_every=10 ** 9), "x2"),
                                ("SAC (learned temperature)", SAC, dict(log_every=10 ** 9, tune_alpha=True), "x2")):
        algos = []
        for i in range(n):
            t.manual_seed(300 + i)
            algos.append(cls(logger=NullLogger(), state_dim=S, action_dim=A, device=f"cuda:{local_rank}", max_batch=B,
                             precision=prec, **kw).create())
        g = LearnerGroup(algos)
        seeds = [2000 + i for i in range(n)]
        g.step_n(handle, 50, B, seeds)
        t.cuda.synchronize(dev)
        t0 = time.perf_counter()
        g.step_n(handle, steps, B, seeds)
        t.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        for a in algos:
            a.learner.check()
        finite = all(bool(t.isfinite(a.actor._oprl_arena).all()) and bool(t.isfinite(a.critic._oprl_arena).all()) for a in algos)
        g.close()
        del algos, g
        out.append(dict(algo=name, dtype=DTYPE_OF[prec], learners=n, value=round(n * steps / dt, 1), unit="steps/s (aggregate)",
                        steps_each=steps, all_finite=finite))
    return out


def dp_single_rank(dev, local_rank, replay, precision, steps):
    """The data-parallel step (oprl_learner_dp_step_n: update_phase / apply with the two gradient all-reduces on RCCL,
    all in C) with ONE rank — what the N > 1 runs execute per rank, minus the wire: its rate against the headline's is
    the fixed price of the exchange structure (un-merged launches + two ncclAllReduce calls per update)."""
    import socket
    import torch.distributed as dist
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    from oprl_amd.parallel import DataParallelLearner
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
    try:
        t.manual_seed(0)
        algo = DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device=f"cuda:{local_rank}", max_batch=B,
                    export_grads=True, precision=precision).create()
        dp = DataParallelLearner(algo, dist.group.WORLD)
        dp.init_native_comm()
        dp.broadcast_parameters()
        dp.step_n(replay.handle, 300, B, seed=0)
        t.cuda.synchronize(dev)
        t0 = time.perf_counter()
        dp.step_n(replay.handle, steps, B, seed=0)
        t.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        ok = dp.healthy()
        finite = bool(t.isfinite(algo.actor._oprl_arena).all() and t.isfinite(algo.critic._oprl_arena).all())
        out = dict(value=round(steps / dt, 1), unit="steps/s", us_per_step=round(dt / steps * 1e6, 2), steps=steps,
                   dtype=precision, exchange="rccl", healthy=bool(ok), finite=finite,
                   path="oprl_learner_dp_step_n at world size 1 (the --gpus N path per rank, RCCL all-reduce of one rank)")
        # ... and the same loop with the exchange INSIDE the dW tiles (peer windows, `--p2p` at N > 1): for a PrecX2
        # learner the data-parallel update is then the single-GPU launch itself (k_ddpg_chain; dw_tile_x2.h), for the
        # others the dW launches exchange their own tiles (k_dw_adam<true>).  One rank: no peer, the structure's fixed price.
        try:
            if dp.init_p2p(2):
                dp.step_n(replay.handle, 300, B, seed=0)
                t.cuda.synchronize(dev)
                t0 = time.perf_counter()
                dp.step_n(replay.handle, steps, B, seed=0)
                t.cuda.synchronize(dev)
                dt2 = time.perf_counter() - t0
                fin2 = bool(t.isfinite(algo.actor._oprl_arena).all() and t.isfinite(algo.critic._oprl_arena).all())
                out["inline"] = dict(value=round(steps / dt2, 1), unit="steps/s", us_per_step=round(dt2 / steps * 1e6, 2),
                                     exchange="p2p-inline", healthy=bool(dp.healthy()), finite=fin2,
                                     path="the same loop, the gradient exchange inside the dW tiles of the update's own "
                                          "launch(es) over peer windows (one rank: no peer)")
            else:
                out["inline"] = dict(value=None, error=dp.p2p_error[:200])
        except Exception as exc:  # noqa: BLE001
            out["inline"] = dict(value=None, error=f"{type(exc).__name__}: {exc}"[:300])
        return out
    finally:
        dist.destroy_process_group()


METRIC = "learner gradient steps/sec, DDPG batch=256 walker-walk"


FALLBACK = {}      # what a later failure may still report: the RCCL measurement of a multi-GPU run, taken with the run's own K / W before anything else is tried (measure() fills it)


def failure_line(args, world, why):
    """The one JSON line of a run that could not finish: same keys, value null, the reason — or, when the RCCL exchange of a
    data-parallel run had already been measured (the run's own W warm-up + K timed updates) and a LATER phase (a
    peer-window probe, the rebuild) failed, that measurement as `value` (`fallback` says so; the exit code is non-zero)."""
    fb = dict(FALLBACK)
    return json.dumps({"metric": METRIC, "value": fb.get("value"), "unit": "steps/s", "n_gpus": world,
                       "steps": fb.get("steps", args.steps), "warmup": fb.get("warmup", args.warmup),
                       "ms_per_step": fb.get("ms_per_step"), "higher_is_better": True, "scaling": "weak",
                       "vs_baseline": None, "dtype": DTYPE_OF[args.precision], "data": "synthetic",
                       "config": {"workload": f"DDPG walker-walk dims S={S} A={A} B={B}", "parallelism": f"dp{world}"},
                       "roofline": None, "cpu_baseline": None, "data_parallel_check": fb.get("data_parallel_check"),
                       "fallback": fb.get("note"), "error": why})


class Watchdog:
    """A hung collective (or a kernel that never returns) must not leave the driver without a line: every phase of the
    run re-arms this timer; when a phase outlives it, rank 0 prints the failure line and every rank leaves through
    os._exit (ctypes and torch drop the GIL inside their C calls, so this thread runs while the main one is stuck)."""

    def __init__(self, seconds, args, world, rank):
        import threading
        self.seconds, self.args, self.world, self.rank = seconds, args, world, rank
        self.phase, self.deadline, self.done = "start-up", time.monotonic() + seconds, False
        self.lock = threading.Lock()
        threading.Thread(target=self._watch, daemon=True).start()

    def kick(self, phase, seconds=None):
        """(`seconds`: a shorter bound for this phase — the peer-window probes, never yet run across real xGMI links: if one
        hangs, the line with the RCCL measurement goes out after minutes, not after the run's whole allowance)"""
        with self.lock:
            self.bound = self.seconds if seconds is None else min(seconds, self.seconds)
            self.phase, self.deadline = phase, time.monotonic() + self.bound

    def finish(self):
        with self.lock:
            self.done = True

    def teardown(self, seconds=60.0):
        """The line is out (or this rank has nothing to print): whatever is left — the closing barrier, the process
        group's destruction — may take this long, then the process leaves quietly with what it has."""
        with self.lock:
            self.quiet, self.phase, self.deadline = True, "teardown", time.monotonic() + seconds

    def _watch(self):
        while True:
            time.sleep(1.0)
            with self.lock:
                if self.done:
                    return
                late = time.monotonic() > self.deadline
                phase = self.phase
            if late and getattr(self, "quiet", False):
                os._exit(0)
            if late:
                why = f"watchdog: phase '{phase}' exceeded {getattr(self, 'bound', self.seconds):.0f} s on rank {self.rank}"
                print(f"bench.py: {why}", file=sys.stderr, flush=True)
                if self.rank == 0:
                    print(failure_line(self.args, self.world, why), flush=True)
                os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=2000)
    ap.add_argument("--learners", type=int, default=8,
                    help="extra measurement: this many independent learners (seeds) on separate "
                         "streams of the same GPU (multi-seed packing, runners/train.py --seeds); 0 = skip")
    ap.add_argument("--precision", choices=("f32", "bf16", "x2"), default="f32",
                    help="arithmetic mode of the TIMED learner: f32 = exact-fp32 MFMA, the reference's arithmetic, the "
                         "package's default and the headline; x2 = fp32 operands as fp16 hi + lo, three fp16 MFMAs per "
                         "product (a parity mode: same gates against the reference's vectors as f32); bf16 = the "
                         "reduced-precision mode (the default run reports both in its `x2` / `bf16` blocks)")
    ap.add_argument("--group", type=int, default=32,
                    help="extra measurement: a LearnerGroup of this many learners stepped by one launch sequence")
    ap.add_argument("--pre-warm", type=int, default=20000,
                    help="untimed updates BEFORE the --warmup ones (GPU clock ramp, first touches): the driver's short "
                         "runs (--steps 20 --warmup 5) otherwise time the learner at ramping clocks, 5 %% low")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the extra blocks: the other BASELINE.json configs (TD3 / SAC / TQC, fp32 and bf16), the "
                         "bf16 DDPG line and the through-the-API rate")
    ap.add_argument("--config-steps", type=int, default=2000)
    ap.add_argument("--p2p", action="store_true", help="(accepted for compatibility: probing the peer windows is the default)")
    ap.add_argument("--no-p2p", action="store_true",
                    help="data-parallel path: RCCL only.  Default: after RCCL has produced its number, the peer-window "
                         "exchanges (csrc/p2p.hip: one kernel per exchange | inside the dW tiles) are probed as well — each "
                         "under the watchdog, the windows' self-test and the replica check — all three rates are reported "
                         "and the fastest healthy one is measured")
    ap.add_argument("--watchdog", type=float, default=900.0,
                    help="seconds any one phase of the run may take before rank 0 prints a JSON line with value null "
                         "and the process exits (a hung collective must not leave the driver without a line)")
    ap.add_argument("--force-dp", action="store_true",
                    help="use the data-parallel path (RCCL all-reduce) even with one rank")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    wd = Watchdog(args.watchdog, args, world, rank)
    try:
        measure(args, wd)
        wd.finish()
    except BaseException as exc:  # noqa: BLE001  (whatever it was: the line is printed, then the error is shown)
        if isinstance(exc, SystemExit) and exc.code in (0, None):
            raise
        wd.finish()
        import traceback
        traceback.print_exc()
        if rank == 0:
            print(failure_line(args, world, f"{type(exc).__name__}: {exc}"[:400]), flush=True)
        sys.stdout.flush()
        os._exit(1)       # (not sys.exit: a peer stuck in a collective would keep this rank's teardown waiting)


def measure(args, wd):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: measuring WORLD_SIZE ranks", file=sys.stderr)
    assert t.cuda.is_available(), "bench.py needs an MI355X"
    # REHEARSAL (tests/test_gpu_p2p.py, OPRL_BENCH_REHEARSAL=1): the N-rank path of this file — probes, rebuild, timed region,
    # replica check, the line — with every rank on GPU 0: rendezvous over gloo, no RCCL (it refuses two ranks on one device),
    # peer windows only, a small batch and one update per launch (two ranks' whole-chip launches cannot share a GPU).  Not a
    # measurement: the line says so.
    rehearsal = os.environ.get("OPRL_BENCH_REHEARSAL") == "1" and world > 1
    if rehearsal:
        global B
        B = int(os.environ.get("OPRL_BENCH_REHEARSAL_B", "32"))
        os.environ["OPRL_AMD_CHAIN"] = "1"
        local_rank = 0
    t.cuda.set_device(local_rank)
    dev = t.device("cuda", local_rank)

    from oprl_amd import _capi
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    lib = _capi.load()

    dist = None
    use_dp = world > 1 or args.force_dp
    use_p2p = False
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        import datetime
        wd.kick("RCCL rendezvous")
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=max(60.0, args.watchdog)))
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev,
                                    timeout=datetime.timedelta(seconds=max(60.0, args.watchdog)))

    replay = make_replay(dev, seed=rank)               # disjoint shard per rank
    K, W = args.steps, args.warmup

    def barrier():
        t.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            t.cuda.synchronize(dev)

    def all_max(x):
        """max over ranks of a python float (the rehearsal's gloo group reduces on the host)"""
        tt_ = t.tensor([x], dtype=t.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt_, op=dist.ReduceOp.MAX)
        return float(tt_.item())

    def make_learner():
        t.manual_seed(0)                               # reference-style init, same on all ranks
        return DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device=f"cuda:{local_rank}",
                    max_batch=B, export_grads=use_dp, precision=args.precision).create()

    dp = None
    p2p_level = 0
    if not use_dp:
        algo = make_learner()
        learner = algo.learner

        def run(n):
            learner.step_n(replay.handle, n, B, seed=0)
        run(args.pre_warm)          # clock ramp / first-touch, before the W warm-up steps of the contract
        for _ in range(3 if args.pre_warm > 0 else 0):
            barrier()               # (... and calls of the timed call's own shape, bracketed like the timed region:
            run(min(K, 32))         # host-side first touches; the first drained short call reads 8 us longer)
        barrier()
        run(W)
    else:
        from oprl_amd.parallel import DataParallelLearner
        # The two gradient exchanges per update can run (2) inside the dW + Adam launches over peer
        # windows (csrc/p2p.hip, k_dw_adam<true>), (1) as one window kernel per exchange, or (0) on RCCL.
        # A level is usable only if every rank's window self-test passed AND, after some updates, every
        # replica is finite and identical.  Which usable level is fastest depends on the node (xGMI
        # store granularity against RCCL's ring latency), so each one is PROBED on fresh replicas and the
        # fastest is rebuilt for the measurement.
        def build(level):
            algo_ = make_learner()
            dp_ = DataParallelLearner(algo_, dist.group.WORLD)
            if not rehearsal:
                dp_.init_native_comm()
                dp_.broadcast_parameters()         # (oprl_comm_broadcast_params: ncclBroadcast from rank 0 in C)
            ok_ = True
            if level > 0:
                try:
                    ok_ = dp_.init_p2p(level)
                except Exception as exc:  # noqa: BLE001
                    ok_ = False
                    if rank == 0:
                        print(f"bench.py: peer windows unavailable ({exc})", file=sys.stderr)
                if not ok_ and rank == 0:
                    print(f"bench.py: peer-window level {level} not available ({dp_.p2p_error})", file=sys.stderr)
            if ok_:
                dp_.step_n(replay.handle, 2, B, seed=0)          # (a broken exchange shows at once: bounded waits, NaN)
                ok_ = dp_.healthy()
            return algo_, dp_, ok_

        probes = {}
        # RCCL first, always: it is the exchange that is known to work, and once it has produced its number a failure of
        # anything later still reports that number (FALLBACK).  Then the peer-window levels — never yet run across real
        # xGMI links — by default: each under the watchdog, the windows' self-test, two trial updates and the replica check;
        # DESIGN.md section 6 holds the latency budget their rates are to be checked against.
        levels = (0,) if args.no_p2p else (0, 1, 2)
        if rehearsal:
            levels = (1, 2)
        n_probe = 1000 if not rehearsal else 60
        for level in levels:
            wd.kick(f"data-parallel probe, exchange level {level}", None if level == 0 else 240.0)
            try:
                algo, dp, ok = build(level)
            except Exception as exc:  # noqa: BLE001
                if level == 0:
                    raise
                if rank == 0:
                    print(f"bench.py: exchange level {level} failed to build ({exc})", file=sys.stderr)
                continue
            if ok:
                dp.step_n(replay.handle, 300 if not rehearsal else 20, B, seed=0)
                best = 1e30
                for _rep in range(2):                     # best of two n_probe-update probes
                    barrier()
                    tp0 = time.perf_counter()
                    dp.step_n(replay.handle, n_probe, B, seed=0)
                    barrier()
                    best = min(best, all_max(time.perf_counter() - tp0))
                ok = dp.healthy()
                if ok:
                    probes[level] = best / n_probe * 1e6
                    if level == 0:
                        # ... and the contract's own protocol on RCCL at once — W warm-up updates, K timed ones between
                        # barriers, the max over ranks: what a LATER failure (a peer-window probe, the rebuild) still reports
                        # is this measurement, same K and W as asked for, not a probe of another length
                        dp.step_n(replay.handle, W, B, seed=0)
                        barrier()
                        tf0 = time.perf_counter()
                        dp.step_n(replay.handle, K, B, seed=0)
                        barrier()
                        dtf = all_max(time.perf_counter() - tf0)
                        if dp.healthy() and rank == 0:
                            FALLBACK.update(value=round(world * K / dtf, 1), steps=K, warmup=W, ms_per_step=round(dtf / K * 1e3, 5),
                                            data_parallel_check={"exchange": "rccl", "replicas_identical": True, "finite": True,
                                                                 "probe_us_per_step": {"rccl": round(probes[0], 2)}},
                                            note=f"measured over RCCL with this run's own protocol ({W} warm-up + {K} timed updates, "
                                                 "barrier-bracketed, max over ranks) before the peer-window exchanges were probed: a "
                                                 "later phase of the run failed (see `error`)")
            if not ok and rank == 0:
                print(f"bench.py: exchange level {level} unusable on this node", file=sys.stderr)
            del dp, algo
        if not probes:
            raise SystemExit("bench.py: no usable gradient exchange")
        p2p_level = min(probes, key=probes.get)
        wd.kick("data-parallel rebuild + warm-up")
        algo, dp, ok = build(p2p_level)
        assert ok, "the probed exchange level failed on rebuild"
        learner = algo.learner
        dp.step_n(replay.handle, args.pre_warm, B, seed=0)
        dp.step_n(replay.handle, W, B, seed=0)
        use_p2p = p2p_level > 0

        def run(n):
            dp.step_n(replay.handle, n, B, seed=0)

    wd.kick("timed region")
    barrier()
    t0 = time.perf_counter()
    run(K)
    barrier()
    dt = time.perf_counter() - t0
    wd.kick("checks + instrumented pass")
    if dist is not None:
        dt = all_max(dt)
    value = world * K / dt
    dp_check = None
    if use_dp:
        # replicas must have stayed identical (every rank sums the gradients in the same order) and finite
        spread = dp.replica_checksum()                       # collective: [max - min] over ranks of two checksums
        finite = bool(t.isfinite(algo.actor._oprl_arena).all() and t.isfinite(algo.critic._oprl_arena).all())
        dp_check = {"replicas_identical": bool(float(spread.abs().max()) == 0.0), "finite": finite,
                    "exchange": {2: "p2p-inline", 1: "p2p", 0: "rccl"}[p2p_level],
                    "probe_us_per_step": {({2: "p2p-inline", 1: "p2p", 0: "rccl"}[k]): round(v, 2) for k, v in probes.items()},
                    # (the launch form of a rank whose exchange runs inside the dW tiles: 4 = the single-GPU whole-update launch)
                    "inline_form": learner.debug_form(B)["dp_inline_form"]}

    out = None
    if rank == 0:
        # ---- instrumented pass: per-kernel durations from HIP events on the stream
        roof = None
        if True:
            P = max(1, min(args.profile_steps, K))
            prof_learner, us_per_step_plain = learner, 1e6 * dt / K
            if use_dp:
                # the kernels of a data-parallel rank are those of the single-GPU learner (plus the exchange
                # inside / after the dW launches): the roofline pass runs on a plain replica on rank 0
                t.manual_seed(0)
                plain = DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device=f"cuda:{local_rank}",
                             max_batch=B, precision=args.precision).create()
                prof_learner = plain.learner
                prof_learner.step_n(replay.handle, 300, B, seed=1)
                t.cuda.synchronize(dev)
                tq0 = time.perf_counter()
                prof_learner.step_n(replay.handle, P, B, seed=1)
                t.cuda.synchronize(dev)
                us_per_step_plain = 1e6 * (time.perf_counter() - tq0) / P
            lib.oprl_profile_enable(1)
            prof_learner.step_n(replay.handle, P, B, seed=1)
            NK = 6
            cnt = (C.c_int64 * NK)()
            ms = (C.c_double * NK)()
            _capi.check(lib.oprl_profile_read(cnt, ms, 1))
            lib.oprl_profile_enable(0)
            names = ["k_mlp_slice", "k_dw_adam", "k_replay_gather", "other", "k_ddpg_phase1", "k_ddpg_phase2"]
            # An event pair reads (kernel + the closing event packet).  In the timed loop above the
            # same launches run back to back, so one step = the sum of their durations; what the
            # instrumented pass reads in excess, spread over its launches, is the per-launch event
            # overhead (~2.4 us), and is subtracted.
            raw_us_per_step = sum(ms[i] for i in range(NK)) * 1e3 / P
            launches_per_step = sum(cnt[i] for i in range(NK)) / P
            ev_us = max(raw_us_per_step - us_per_step_plain, 0.0) / max(launches_per_step, 1.0)
            kern = {names[i]: dict(launches_per_step=cnt[i] / P,
                                   us_per_launch=max(ms[i] * 1e3 / cnt[i] - ev_us, 0.0),
                                   us_per_launch_raw=ms[i] * 1e3 / cnt[i],
                                   us_per_step=max(ms[i] * 1e3 / cnt[i] - ev_us, 0.0) * cnt[i] / P)
                    for i in range(NK) if cnt[i]}
            # which launch structure ran (csrc/learner.hip): whole updates per launch (k_ddpg_chain — counted in phase 1's
            # slot; a launch that holds ONE update is labelled k_ddpg_update below), the merged launches (phase 1 + the critic's tiles | phase 2 [+ the actor's
            # tiles]), or the plain sequence
            whole = "k_ddpg_phase1" in kern and "k_ddpg_phase2" not in kern and "k_dw_adam" not in kern
            # ... or SEVERAL updates per launch (k_ddpg_chain: step_n's K-loop inside the launch, up to 32 updates each)
            upl = 1.0
            if whole:
                upl = 1.0 / max(kern["k_ddpg_phase1"]["launches_per_step"], 1e-9)
                kern["k_ddpg_chain" if upl > 1.01 else "k_ddpg_update"] = kern.pop("k_ddpg_phase1")
            dom = ("k_ddpg_chain" if upl > 1.01 else "k_ddpg_update") if whole else ("k_ddpg_phase1" if "k_ddpg_phase1" in kern else "k_mlp_slice")
            merged = dom == "k_ddpg_phase1" and kern.get("k_dw_adam", {}).get("launches_per_step", 2.0) < 1.5
            if whole:
                # (per LAUNCH, as the contract asks: one update's algorithmic work x the updates one launch runs)
                macs = (MACS_SLICE + MACS_DW) * upl
                nbytes = (STATE_BYTES + 4 * B * (2 * S + A + 2)) * upl
            elif dom == "k_ddpg_phase1":
                macs = MACS_P1 + (F_CRITIC if merged else 0)
                nbytes = (32 * (F_CRITIC + HID * 2 + 1) if merged else 0) + 4 * B * (2 * S + A + 2)
            else:
                macs = MACS_SLICE / kern[dom]["launches_per_step"]
                nbytes = 0
            flop_per_launch = 2.0 * B * macs
            if kern[dom]["us_per_launch"] <= 0.0:          # (a very short run: the overhead estimate swallowed the reading)
                kern[dom]["us_per_launch"] = kern[dom]["us_per_launch_raw"]
            dur = kern[dom]["us_per_launch"] * 1e-6
            peak = PEAK_OF[args.precision]
            ach_flops, ach_bytes = flop_per_launch / dur / 1e12, nbytes / dur / 1e9
            # the binding roof is the one whose ideal time for this launch is longer
            t_mfma, t_hbm = flop_per_launch / (peak * 1e12), nbytes / (PEAK_HBM_TBS * 1e12)
            hbm_bound = t_hbm > t_mfma
            # HBM bytes and matrix-core utilisation are NOT measured in this run: they are read from the committed rocprofv3
            # --pmc passes of this same command (the builder's run: tools/profile_round.sh) and labelled as such;
            # `traffic` is per launch like `achieved` — the file's per-update figure x the updates this run's launches held
            traffic = mfma_util = traffic_per_update = None
            counters_source = None
            try:
                pmc = json.load(open(ROOT / "profiles" / PMC_FILE[args.precision]))
                pk = next((k for k in (dom, dom + "_dw") if k in pmc), dom)
                if pk in pmc:
                    per_upd = pmc[pk].get("hbm_bytes_per_update", pmc[pk].get("hbm_bytes_per_launch"))
                    traffic_per_update = per_upd
                    traffic = int(per_upd * upl) if per_upd is not None else None
                    mfma_util = pmc[pk].get("mfma_util")
                    counters_source = (f"profiles/{PMC_FILE[args.precision]} (builder's rocprofv3 --pmc passes of this command, "
                                       "tools/profile_round.sh; not measured in this run)")
            except Exception:  # noqa: BLE001
                pass
            arith = {"f32": "exact-fp32 v_mfma_f32_16x16x4_f32", "bf16": "v_mfma_f32_16x16x32_bf16, fp32 accumulate",
                     "x2": "v_mfma_f32_16x16x32_f16 x 3 per product (fp16 hi + lo operands), fp32 accumulate"}[args.precision]
            what = {"k_ddpg_update": "k_ddpg_update (the WHOLE update as one launch: target chain, critic forward / backward, "
                                     "the critic's dW + Adam + Polyak tiles, critic pass for the actor loss, the actor's "
                                     "backward and dW + Adam + Polyak tiles, next batch's gather; ",
                    "k_ddpg_chain": f"k_ddpg_chain ({upl:.1f} WHOLE updates per launch — step_n's K-loop inside the launch: per "
                                    "update the target chain, critic forward / backward, the critic's dW + Adam + Polyak tiles, "
                                    "the critic pass for the actor loss, the actor's unit-seed backward and dW + Adam + Polyak "
                                    "tiles, the next batch's gather; an update's roles start behind the flags of the one before; ",
                    "k_ddpg_phase1": ("k_ddpg_phase1_dw (phase 1 + the critic's dW / Adam tiles in one launch; " if merged
                                      else "k_ddpg_phase1<256> ("),
                    "k_mlp_slice": "k_mlp_slice ("}[dom]
            roof = dict(bound="hbm" if hbm_bound else "mfma", kernel=what + arith + ")",
                        achieved=round(ach_bytes if hbm_bound else ach_flops, 3),
                        peak=PEAK_HBM_TBS * 1e3 if hbm_bound else round(peak, 1), unit="GB/s" if hbm_bound else "TFLOP/s",
                        frac=round((t_hbm if hbm_bound else t_mfma) / dur, 5), traffic=traffic, mfma_util=mfma_util,
                        traffic_per_update=traffic_per_update, counters_source=counters_source,
                        updates_per_launch=round(upl, 2), us_per_update=round(dur * 1e6 / upl, 3),
                        flop_per_launch=flop_per_launch, bytes_per_launch=nbytes,
                        other_roof=dict(bound="mfma" if hbm_bound else "hbm",
                                        achieved=round(ach_flops if hbm_bound else ach_bytes, 3),
                                        peak=round(peak, 1) if hbm_bound else PEAK_HBM_TBS * 1e3,
                                        unit="TFLOP/s" if hbm_bound else "GB/s",
                                        frac=round((t_mfma if hbm_bound else t_hbm) / dur, 5)),
                        kernels=kern, event_overhead_us=round(ev_us, 3),
                        measured_on=("the timed learner" if not use_dp else
                                     "a single-GPU replica of the same kernels on rank 0 "
                                     f"({us_per_step_plain:.1f} us per update without the exchange)"),
                        note="durations from hipEvent pairs around each launch on the launch stream "
                             f"(serialised pass of {P} steps) minus event_overhead_us, the per-launch "
                             "excess of that pass over the un-instrumented timed loop (where the same "
                             "launches run back to back, so a step is the sum of their durations); they "
                             "agree with rocprofv3 --kernel-trace --stats (profiles/r06_kernel_stats_*.csv); "
                             "sum of kernel time per step = "
                             f"{sum(k['us_per_step'] for k in kern.values()):.1f} us; bytes_per_launch = 32 B per trained "
                             "parameter (theta, m, v, theta_target read + written) + the gathered minibatch rows; "
                             "flop_per_launch = 2 x B x the algorithmic MACs of the launch (SURVEY.md 8d); the x2 peak is "
                             "the dense fp16 MFMA peak / 3 (three hardware products per algorithmic one); traffic = "
                             "(2*FETCH_SIZE + WRITE_SIZE) KB and mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (average launch "
                             f"duration x 2.4 GHz x 1024 SIMDs) from profiles/{PMC_FILE[args.precision]} (separate --pmc "
                             "passes, tools/profile_round.sh).  Neither roof binds: an update is a CHAIN of dependent "
                             "16-row-slice stages (target chain -> TD seeds -> critic tiles -> critic pass -> du -> actor "
                             "tiles) whose length is set by cross-workgroup flag hops, cold weight-shard loads and "
                             "workgroup dispatch, not by bandwidth or the matrix cores (DESIGN.md section 6: the floor table; timeline in "
                             "profiles/r06_stage_stamps_f32.txt)")
        multi = None
        group = None
        if not use_dp and args.learners > 1:
            wd.kick("multi-learner + packed group")
            multi = multi_learner(args.learners, dev, local_rank, steps=max(200, min(K, 2000)))
            group = packed_group(args.group, dev, local_rank, steps=max(200, min(K, 1000)))
        configs = bf16 = api = exact = dp1 = x2blk = None
        if not use_dp and not args.no_configs:
            cache = {(S, A): replay}

            def replays(S_, A_):
                # (one extra replay resident at a time: the humanoid one is 0.5 GB)
                for k in [k for k in cache if k != (S, A) and k != (S_, A_)]:
                    del cache[k]
                if (S_, A_) not in cache:
                    cache[(S_, A_)] = make_replay(dev, seed=rank, S=S_, A=A_)
                return cache[(S_, A_)]
            wd.kick("config table")
            configs = config_table(dev, replays, n_steps=args.config_steps)
            d16 = next(r for r in configs if r["name"].startswith("DDPG") and r["dtype"] == "bf16")
            bf16 = dict(value=d16["steps_per_s"], unit="steps/s", us_per_step=d16["us_per_step"],
                        roofline_frac=d16["roofline_frac"], roof=d16["roof"], peak_tflops=PEAK_BF16_MATRIX_TFLOPS,
                        q_rel_dev_vs_f32_after_10_updates=round(bf16_q_deviation(dev, replay), 6),
                        note="OPRL_PREC_BF16: v_mfma_f32_16x16x32_bf16, fp32 accumulate / master / Adam; the "
                             "headline `value` above is exact fp32")
            api = api_rate(dev, replay, args.precision)
            d32 = next(r for r in configs if r["name"].startswith("DDPG") and r["dtype"] == "f32")
            exact = dict(value=d32["steps_per_s"], unit="steps/s", us_per_step=d32["us_per_step"],
                         note="OPRL_PREC_F32: exact-fp32 MFMA (v_mfma_f32_16x16x4_f32), the headline's mode, from the config table's "
                              f"{d32['steps']}-update run")
            dx2 = next(r for r in configs if r["name"].startswith("DDPG") and r["dtype"] == "x2")
            x2blk = dict(value=dx2["steps_per_s"], unit="steps/s", us_per_step=dx2["us_per_step"],
                         roofline_frac=dx2["roofline_frac"], roof=dx2["roof"],
                         note="OPRL_PREC_X2: fp32 operands as fp16 hi + lo, three fp16 MFMAs per product, fp32 accumulate / master / "
                              "Adam; a parity mode (outputs 2e-5, parameter digests 1e-4 against the reference's golden vectors; "
                              "Adam-moment digests 5e-3: one ReLU flip of one row), finite range |x| < 4094 (never silent)")
            wd.kick("single-rank data-parallel rate")
            try:
                dp1 = dp_single_rank(dev, local_rank, replay, args.precision, max(200, min(K, 3000)))
            except Exception as exc:  # noqa: BLE001
                dp1 = dict(value=None, error=f"{type(exc).__name__}: {exc}"[:300])
        wd.kick("cpu baseline")
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()   # rank 0 at N = 1 only
        out = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_OF[args.precision], "precision_mode": args.precision, "data": "synthetic", "pre_warm_steps": args.pre_warm,
            "config": {"workload": f"DDPG walker-walk dims S={S} A={A} B={B}, hidden (256,256), replay "
                                   f"{E}x{L} transitions resident in HBM, device-side uniform sampling, "
                                   + {"f32": "exact-fp32 MFMA (OPRL_PREC_F32, a parity mode)",
                                      "x2": "fp32 operands as fp16 hi + lo, three fp16 MFMAs per product, fp32 accumulate / "
                                            "master / Adam (OPRL_PREC_X2, a parity mode: the exact-fp32 mode's gates against the "
                                            "reference's golden vectors - outputs 2e-5, parameter digests 1e-4 - except the Adam-moment "
                                            "digests at 5e-3: one ReLU flip of one row; tests/test_gpu_x2.py, tests/scenarios.py)",
                                      "bf16": "bf16 MFMA inputs, fp32 accumulate / master / Adam (OPRL_PREC_BF16, NOT a "
                                              "parity mode)"}[args.precision],
                       "path": "oprl_learner_step_n" if not use_dp else
                               ("oprl_learner_dp_step_n: update_phase/apply + 2 gradient all-reduces (critic, actor) per step, all in C; "
                                + {2: "all-reduced per tile inside the dW + Adam launches over xGMI peer windows (csrc/p2p.hip, k_dw_adam<true>)",
                                   1: "one-shot peer-window all-reduce over xGMI, one kernel per exchange (csrc/p2p.hip)",
                                   0: "RCCL ncclAllReduce"}[p2p_level]),
                       "parallelism": f"dp{world}", "global_batch": B * world},
            "roofline": roof, "cpu_baseline": cpu, "configs": configs, "exact_f32": exact, "x2": x2blk, "bf16": bf16, "api_rate": api,
            "dp_single_rank": dp1,
            "multi_learner": multi, "packed_group": group, "data_parallel_check": dp_check,
            "rehearsal": ("every rank on GPU 0, gloo rendezvous, peer windows only, B = %d, one update per launch: a code-path "
                          "rehearsal (tests/test_gpu_p2p.py), NOT a measurement" % B) if rehearsal else None,
            "flop_per_step": 2.0 * B * (MACS_SLICE + MACS_DW), "state_bytes_per_step": STATE_BYTES,
        }
        sys.stderr.flush()
        try:        # (RCCL prints its version banner through C stdio: out before the line, so that the line is the last one)
            C.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    wd.teardown(60.0 if rank == 0 else 600.0)      # (the other ranks wait at the barrier while rank 0 runs its extra passes)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
