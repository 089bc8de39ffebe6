"""Debug tool: per-phase timestamps inside k_mlp_slice for one DDPG update
(needs a GPU).  Prints, per launch, the phase durations of workgroup 0 and the
spread over workgroups, in shader cycles and in microseconds."""
import os
import sys
from pathlib import Path
os.environ.setdefault("OPRL_AMD_TRACE", "1")      # the library build with the stamps compiled in (python -m oprl_amd.build --trace)
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch as t
from oprl_amd import _capi
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.logging import NullLogger

S, A, B = 24, 6, 256
t.manual_seed(0)
algo = DDPG(logger=NullLogger(), state_dim=S, action_dim=A, device="cuda", max_batch=B,
            no_fuse="--generic" in sys.argv, precision="bf16" if "--bf16" in sys.argv else ("x2" if "--x2" in sys.argv else "f32")).create()
L = algo.learner
for a in sys.argv:             # --cluster=1: the single-CU-per-slice passes (what packed exact-fp32 learners run)
    if a.startswith("--cluster="):
        L.set_cluster(int(a.split("=")[1]))
NS, NST = 24, 24
buf = t.zeros((NS, 64, NST, 2), dtype=t.int64, device="cuda")
batch = [t.randn(B, S, device="cuda"), t.rand(B, A, device="cuda") * 2 - 1, t.rand(B, 1, device="cuda"),
         t.zeros(B, 1, device="cuda"), t.randn(B, S, device="cuda")]
for _ in range(50):
    algo.update(*batch)
_capi.check(L.lib.oprl_learner_set_trace(L.handle, _capi.ptr(buf)))
replay = None
if "--step-n" in sys.argv:     # the benchmarked mode: in-kernel replay gather
    import bench
    replay = bench.make_replay(t.device("cuda"), seed=0)
    L.step_n(replay.handle, 50, B, seed=3)
for _ in range(3):
    buf.zero_()
    if replay is not None:
        n_chain = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--chain=")), 4)
        L.step_n(replay.handle, n_chain, B, seed=5)     # the stamps left are the LAST update's: rows staged by its predecessor (k_ddpg_chain: the last update of the launch)
    else:
        algo.update(*batch)
t.cuda.synchronize()
tr = buf.cpu().numpy()
fused = "--generic" not in sys.argv
names = (["P1 role A target chain", "P1 role B critic f+b", "P1 role C actor fwd", "P2 critic f+b, actor bwd",
          "dW critic (wg0 = hidden-layer tile 0)", "dW actor", "P2 role U (actor unit backward)"]
         if fused else ["actor_t fwd", "critic_t fwd", "critic fwd+bwd", "actor fwd", "critic(s,pi) fwd+bwd", "actor bwd"])
for slot in range(len(names)):
    x = tr[slot, :16]                       # 16 workgroups
    if slot in (4, 5) and fused:
        x = tr[slot, 16:32]                 # dW: the hidden layer's first 16 tiles
    n = int((x[0, :, 0] != 0).sum())
    if n < 2:
        continue
    cyc = x[:, :n, 0].astype(np.float64)
    rt = x[:, :n, 1].astype(np.float64)     # 100 MHz ticks
    d_cyc = np.diff(cyc, axis=1)
    d_us = np.diff(rt, axis=1) / 100.0
    tot_us = (rt[:, -1] - rt[:, 0]) / 100.0
    clk = (cyc[:, -1] - cyc[:, 0]) / np.maximum(tot_us, 1e-9) / 1e3
    print(f"{names[slot]:26s} stamps={n} wg0 phases(us)={np.round(d_us[0], 2).tolist()} total wg0={tot_us[0]:.2f}us "
          f"max-wg={tot_us.max():.2f}us  cyc/us~{clk.mean():.2f} GHz  start spread={(rt[:,0].max()-rt[:,0].min())/100:.2f}us")
    if "--cycles" in sys.argv:
        print(f"{'':22s} wg0 phases(cycles)={d_cyc[0].astype(int).tolist()}")
    if slot in (4, 5) and fused and "--dw-tiles" in sys.argv:
        # the traced workgroups of a dW launch are the first 16 tiles of each layer (slot index = 16 * layer + tile)
        for layer in range(3):
            y = tr[slot, 16 * layer:16 * layer + 16]
            y = y[y[:, 0, 1] != 0]
            if len(y) == 0:
                continue
            ph = np.diff(y[:, :n, 1].astype(np.float64), axis=1) / 100.0
            print(f"{'':10s} layer {layer}: {len(y)} tiles traced, phases mean(us)={np.round(ph.mean(0), 2).tolist()} "
                  f"max(us)={np.round(ph.max(0), 2).tolist()} total mean={ph.sum(1).mean():.2f} max={ph.sum(1).max():.2f}")

# absolute timeline on the shared 100 MHz clock: first/last stamp over all traced workgroups
if fused:
    print("timeline (us from the first stamp of the update; first-in .. last-out over traced workgroups):")
    spans = []
    for slot in range(len(names)):
        x = tr[slot]
        ok = x[:, 0, 1] != 0
        if not ok.any():
            continue
        firsts = x[ok, 0, 1].astype(np.float64)
        lasts = np.array([row[row[:, 1] != 0, 1].max() for row in x[ok]], dtype=np.float64)
        spans.append((names[slot], firsts.min(), firsts.max(), lasts.min(), lasts.max()))
    t0 = min(sp[1] for sp in spans)
    for nm, f0, f1, l0, l1 in spans:
        print(f"  {nm:34s} start {(f0-t0)/100:7.2f} .. {(f1-t0)/100:7.2f}   end {(l0-t0)/100:7.2f} .. {(l1-t0)/100:7.2f}")

    if "--per-wg" in sys.argv:
        for slot in range(len(names)):
            x = tr[slot]
            ok = np.nonzero(x[:, 0, 1] != 0)[0]
            st_ = [(int(w), round((x[w, 0, 1] - t0) / 100, 2),
                    round((x[w][x[w][:, 1] != 0, 1].max() - t0) / 100, 2)) for w in ok]
            print(f"  {names[slot]}: (wg, start, end) {st_}")

    if "--per-wave" in sys.argv:
        # slice 0, all 16 waves (wave 0 = slot 0, wave w = slot 16 + w): time of every stamp, us from
        # that role's first stamp
        for slot in range(4):
            x = tr[slot]
            rows = [x[0]] + [x[16 + w] for w in range(1, 16)]
            base = min(r[0, 1] for r in rows if r[0, 1] != 0)
            n = int((x[0][:, 1] != 0).sum())
            print(f"  {names[slot]} — slice 0, stamp times per wave (us):")
            for w, r in enumerate(rows):
                print(f"    wave {w:2d}: " + " ".join(f"{(r[k, 1] - base) / 100:6.2f}" if r[k, 1] != 0 else "   -  " for k in range(n)))
