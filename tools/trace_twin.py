"""(tools/trace_sac.py for either twin-critic config: `trace_twin.py PREC K td3` = TD3 cheetah B = 256)  Stage stamps of SAC's phase launches at B = 1024 (humanoid dims): per role, wave 0 of every slice's lead member
(needs a GPU and the trace build: python -m oprl_amd.build --trace)."""
import os
import sys
from pathlib import Path
os.environ.setdefault("OPRL_AMD_TRACE", "1")
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch as t
import bench
from oprl_amd import _capi

prec = sys.argv[1] if len(sys.argv) > 1 else "x2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S, A, B = (17, 6, 256) if (len(sys.argv) > 3 and sys.argv[3] == 'td3') else (67, 21, 1024)
dev = t.device("cuda", 0)
replay = bench.make_replay(dev, seed=0, S=S, A=A)
t.manual_seed(0)
algo = bench._make_algo('TD3' if B == 256 else 'SAC', S, A, B, {'log_every': 10 ** 9}, dev, prec)
L = algo.learner
NS, NST = 24, 24
buf = t.zeros((NS, 64, NST, 2), dtype=t.int64, device="cuda")
L.step_n(replay.handle, 50, B, seed=3)
_capi.check(L.lib.oprl_learner_set_trace(L.handle, _capi.ptr(buf)))
for _ in range(3):
    buf.zero_()
    L.step_n(replay.handle, K, B, seed=5)     # the stamps left are the last update's (K > 1: its rows were staged by the update before)
t.cuda.synchronize()
tr = buf.cpu().numpy()
names = {0: "P1 role A", 1: "P1 role B1", 2: "P1 role B2", 3: "P1 role C", 6: "P2"}
t00 = None
for slot, name in names.items():
    x = tr[slot][:B // 16]            # [slices][NST][2]
    n = int((x[0, :, 1] != 0).sum())
    if n < 2:
        continue
    rt = x[:, :n, 1].astype(np.float64) / 100.0     # us
    if t00 is None:
        t00 = min(tr[s_][:B // 16, 0, 1][tr[s_][:B // 16, 0, 1] != 0].min() for s_ in names if (tr[s_][:B // 16, 0, 1] != 0).any()) / 100.0
    ent = rt[:, 0] - t00
    tot = rt[:, -1] - rt[:, 0]
    d = np.diff(rt, axis=1)
    print(f"{name}: {n} stamps; entry (us after the launch's first stamp) min {ent.min():.2f} median {np.median(ent):.2f} max {ent.max():.2f}; "
          f"residency min {tot.min():.2f} median {np.median(tot):.2f} max {tot.max():.2f}")
    print("   stage medians:", " ".join(f"{v:.2f}" for v in np.median(d, axis=0)))
    print("   stage maxima: ", " ".join(f"{v:.2f}" for v in d.max(axis=0)))
for slot, name in ((4, "dW critics"), (5, "dW actor")):
    for item in range(3):
        x = tr[slot, 16 * item:16 * item + 16]
        n = int((x[0, :, 1] != 0).sum())
        if n < 2:
            continue
        rt = x[:, :n, 1].astype(np.float64) / 100.0
        d = np.diff(rt, axis=1)
        print(f"{name} item {item}: {n} stamps; entry after the update's first stamp {np.median(rt[:, 0]) - t00:.2f}; residency median {np.median(rt[:, -1] - rt[:, 0]):.2f} max {(rt[:, -1] - rt[:, 0]).max():.2f}")
        print("   stage medians:", " ".join(f"{v:.2f}" for v in np.median(d, axis=0)))
