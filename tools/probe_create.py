import sys, time
sys.path.insert(0, "/root/repo")
import torch as t
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.algos.tqc import TQC
from oprl_amd.logging import NullLogger
for cls, kw in ((DDPG, dict(precision="f32")), (DDPG, dict(precision="x2")), (TQC, dict(precision="f32", log_every=10**9))):
    ts = []
    for i in range(6):
        t.cuda.synchronize()
        t0 = time.perf_counter()
        a = cls(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=256, **kw)
        t1 = time.perf_counter()
        a.create()
        t.cuda.synchronize()
        t2 = time.perf_counter()
        del a
        t3 = time.perf_counter()
        ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    print(cls.__name__, kw.get("precision"), "ctor / create / destroy ms:", [tuple(round(x, 1) for x in r) for r in ts[1:]], flush=True)
