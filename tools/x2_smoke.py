import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.logging import NullLogger
from oracle import fixtures as fx
t.manual_seed(0)
a = DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", precision="x2").create()
for step in range(3):
    batch = [x.cuda() for x in fx.make_batch(900 + step, 256, 24, 6)]
    a.update(*batch)
    t.cuda.synchronize()
    print("step", step, "ok", bool(t.isfinite(a.critic._oprl_arena).all()), bool(t.isfinite(a.actor._oprl_arena).all()), flush=True)
