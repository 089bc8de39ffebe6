"""Where do the Adam moments of an x2 learner differ from the f32 learner's? (debug tool)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch as t
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.logging import NullLogger
from oracle import fixtures as fx

from tests import hip_adapters as ha
def make(p):
    if "--default-init" in sys.argv:
        t.manual_seed(0)
        return DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", precision=p).create()
    actor = fx.make_net(101, fx.actor_dims(24, 6))
    critic = fx.make_net(102, fx.critic_dims(24, 6))
    return ha.HipDDPG(24, 6, actor, critic, precision=p).algo
a, b = make("x2"), make("f32")
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10
for step in range(n_steps):
    batch = [x.cuda() for x in fx.make_batch(110 + step, 256, 24, 6)]
    a.update(*batch); b.update(*batch)
    t.cuda.synchronize()
    for name in ("critic_m", "actor_m"):
        ma, mb = getattr(a.learner, name).cpu().numpy(), getattr(b.learner, name).cpu().numpy()
        dims = [(256, 30), (256,), (256, 256), (256,), (1, 256), (1,)] if name == "critic_m" else [(256, 24), (256,), (256, 256), (256,), (6, 256), (6,)]
        off = 0
        line = []
        for i, d in enumerate(dims):
            n = int(np.prod(d))
            xa, xb = ma[off:off + n].reshape(d), mb[off:off + n].reshape(d)
            off += n
            diff = np.abs(xa - xb)
            idx = np.unravel_index(diff.argmax(), diff.shape)
            line.append(f"{i}:{diff.max() / np.abs(xb).max():.1e}@{idx}")
        print(f"step {step + 1} {name}: " + "  ".join(line))
