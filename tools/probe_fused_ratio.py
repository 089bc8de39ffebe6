import os, sys
sys.path.insert(0, "/root/repo")
import torch as t
import tests.test_gpu_fused as tf
from tests import scenarios
fx = tf.fx
for cluster in (8, "8s", "8p", 4, "4g", 2, 1):
    for B in (256, 8, 100):
        os.environ["OPRL_AMD_NO_LEAN"] = "1" if cluster == "4g" else "0"
        os.environ["OPRL_AMD_NO_WIDE"] = "0" if cluster in (8, "8s", "8p") else "1"
        os.environ["OPRL_AMD_FORM"] = "plain" if cluster == "8s" else ("p2" if cluster == "8p" else "chain")
        os.environ["OPRL_AMD_CLUSTER"] = str(4 if cluster in ("4g", 8, "8s", "8p") else cluster)
        fused, generic = tf._ddpg(), tf._ddpg(no_fuse=True)
        for step in range(4):
            batch = [x.cuda() for x in fx.make_batch(70 + step, B, 24, 6)]
            fused.update(*batch); generic.update(*batch)
        t.cuda.synchronize()
        r = {}
        for m in ("actor", "critic"):
            a, b = getattr(fused, m)._oprl_arena, getattr(generic, m)._oprl_arena
            r[m] = float((a - b).abs().max() / b.abs().max())
        for w in ("actor_m", "critic_m"):
            a, b = getattr(fused.learner, w), getattr(generic.learner, w)
            r[w] = float((a - b).abs().max() / b.abs().max())
        print(cluster, B, {k: f"{v:.1e}" for k, v in r.items()}, flush=True)
