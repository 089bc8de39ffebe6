"""Aggregate steps/s of n packed DDPG learners (bench.multi_learner) for a few (n, precision) settings;
cluster size via OPRL_AMD_CLUSTER in the environment."""
import functools
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench
import oprl_amd.algos.ddpg as ddpg_mod

dev = t.device("cuda", 0)
orig = ddpg_mod.DDPG
for prec in sys.argv[2].split(","):
    ddpg_mod.DDPG = functools.partial(orig, precision=prec)
    for n in [int(x) for x in sys.argv[1].split(",")]:
        r = bench.multi_learner(n, dev, 0, steps=1000)
        print(f"cluster={os.environ.get('OPRL_AMD_CLUSTER', '4')} prec={prec} learners={n}: {r['value']:.0f} steps/s aggregate "
              f"({r['per_learner']:.0f} each) verified={r['verified']}", flush=True)
