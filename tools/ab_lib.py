"""A/B of two builds of the library on one configuration (needs a GPU): `ab_lib.py CONFIG prec [prec ..]` runs the same loop in
two child processes — OPRL_AMD_LIB unset (the product) and = oprl_amd/lib/liboprl_amd_prev.so (a build of the sources before
a change, linked by hand) — and prints us per update and a checksum of the parameters after the same 300 updates: equal
checksums = the change left every bit where it was."""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CODE = r'''
import sys, time, hashlib
sys.path.insert(0, "%s")
import torch as t
import bench
KEY = {"ddpg": "DDPG walker-walk B=256", "td3": "TD3 cheetah-run B=256", "sac": "SAC humanoid-walk B=1024",
       "tqc": "TQC walker-walk B=256 5x25"}
cfg, prec = sys.argv[1], sys.argv[2]
cls, S, A, B, extras, _, _ = bench.BASELINE_CONFIGS[KEY[cfg]]
dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0, S=S, A=A)
t.manual_seed(0)
algo = bench._make_algo(cls, S, A, B, extras, dev, prec)
L = algo.learner
L.step_n(replay.handle, 300, B, seed=0)
t.cuda.synchronize()
h = hashlib.sha256()
for m in ("actor", "critic"):
    h.update(getattr(algo, m)._oprl_arena.cpu().numpy().tobytes())
n = 400 if cfg == "tqc" else 2000
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    L.step_n(replay.handle, n, B, seed=0)
    t.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
L.check()
print(f"{best / n * 1e6:8.2f} us/update   parameters after 300 updates: sha256 {h.hexdigest()[:16]}")
''' % str(ROOT)

cfg = sys.argv[1]
for prec in sys.argv[2:] or ["x2"]:
    for name, lib in (("product", None), ("previous", str(ROOT / "oprl_amd" / "lib" / "liboprl_amd_prev.so"))):
        env = dict(os.environ)
        if lib:
            env["OPRL_AMD_LIB"] = lib
        out = subprocess.run([sys.executable, "-c", CODE, cfg, prec], env=env, capture_output=True, text=True, timeout=900)
        last = (out.stdout.strip().splitlines() or [out.stderr[-400:]])[-1]
        print(f"{cfg} {prec:5s} {name:9s} {last}", flush=True)
