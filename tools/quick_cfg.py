"""Quick rate probe of a BASELINE.json configuration (needs a GPU): `quick_cfg.py CONFIG spec [spec ..]` with CONFIG in
{ddpg, ddpg128, td3, sac, tqc} and spec = `prec[:ENV=V[,ENV=V..]]` (environment switches are read at learner creation),
e.g. `quick_cfg.py sac x2 x2:OPRL_AMD_NO_SPLIT=1 f32 bf16`: us per update through step_n, finiteness and the error word."""
import os
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench

KEY = {"ddpg": "DDPG walker-walk B=256", "ddpg128": "DDPG walker-walk B=128 (the reference scripts' batch)",
       "td3": "TD3 cheetah-run B=256", "sac": "SAC humanoid-walk B=1024", "tqc": "TQC walker-walk B=256 5x25"}
cfg, _, b_over = sys.argv[1].partition("@")        # `td3@128`: the config's dims at another batch size
if cfg.startswith("ddpg") and cfg not in KEY:        # ddpg512, ddpg1024, ...: DDPG at walker dims and that batch size
    cls, S, A, B, extras = "DDPG", 24, 6, int(cfg[4:]), {}
else:
    cls, S, A, B, extras, _, _ = bench.BASELINE_CONFIGS[KEY[cfg]]
if b_over:
    B = int(b_over)
dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0, S=S, A=A)
n = 400 if cfg == "tqc" else 2000
for spec in sys.argv[2:] or ["x2"]:
    prec, _, envs = spec.partition(":")
    kv = [e.split("=") for e in envs.split(",") if e]
    for k, v in kv:
        os.environ[k] = v
    algo = bench._make_algo(cls, S, A, B, extras, dev, prec)
    for k, _ in kv:
        del os.environ[k]
    L = algo.learner
    L.step_n(replay.handle, n // 4, B, seed=0)
    t.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        L.step_n(replay.handle, n, B, seed=0)
        t.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    L.check()
    fin = all(bool(t.isfinite(getattr(algo, m)._oprl_arena).all()) for m in ("actor", "critic"))
    print(f"{sys.argv[1]:10s} {spec:44s} {best / n * 1e6:8.2f} us/update ({n / best / 1e3:6.2f}k/s)   finite={fin}", flush=True)
    del algo, L
