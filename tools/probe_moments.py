"""Adam-moment digests of the scripted scenarios: HIP learner vs the CPU oracle vs the golden vectors, per key (needs a GPU)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from tests import scenarios as sc
from tests import hip_adapters as ha

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
kw = {} if prec == "f32" else {"precision": prec}
def n_over(a, b, tol=1e-4):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return int((np.abs(a - b) > tol * max(np.abs(b).max(), 1e-30)).sum())


cases = {
    "ddpg": (lambda: sc.ddpg_scenario(lambda *a: ha.HipDDPG(*a, **kw)), lambda: sc.ddpg_scenario(sc.OracleDDPG), "ddpg_walker_b256"),
    "td3": (lambda: sc.td3_scenario(lambda *a: ha.HipTD3(*a, **kw)), lambda: sc.td3_scenario(sc.OracleTD3), "td3_cheetah_b256"),
    "sac_tune": (lambda: sc.sac_scenario(lambda *a: ha.HipSAC(*a, **kw), "walker", 256, 350, True, 3),
                 lambda: sc.sac_scenario(sc.OracleSAC, "walker", 256, 350, True, 3), "sac_walker_tune_b256"),
    "sac_hum": (lambda: sc.sac_scenario(lambda *a: ha.HipSAC(*a, **kw), "humanoid", 1024, 300, False, 2),
                lambda: sc.sac_scenario(sc.OracleSAC, "humanoid", 1024, 300, False, 2), "sac_humanoid_b1024"),
    "tqc": (lambda: sc.tqc_scenario(lambda *a: ha.HipTQC(*a, **kw)), lambda: sc.tqc_scenario(sc.OracleTQC), "tqc_walker_b256"),
}
for name, (hip, ora, gold) in cases.items():
    got, want, g = hip(), ora(), sc.load_golden(gold)
    rows = {}
    for k in want:
        if ".sample" not in k or not any(w in k for w in (".m_", ".v_")):
            continue
        grp = k.split(".")[0] + "." + k.split(".")[1]
        d_o, d_g, o_g = sc.rel_dev(got[k], want[k]), sc.rel_dev(got[k], g[k]), sc.rel_dev(want[k], g[k])
        r = rows.setdefault(grp, [0.0, 0.0, 0.0, "", 0, 0, 0])
        if d_o > r[0]:
            r[0], r[3] = d_o, k
        r[1], r[2] = max(r[1], d_g), max(r[2], o_g)
        r[4] = max(r[4], n_over(got[k], want[k])); r[5] = max(r[5], n_over(got[k], g[k])); r[6] += got[k].size
    for grp, (d_o, d_g, o_g, k, no, ng, sz) in rows.items():
        print(f"{prec} {name:9s} {grp:18s} hip-oracle {d_o:.2e}  hip-golden {d_g:.2e}  oracle-golden {o_g:.2e}   elements over 1e-4 in the worst "
              f"key: {no} / {ng} (vs oracle / golden; {sz} sampled)   worst {k}", flush=True)
