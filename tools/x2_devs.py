"""Per-key deviation of a golden scenario for the f32 and x2 modes (debug tool)."""
import functools, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from tests import scenarios as sc
from tests import hip_adapters as ha

which = sys.argv[1] if len(sys.argv) > 1 else "ddpg"
gold = sc.load_golden({"ddpg": "ddpg_walker_b256", "td3": "td3_cheetah_b256"}[which])
scen = {"ddpg": sc.ddpg_scenario, "td3": sc.td3_scenario}[which]
cls = {"ddpg": ha.HipDDPG, "td3": ha.HipTD3}[which]
outs = {p: scen(functools.partial(cls, precision=p)) for p in ("f32", "x2")}
for k, w in gold.items():
    if k == "meta" or w.dtype.kind in "US":
        continue
    d = {p: sc.rel_dev(outs[p][k], w) for p in outs}
    flag = " <<<" if d["x2"] > 3 * max(d["f32"], 1e-7) else ""
    print(f"{k:32s} f32 {d['f32']:.2e}   x2 {d['x2']:.2e}{flag}")
