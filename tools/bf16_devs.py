"""Per-key deviations of the bf16 mode vs its CPU emulation and vs the fp32 golden vectors (debug aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from oracle import oprl_oracle as orc
from tests import hip_adapters as ha
from tests import scenarios as sc
from tests.test_gpu_bf16 import _bf16_updates

which = sys.argv[1] if len(sys.argv) > 1 else "ddpg"
if which == "ddpg":
    got = sc.ddpg_scenario(lambda *a: ha.HipDDPG(*a, precision="bf16"))
    class EmuD(_bf16_updates(sc.OracleDDPG)):
        def hook_step1(self):
            self._g = (self.o.last["g_critic"], self.o.last["g_actor"])
    emu = sc.ddpg_scenario(EmuD)
    gold = sc.load_golden("ddpg_walker_b256")
elif which == "td3":
    got = sc.td3_scenario(lambda *a: ha.HipTD3(*a, precision="bf16"))
    emu = sc.td3_scenario(_bf16_updates(sc.OracleTD3))
    gold = sc.load_golden("td3_cheetah_b256")
elif which == "tqc":
    got = sc.tqc_scenario(lambda *a: ha.HipTQC(*a, precision="bf16"))
    emu = sc.tqc_scenario(_bf16_updates(sc.OracleTQC, min_dim=512))
    gold = sc.load_golden("tqc_walker_b256")
else:
    got = sc.sac_scenario(lambda *a: ha.HipSAC(*a, precision="bf16"), "walker", 256, 350, True, 3)
    emu = sc.sac_scenario(_bf16_updates(sc.OracleSAC), "walker", 256, 350, True, 3)
    gold = sc.load_golden("sac_walker_tune_b256")
rows = []
for k, w in emu.items():
    if w.dtype.kind in "US" or k == "meta":
        continue
    rows.append((sc.rel_dev(got[k], w), sc.rel_dev(got[k], gold[k]) if k in gold else float("nan"), k))
for e, r, k in sorted(rows, key=lambda x: x[2]):
    print(f"{k:36s} vs emu {e:9.2e}   vs fp32 ref {r:9.2e}")
