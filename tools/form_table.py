"""Prints the launch-form table tests/test_gpu_forms.py holds (needs a GPU): for every (algorithm, precision, variant) the
twelve numbers of oprl_learner_debug_form at each batch size.  Variants: plain, export_grads, set_cluster(4) (a learner
that shares the chip), and a few environment switches."""
import ctypes as C
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench

BATCHES = (1, 8, 100, 256, 512, 1024)
ALGOS = {"DDPG": (24, 6, {}), "TD3": (17, 6, {}), "SAC": (24, 6, {}), "TQC": (24, 6, {})}
VARIANTS = [("plain", {}, {}, None), ("export_grads", dict(export_grads=True), {}, None), ("cluster4", {}, {}, 4),
            ("FORM=two", {}, {"OPRL_AMD_FORM": "two"}, None), ("FORM=plain", {}, {"OPRL_AMD_FORM": "plain"}, None),
            ("NO_WIDE", {}, {"OPRL_AMD_NO_WIDE": "1"}, None), ("NO_XCD_LOCAL", {}, {"OPRL_AMD_NO_XCD_LOCAL": "1"}, None),
            ("CHAIN=1", {}, {"OPRL_AMD_CHAIN": "1"}, None)]


def form_rows(algo_name, prec, variant):
    name, kw, env, cluster = variant
    S, A, extras = ALGOS[algo_name]
    for k, v in env.items():
        os.environ[k] = v
    try:
        t.manual_seed(0)
        algo = bench._make_algo(algo_name, S, A, max(BATCHES), dict(extras, **kw), t.device("cuda", 0), prec)
    finally:
        for k in env:
            del os.environ[k]
    L = algo.learner
    if cluster is not None:
        L.set_cluster(cluster)
    out = (C.c_int32 * 12)()
    rows = {}
    for B in BATCHES:
        rc = L.lib.oprl_learner_debug_form(L.handle, B, out)
        assert rc == 0
        rows[B] = tuple(int(x) for x in out)
    del algo, L
    return rows


if __name__ == "__main__":
    print("EXPECTED = {")
    for a in ALGOS:
        for prec in ("f32", "x2", "bf16"):
            for v in VARIANTS:
                if a == "TQC" and v[0] not in ("plain",):
                    continue
                try:
                    rows = form_rows(a, prec, v)
                except Exception as exc:  # noqa: BLE001
                    print(f"    # ({a}, {prec}, {v[0]}): {type(exc).__name__}: {str(exc)[:80]}")
                    continue
                print(f"    ({a!r}, {prec!r}, {v[0]!r}): {{")
                for B, r in rows.items():
                    print(f"        {B}: {r},")
                print("    },")
    print("}")
