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

BATCHES = (1, 8, 100, 128, 256, 512, 1024)      # (128: the batch the reference's scripts train at, trainers/base_trainer.py:28)
FIELDS = ["fused", "lean", "form", "updates_per_chain_launch", "wide", "nc", "twin_split", "p2_pair", "arith", "xcd_local",
          "shared_chip", "dp_inline_form"]
ALGOS = {"DDPG": (24, 6, {}), "TD3": (17, 6, {}), "SAC": (24, 6, {}), "TQC": (24, 6, {})}
VARIANTS = [("plain", {}, {}, None), ("export_grads", dict(export_grads=True), {}, None), ("cluster4", {}, {}, 4),
            ("FORM=two", {}, {"OPRL_AMD_FORM": "two"}, None), ("FORM=plain", {}, {"OPRL_AMD_FORM": "plain"}, None),
            ("NO_WIDE", {}, {"OPRL_AMD_NO_WIDE": "1"}, None), ("NO_XCD_LOCAL", {}, {"OPRL_AMD_NO_XCD_LOCAL": "1"}, None),
            ("CHAIN=1", {}, {"OPRL_AMD_CHAIN": "1"}, None)]


def form_rows(algo_name, prec, variant):
    name, kw, env, cluster = variant
    S, A, extras = ALGOS[algo_name]
    # (the switches every variant owns: a value exported by the caller must neither leak into a variant nor be lost)
    switches = ("OPRL_AMD_FORM", "OPRL_AMD_NO_WIDE", "OPRL_AMD_NO_XCD_LOCAL", "OPRL_AMD_CHAIN")
    saved = {k: os.environ.pop(k) for k in switches if k in os.environ}
    for k, v in env.items():
        os.environ[k] = v
    try:
        t.manual_seed(0)
        algo = bench._make_algo(algo_name, S, A, max(BATCHES), dict(extras, **kw), t.device("cuda", 0), prec)
    finally:
        for k in env:
            os.environ.pop(k, None)
        os.environ.update(saved)
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


def write_json(path, note):
    """The table as tests/golden/launch_forms.json holds it (run on an MI355X: `python tools/form_table.py --json <path> [note]`)."""
    import json
    rows = []
    for a in ALGOS:
        for prec in ("f32", "x2", "bf16"):
            for v in VARIANTS:
                if a == "TQC" and v[0] not in ("plain",):
                    continue
                forms = form_rows(a, prec, v)
                rows.append(dict(algo=a, precision=prec, variant=v[0], forms={str(B): list(r) for B, r in forms.items()}))
    with open(path, "w") as f:
        f.write('{\n"fields": ' + json.dumps(FIELDS) + ',\n"generated_by": ' + json.dumps(note) + ',\n"rows": [\n')
        f.write(",\n".join(json.dumps(r) for r in rows))
        f.write("\n]\n}\n")
    return len(rows)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--json":
        n = write_json(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "tools/form_table.py on MI355X (256 compute units)")
        print(f"{n} rows -> {sys.argv[2]}")
        sys.exit(0)
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
