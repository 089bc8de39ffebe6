"""Launch-form selection held against a table.

Which kernels an update runs — whole updates per launch (k_ddpg_chain), the merged phase launches, the plain phase + dW
launches, the generic sequence; lean or generic passes; clusters of eight; XCD-local exchanges — is decided by a dozen
interacting conditions in csrc/learner.hip (ddpg_args, critic_phase).  oprl_learner_debug_form reports the decision as
twelve numbers; tests/golden/launch_forms.json (tools/form_table.py, MI355X) holds them for {DDPG, TD3, SAC, TQC} x
{f32, x2, bf16} x {plain, export_grads, set_cluster(4), five environment switches} x B in {1, 8, 100, 128, 256, 512, 1024}.
A mode that falls off its fast form is a red test here, not a line in a benchmark table.  Reference: none (the reference
has one path, autograd: /root/reference/src/oprl/algos/ddpg.py:61-107)."""
import json
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

TABLE = json.loads((Path(__file__).parent / "golden" / "launch_forms.json").read_text())
FIELDS = TABLE["fields"]
ROWS = TABLE["rows"]


def _form_table():
    import importlib.util
    spec = importlib.util.spec_from_file_location("form_table", Path(__file__).resolve().parents[1] / "tools" / "form_table.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _id(r):
    return f"{r['algo']}-{r['precision']}-{r['variant']}"


@pytest.mark.parametrize("row", ROWS, ids=[_id(r) for r in ROWS])
def test_launch_form_matches_the_table(row):
    import torch as t
    if t.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the table is for a 256-compute-unit MI355X")
    form_table = _form_table()
    variant = next(v for v in form_table.VARIANTS if v[0] == row["variant"])
    got = form_table.form_rows(row["algo"], row["precision"], variant)
    for B, want in row["forms"].items():
        have = list(got[int(B)])
        diff = {FIELDS[i]: (have[i], want[i]) for i in range(12) if have[i] != want[i]}
        assert not diff, f"{_id(row)} B={B}: (got, table) {diff}"


def test_the_headline_modes_take_the_whole_update_form():
    """The table's own sanity: every arithmetic of DDPG at B <= 256 — B = 128, the reference scripts' batch, included — runs
    whole updates, 32 per launch, lean passes, role A and the critic pass on clusters of eight; and a gradient-exporting
    DDPG learner (a data-parallel rank) takes the same form once its exchange runs inside the tiles (dp_inline_form)."""
    for r in ROWS:
        if r["algo"] == "DDPG" and r["variant"] == "plain":
            for B in ("1", "8", "100", "128", "256"):
                f = dict(zip(FIELDS, r["forms"][B]))
                assert (f["fused"], f["lean"], f["form"], f["updates_per_chain_launch"], f["wide"]) == (1, 1, 4, 32, 3), (r, B)
        if r["algo"] == "DDPG" and r["variant"] == "export_grads":
            for B in ("1", "8", "100", "128", "256"):
                f = dict(zip(FIELDS, r["forms"][B]))
                assert f["dp_inline_form"] == 4, (r, B)
