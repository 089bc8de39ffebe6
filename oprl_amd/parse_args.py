"""Command line of the config scripts.  The flags and their defaults are the reference's
(src/oprl/parse_args.py) — scripts written against it parse the same way — except ``--device``, which
defaults to the only kind of device this learner runs on."""
from __future__ import annotations

import argparse

# (flag, type, default, help) — shared by both entry points
_SHARED = (
    ("--config", str, None, "path of a config file (accepted for compatibility; the scripts are the config)"),
    ("--env", str, "cartpole-balance", "environment name, e.g. walker-walk"),
    ("--device", str, "cuda", "device of the learner (a ROCm GPU; there is no CPU path)"),
    # extension: the learner's arithmetic mode.  Scripts and algorithm classes alike default to
    # exact fp32 — the reference's arithmetic, no input range.  "x2" is the faster parity mode, opt-in: it has a finite
    # range (|observation|, |hidden activation| < 4094, |w| < 256; leaving it raises from update() / check(), it is never
    # silent) — meant for normalised observations
    ("--precision", str, "f32", "f32 (exact fp32 MFMA, the default) | x2 (fp32 as fp16 hi + lo on the matrix cores: parity mode, |obs| < 4094) | bf16"),
)
_SINGLE = (
    ("--seeds", int, 1, "how many seeds to train, one process each"),
    ("--start_seed", int, 0, "first seed; the others count up from it"),
)
_DISTRIB = (
    ("--seed", int, 0, "random seed of the run"),
    # extension (BASELINE.json config 5): > 1 = that many data-parallel learner processes, one per GPU,
    # fed over shared-memory rings (runners/train_distrib.py::run_dp_training); 1 = the reference's layout
    ("--learners", int, 1, "data-parallel learner processes (one per GPU)"),
    ("--actors", int, 0, "CPU actor processes (0: the script's default)"),
)


def _parser(title: str, extra) -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description=title)
    for flag, kind, default, text in (*_SHARED, *extra):
        parser.add_argument(flag, type=kind, default=default, help=text)
    return parser


def parse_args() -> argparse.Namespace:
    """Flags of the single-process scripts (configs/ddpg.py ...)."""
    return _parser("Run training", _SINGLE).parse_args()


def parse_args_distrib() -> argparse.Namespace:
    """Flags of configs/distrib_ddpg.py."""
    return _parser("Run distrib training", _DISTRIB).parse_args()
