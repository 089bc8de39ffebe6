"""Command-line flags of the config scripts (reference:
/root/reference/src/oprl/parse_args.py).  Same flag names and defaults, except
that ``--device`` defaults to ``cuda``: this learner has no CPU path."""
from __future__ import annotations

import argparse


def _common(description: str) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=description)
    p.add_argument("--config", type=str, help="Path to the config file.")
    p.add_argument("--env", type=str, default="cartpole-balance", help="Name of the environment.")
    p.add_argument("--device", type=str, default="cuda", help="Device to perform training on.")
    return p


def parse_args() -> argparse.Namespace:
    p = _common("Run training")
    p.add_argument("--seeds", type=int, default=1,
                   help="Number of parallel processes launched with different random seeds.")
    p.add_argument("--start_seed", type=int, default=0,
                   help="Number of the first seed. Following seeds will be incremented from it.")
    return p.parse_args()


def parse_args_distrib() -> argparse.Namespace:
    p = _common("Run distrib training")
    p.add_argument("--seed", type=int, default=0, help="Random seed")
    return p.parse_args()
