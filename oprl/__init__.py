"""Drop-in alias: ``import oprl.…`` resolves to the MI355X-native ``oprl_amd``
modules, so the reference's config scripts (configs/ddpg.py etc., which import
``oprl.algos.ddpg``, ``oprl.buffers.episodic_buffer``, ``oprl.runners.train`` …)
run unchanged against this learner.  See INTEGRATION.md."""
import importlib as _il
import sys as _sys

_SUBMODULES = [
    "logging", "parse_args",
    "algos", "algos.protocols", "algos.base_algorithm", "algos.nn_models", "algos.nn_functions",
    "algos.ddpg", "algos.td3", "algos.sac", "algos.tqc",
    "buffers", "buffers.protocols", "buffers.episodic_buffer",
    "environment", "environment.protocols", "environment.make_env",
    "runners", "runners.config", "runners.train", "runners.train_distrib",
    "trainers", "trainers.protocols", "trainers.base_trainer",
    "distrib", "distrib.queue", "distrib.env_worker", "distrib.policy_update_worker",
]
for _name in _SUBMODULES:
    _mod = _il.import_module(f"oprl_amd.{_name}")
    _sys.modules[f"{__name__}.{_name}"] = _mod
    if "." not in _name:
        globals()[_name] = _mod
