"""``make_env(name, seed)`` (reference: /root/reference/src/oprl/environment/make_env.py).

The reference dispatches a task name to its dm_control / gymnasium / safety-gymnasium wrappers; those
simulators are CPU physics outside the learner hot path and are not part of this
build.  What this factory serves is the synthetic stand-in of synthetic.py, and it says so:

* ``synthetic:<task>`` (e.g. ``synthetic:walker-walk``) is the explicit spelling;
* a bare dm_control task name is accepted so that the reference's config scripts and command lines keep
  running, but a warning names the substitution and the environment reports ``env_family == "synthetic"``
  (never "dm_control"), which runners log;
* any other name (gymnasium ``Ant-v4``, ``Safety*``) raises: there is no stand-in with made-up dims."""
import logging

from oprl_amd.environment.protocols import EnvProtocol
from oprl_amd.environment.synthetic import DM_CONTROL_DIMS, SyntheticEnv

PREFIX = "synthetic:"
_warned: set[str] = set()


def make_env(name: str, seed: int = 0) -> EnvProtocol:
    if name.startswith(PREFIX):
        return SyntheticEnv(name[len(PREFIX):], seed=seed)
    if name in DM_CONTROL_DIMS:
        if name not in _warned:
            _warned.add(name)
            logging.warning("make_env(%r): the dm_control simulator is not part of this build; serving the "
                            "synthetic linear-dynamics stand-in with its observation/action dims "
                            "(spell it %r to say so explicitly)", name, PREFIX + name)
        return SyntheticEnv(name, seed=seed)
    raise ValueError(f"unknown env {name!r}: this build serves {PREFIX}<task> for the dm_control tasks "
                     f"{sorted(DM_CONTROL_DIMS)}; gymnasium / safety-gymnasium wrappers are out of scope")
