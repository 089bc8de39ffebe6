"""``make_env(name, seed)`` (reference: /root/reference/src/oprl/environment/make_env.py).
Real simulators are out of scope (CPU physics); see synthetic.py."""
from oprl_amd.environment.protocols import EnvProtocol
from oprl_amd.environment.synthetic import SyntheticEnv


def make_env(name: str, seed: int = 0) -> EnvProtocol:
    return SyntheticEnv(name, seed=seed)
