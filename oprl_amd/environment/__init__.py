"""Environment factory (reference: /root/reference/src/oprl/environment/)."""
from oprl_amd.environment.protocols import EnvProtocol
from oprl_amd.environment.synthetic import DM_CONTROL_DIMS, SyntheticEnv
from oprl_amd.environment.make_env import make_env   # (rebinds the name from the submodule to the function)

__all__ = ["EnvProtocol", "SyntheticEnv", "make_env", "DM_CONTROL_DIMS"]
