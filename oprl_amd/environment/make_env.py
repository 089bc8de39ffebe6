"""``oprl.environment.make_env`` as a module path (reference: environment/make_env.py): the
function lives in the package's ``__init__``."""
from oprl_amd.environment import make_env  # noqa: F401

__all__ = ["make_env"]
