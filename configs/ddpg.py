"""DDPG on the MI355X learner — the counterpart of the reference's configs/ddpg.py (which also runs against
this repo unchanged through the ``oprl`` alias package).

    python configs/ddpg.py --env walker-walk --device cuda [--seeds N]
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from _common import TrainingScript  # noqa: E402
from oprl.algos.ddpg import DDPG  # noqa: E402

script = TrainingScript(DDPG, "DDPG", estimate_q_every=5000, log_every=2500)
# the names a reference-style script defines at module level
make_env, make_algo, make_replay_buffer, make_logger, config = (
    script.make_env, script.make_algo, script.make_replay_buffer, script.make_logger, script.config)

if __name__ == "__main__":
    script.run()
