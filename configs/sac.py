"""SAC training script in the reference's config-script shape
(/root/reference/configs/sac.py): the same factories and the same
``run_training`` call, importing through the ``oprl`` alias package.  The
reference's own configs/sac.py runs against this repo unchanged; this copy only
exists so the repo is self-contained.

    python configs/sac.py --env walker-walk --device cuda
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from oprl.algos.sac import SAC  # noqa: E402
from oprl.algos.protocols import AlgorithmProtocol  # noqa: E402
from oprl.buffers.episodic_buffer import EpisodicReplayBuffer  # noqa: E402
from oprl.buffers.protocols import ReplayBufferProtocol  # noqa: E402
from oprl.environment import make_env as _make_env  # noqa: E402
from oprl.environment.protocols import EnvProtocol  # noqa: E402
from oprl.logging import LoggerProtocol, make_text_logger_func  # noqa: E402
from oprl.parse_args import parse_args  # noqa: E402
from oprl.runners.config import CommonParameters  # noqa: E402
from oprl.runners.train import run_training  # noqa: E402

args = parse_args()


def make_env(seed: int) -> EnvProtocol:
    return _make_env(args.env, seed=seed)


_probe = make_env(seed=0)
STATE_DIM: int = _probe.observation_space.shape[0]
ACTION_DIM: int = _probe.action_space.shape[0]

config = CommonParameters(state_dim=STATE_DIM, action_dim=ACTION_DIM, num_steps=int(100_000),
                          eval_every=2500, device=args.device, estimate_q_every=5000, log_every=1000)


def make_algo(logger: LoggerProtocol) -> AlgorithmProtocol:
    return SAC(state_dim=STATE_DIM, action_dim=ACTION_DIM, device=args.device, logger=logger).create()


def make_replay_buffer() -> ReplayBufferProtocol:
    return EpisodicReplayBuffer(buffer_size_transitions=max(config.num_steps, int(1e6)),
                                state_dim=STATE_DIM, action_dim=ACTION_DIM, device=config.device).create()


make_logger = make_text_logger_func(algo="SAC", env=args.env)

if __name__ == "__main__":
    run_training(make_algo=make_algo, make_env=make_env, make_replay_buffer=make_replay_buffer,
                 make_logger=make_logger, config=config, seeds=args.seeds, start_seed=args.start_seed)
