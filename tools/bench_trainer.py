"""Env steps/s of the single-process trainer loop (env step -> add_transition ->
sample -> update, one update per env step) on the synthetic walker-shaped env — the caller around
the hot path; the reference's loop on the same box is bound by its CPU update (~150-350 steps/s)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import cProfile
import pstats
import torch as t
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.algos.sac import SAC
from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
from oprl_amd.environment import make_env
from oprl_amd.logging import NullLogger
from oprl_amd.trainers.base_trainer import BaseTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
prec = next((a.split("=")[1] for a in sys.argv if a.startswith("--precision=")), "x2")
prof = "--profile" in sys.argv
for name, cls in (("DDPG", DDPG), ("SAC", SAC)):
    def env(seed):
        return make_env("walker-walk", seed=seed)
    e = env(0)
    S, A = e.observation_space.shape[0], e.action_space.shape[0]
    algo = cls(logger=NullLogger(), state_dim=S, action_dim=A, device="cuda", precision=prec).create()
    buf = EpisodicReplayBuffer(buffer_size_transitions=int(1e6), state_dim=S, action_dim=A, device="cuda").create()
    tr = BaseTrainer(logger=NullLogger("/tmp/oprl_amd_bench"), env=e, make_env_test=env, replay_buffer=buf, algo=algo,
                     num_steps=n, start_steps=1000, batch_size=256, eval_interval=10 ** 9, save_policy_every=0,
                     stdout_log_every=10 ** 9)
    t.cuda.synchronize()
    t0 = time.perf_counter()
    if prof:
        pr = cProfile.Profile(); pr.enable()
    tr.train()
    t.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof:
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
    print(f"{name} [{prec}]: trainer loop {n / dt:9.1f} env steps/s ({dt / n * 1e6:.1f} us per step, one update each)", flush=True)
