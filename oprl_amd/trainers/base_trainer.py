"""Single-process training loop: env step -> buffer.add_transition ->
buffer.sample -> algo.update, once per env step — the caller of the hot path
(reference: /root/reference/src/oprl/trainers/base_trainer.py:38-120, row N1 of
SURVEY.md §8f).  Same fields and call order; the per-step work underneath
(sample + update) is the HIP path.  Logging reads GPU scalars only at the
logging cadence, so there is no per-step host sync."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable

import numpy as np
import torch as t

from oprl_amd.algos.protocols import AlgorithmProtocol
from oprl_amd.buffers.protocols import ReplayBufferProtocol
from oprl_amd.environment.protocols import EnvProtocol
from oprl_amd.logging import LoggerProtocol, create_stdout_logger
from oprl_amd.trainers.protocols import TrainerProtocol

logger = create_stdout_logger()


@dataclass
class BaseTrainer(TrainerProtocol):
    logger: LoggerProtocol
    env: EnvProtocol
    make_env_test: Callable[[int], EnvProtocol]
    replay_buffer: ReplayBufferProtocol
    algo: AlgorithmProtocol
    gamma: float = 0.99
    num_steps: int = int(1e6)
    start_steps: int = int(10e3)
    batch_size: int = 128          # the reference's effective batch (SURVEY.md §8d caveat)
    eval_interval: int = int(2e3)
    num_eval_episodes: int = 10
    save_buffer_every: int = 0
    save_policy_every: int = int(100_000)
    save_checkpoint_every: int = 0      # > 0: full learner + replay state every so many env steps (N4)
    estimate_q_every: int = 0
    stdout_log_every: int = int(1e5)
    fused_sample_update: bool = True    # sample + update as one C call (algo.update_from_buffer); False: the two calls
    device: str = "cuda"
    seed: int = 0

    def train(self) -> None:
        self.algo.check_created()
        self.replay_buffer.check_created()
        state, _ = self.env.reset()
        for env_step in range(self.num_steps + 1):
            if env_step <= self.start_steps:
                action = self.env.sample_action()
            else:
                action = self.algo.actor.explore(state)
            next_state, reward, terminated, truncated, _ = self.env.step(action)
            self.replay_buffer.add_transition(state, action, reward, terminated,
                                              episode_done=terminated or truncated)
            if terminated or truncated:
                next_state, _ = self.env.reset()
            state = next_state
            if len(self.replay_buffer) < self.batch_size:
                continue
            fused = self.fused_sample_update and hasattr(self.algo, "update_from_buffer")
            if fused:
                # reference: sample() then update(*batch) (base_trainer.py:63-70); here the update
                # gathers its own rows on the device.  The logging below wants a batch of rewards
                # only at its own cadence.
                self.algo.update_from_buffer(self.replay_buffer, self.batch_size)
                need_rewards = (env_step % self.eval_interval == 0) or (env_step % self.stdout_log_every == 0)
                rewards = self.replay_buffer.sample(self.batch_size)[2] if need_rewards else None
            else:
                batch = self.replay_buffer.sample(self.batch_size)
                self.algo.update(*batch)
                rewards = batch[2]
            self._log_evaluation(env_step, rewards)
            self._save_policy(env_step)
            if self.save_checkpoint_every > 0 and env_step % self.save_checkpoint_every == 0:
                self.save_checkpoint(self.logger.log_dir / "checkpoints" / f"{env_step}.ckpt", env_step)
            self._log_stdout(env_step, rewards)

    def _log_evaluation(self, env_step: int, rewards: t.Tensor) -> None:
        if env_step % self.eval_interval != 0:
            return
        metrics = self.evaluate()
        rb = self.replay_buffer
        self.logger.log_scalar("trainer/ep_reward", metrics["return"], env_step)
        self.logger.log_scalar("trainer/avg_reward", rewards.mean().item(), env_step)
        self.logger.log_scalar("trainer/buffer_transitions", len(rb), env_step)
        self.logger.log_scalar("trainer/buffer_episodes", rb.episodes_counter, env_step)
        self.logger.log_scalar("trainer/buffer_last_ep_len", rb.last_episode_length, env_step)

    def evaluate(self) -> dict[str, float]:
        returns = []
        for i_ep in range(self.num_eval_episodes):
            env_test = self.make_env_test(self.seed + i_ep)
            state, _ = env_test.reset()
            total, done = 0.0, False
            while not done:
                state, reward, terminated, truncated, _ = env_test.step(self.algo.actor.exploit(state))
                total += reward
                done = terminated or truncated
            returns.append(total)
        return {"return": float(np.mean(returns))}

    def _save_policy(self, env_step: int) -> None:
        if self.save_policy_every > 0 and env_step % self.save_policy_every == 0:
            path = self.logger.log_dir / "weights" / f"{env_step}.w"
            path.parent.mkdir(parents=True, exist_ok=True)
            t.save(self.algo.actor, path)

    # Full-state checkpoint (the reference only pickles the policy, base_trainer.py:113-120):
    # learner arenas + Adam moments + counters and the replay with its write / sample positions;
    # restoring both resumes the update stream bit for bit (tests/test_gpu_callers.py).
    def save_checkpoint(self, path, env_step: int = 0) -> None:
        from pathlib import Path
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        t.save({"env_step": int(env_step), "algo": self.algo.state_dict(),
                "replay": self.replay_buffer.state_dict()}, path)

    def load_checkpoint(self, path) -> int:
        ck = t.load(path, weights_only=False)
        self.algo.load_state_dict(ck["algo"])
        self.replay_buffer.load_state_dict(ck["replay"])
        return int(ck["env_step"])

    # Q-value sanity probe (reference base_trainer.py:122-174; defined there but never called from
    # train(), kept with the same names and semantics): discounted Monte-Carlo return of the greedy
    # policy against the critic's Q at the first state-action of the same episodes.
    def _estimate_q(self, env_step: int) -> None:
        if self.estimate_q_every > 0 and env_step % self.estimate_q_every == 0:
            q_true = self.estimate_true_q()
            q_critic = self.estimate_critic_q()
            self.logger.log_scalar("trainer/Q-estimate", q_true, env_step)
            self.logger.log_scalar("trainer/Q-critic", q_critic, env_step)
            self.logger.log_scalar("trainer/Q_asb_diff", q_critic - q_true, env_step)

    def estimate_true_q(self, eval_episodes: int = 10) -> float:
        qs = []
        for i_eval in range(eval_episodes):
            env = self.make_env_test(self.seed * 100 + i_eval)
            state, _ = env.reset()
            q, discount, done = 0.0, self.gamma, False
            while not done:
                state, r, terminated, truncated, _ = env.step(self.algo.actor.exploit(state))
                q += r * discount          # (the reference starts the discount at gamma^1)
                discount *= self.gamma
                done = terminated or truncated
            qs.append(q)
        return float(np.mean(qs))

    def estimate_critic_q(self, num_episodes: int = 10) -> float:
        qs = []
        for i_eval in range(num_episodes):
            env = self.make_env_test(self.seed * 100 + i_eval)
            state, _ = env.reset()
            action = self.algo.actor.exploit(state)
            s = t.as_tensor(np.asarray(state), dtype=t.float32, device=self.device).unsqueeze(0)
            a = t.as_tensor(np.asarray(action), dtype=t.float32, device=self.device).unsqueeze(0)
            q = self.algo.critic(s, a)
            if isinstance(q, tuple):       # twin critics: Q1
                q = q[0]
            qs.append(float(q.reshape(-1)[0].item()))   # (quantile critics: first entry, as the reference's .item() would need)
        return float(np.mean(qs))

    def _log_stdout(self, env_step: int, rewards: t.Tensor) -> None:
        if env_step % self.stdout_log_every == 0:
            perc = int(env_step / max(self.num_steps, 1) * 100)
            logger.info(f"Env step {env_step:8d} ({perc:2d}%) Avg Reward {rewards.mean().item():10.3f}")
