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

    # ---- the loop ----------------------------------------------------------------------------------
    # Call order per environment step, as in the reference (base_trainer.py:38-74): choose an action
    # (uniform during the first ``start_steps``, then ``actor.explore``), step, store the transition,
    # and — once the buffer holds a batch — one update, then the periodic work.
    def train(self) -> None:
        self.algo.check_created()
        self.replay_buffer.check_created()
        if self.fused_sample_update and hasattr(self.replay_buffer, "handle"):
            self.replay_buffer.eager_flush = True        # (buffers/episodic_buffer.py::add_transition)
        obs, _ = self.env.reset()
        for step in range(self.num_steps + 1):
            obs = self._collect(step, obs)
            if len(self.replay_buffer) < self.batch_size:
                continue
            rewards = self._learn(step, obs)
            self._periodic(step, rewards)

    def _collect(self, step: int, obs):
        """One environment step into the replay; returns the observation the next step starts from."""
        warm_up = step <= self.start_steps
        action = self.env.sample_action() if warm_up else self.algo.actor.explore(obs)
        nxt, reward, terminated, truncated, _ = self.env.step(action)
        over = bool(terminated or truncated)
        self.replay_buffer.add_transition(obs, action, reward, terminated, episode_done=over)
        if over:
            nxt, _ = self.env.reset()
        return nxt

    def _learn(self, step: int, next_obs=None):
        """One update.  Fused (default): ``algo.update_from_buffer`` — the kernels gather their own rows on
        the device, one C call; a batch of rewards is sampled only when the logging below will print it.
        Otherwise the reference's two calls, ``sample()`` then ``update(*batch)``.
        ``next_obs``: what the next step's ``actor.explore`` will be called with — its forward rides behind the update
        (when the next step explores at all: not during the uniform warm-up, not before an evaluation / checkpoint
        touches the actor in between)."""
        if self.fused_sample_update and hasattr(self.algo, "update_from_buffer"):
            ride = next_obs is not None and hasattr(self.algo, "_actor_mlp") and step + 1 > self.start_steps and step + 1 <= self.num_steps and self._quiet(step)
            if ride:
                self.algo.update_from_buffer(self.replay_buffer, self.batch_size, act_next=next_obs)
            else:
                self.algo.update_from_buffer(self.replay_buffer, self.batch_size)
            wanted = step % self.eval_interval == 0 or step % self.stdout_log_every == 0
            return self.replay_buffer.sample(self.batch_size)[2] if wanted else None
        batch = self.replay_buffer.sample(self.batch_size)
        self.algo.update(*batch)
        return batch[2]

    def _quiet(self, step: int) -> bool:
        """No periodic work of this step uses the actor between the update and the next ``explore``."""
        def due(every):
            return every > 0 and step % every == 0
        return not (due(self.eval_interval) or due(self.save_policy_every) or due(self.save_checkpoint_every)
                    or due(self.estimate_q_every))

    def _periodic(self, step: int, rewards) -> None:
        self._log_evaluation(step, rewards)
        self._save_policy(step)
        if self.save_checkpoint_every > 0 and step % self.save_checkpoint_every == 0:
            self.save_checkpoint(self.logger.log_dir / "checkpoints" / f"{step}.ckpt", step)
        self._log_stdout(step, rewards)

    def _log_evaluation(self, env_step: int, rewards: t.Tensor) -> None:
        if env_step % self.eval_interval != 0:
            return
        buf, log = self.replay_buffer, self.logger.log_scalar
        log("trainer/ep_reward", self.evaluate()["return"], env_step)
        log("trainer/avg_reward", rewards.mean().item(), env_step)
        log("trainer/buffer_transitions", len(buf), env_step)
        log("trainer/buffer_episodes", buf.episodes_counter, env_step)
        log("trainer/buffer_last_ep_len", buf.last_episode_length, env_step)

    def evaluate(self) -> dict[str, float]:
        """Mean undiscounted return of the greedy policy over ``num_eval_episodes`` fresh environments."""
        totals = []
        for k in range(self.num_eval_episodes):
            env = self.make_env_test(self.seed + k)
            obs, _ = env.reset()
            ret, over = 0.0, False
            while not over:
                obs, reward, terminated, truncated, _ = env.step(self.algo.actor.exploit(obs))
                ret += reward
                over = bool(terminated or truncated)
            totals.append(ret)
        return {"return": float(np.mean(totals))}

    def _save_policy(self, env_step: int) -> None:
        """Whole-module pickle of the actor, as the reference does (base_trainer.py:113-120)."""
        if self.save_policy_every <= 0 or env_step % self.save_policy_every != 0:
            return
        target = self.logger.log_dir / "weights" / f"{env_step}.w"
        target.parent.mkdir(parents=True, exist_ok=True)
        t.save(self.algo.actor, target)

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

    def _probe_envs(self, n: int):
        """The probe's environments: seeds ``seed * 100 + k``, each reset, as (env, first observation)."""
        for k in range(n):
            env = self.make_env_test(self.seed * 100 + k)
            first, _ = env.reset()
            yield env, first

    def estimate_true_q(self, eval_episodes: int = 10) -> float:
        """Discounted Monte-Carlo return of the greedy policy (the first reward already carries one factor
        of gamma, as in the reference)."""
        returns = []
        for env, obs in self._probe_envs(eval_episodes):
            total, weight, over = 0.0, self.gamma, False
            while not over:
                obs, reward, terminated, truncated, _ = env.step(self.algo.actor.exploit(obs))
                total += weight * reward
                weight *= self.gamma
                over = bool(terminated or truncated)
            returns.append(total)
        return float(np.mean(returns))

    def estimate_critic_q(self, num_episodes: int = 10) -> float:
        """The critic's value of the greedy action at the first state of the same episodes (twin critics:
        Q1; quantile critics: the first entry)."""
        values = []
        for _env, obs in self._probe_envs(num_episodes):
            act = self.algo.actor.exploit(obs)
            s = t.as_tensor(np.asarray(obs), dtype=t.float32, device=self.device).unsqueeze(0)
            a = t.as_tensor(np.asarray(act), dtype=t.float32, device=self.device).unsqueeze(0)
            q = self.algo.critic(s, a)
            q = q[0] if isinstance(q, tuple) else q
            values.append(float(q.reshape(-1)[0].item()))
        return float(np.mean(values))

    def _log_stdout(self, env_step: int, rewards: t.Tensor) -> None:
        if env_step % self.stdout_log_every == 0:
            perc = int(env_step / max(self.num_steps, 1) * 100)
            logger.info(f"Env step {env_step:8d} ({perc:2d}%) Avg Reward {rewards.mean().item():10.3f}")
