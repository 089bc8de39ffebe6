"""CPU tests of the host side: C-ABI surface (every declared symbol exported and
bound, struct layouts agree with the C compiler), replay bookkeeping against
the golden trace recorded from the reference buffer, and the call sequences of
the trainer / distributed-learner loops (fixture G7 of SURVEY.md §8c)."""
import ctypes as C
import pickle
import re
import subprocess
import tempfile
from pathlib import Path

import numpy as np
import pytest
import torch as t

from tests import scenarios as sc

ROOT = Path(__file__).resolve().parents[1]
HEADER = ROOT / "include" / "oprl_amd.h"


def _declared_functions():
    src = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(oprl_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    from oprl_amd import _capi
    lib = _capi.load()              # does not touch the GPU
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in oprl_amd.h but not exported"
        assert n in _capi.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_capi.SIGNATURES) == set(names)
    assert lib.oprl_abi_version() == _capi.OPRL_ABI_VERSION
    assert isinstance(lib.oprl_last_error(), bytes)


def test_struct_layouts_match_the_c_compiler():
    from oprl_amd import _capi
    prog = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "oprl_amd.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(oprl_net), sizeof(oprl_hparams),
             sizeof(oprl_learner_config), offsetof(oprl_learner_config, actor),
             offsetof(oprl_learner_config, critics), offsetof(oprl_learner_config, log_alpha),
             offsetof(oprl_learner_config, hp));
      return 0; }'''
    with tempfile.TemporaryDirectory() as tmp:
        (Path(tmp) / "p.c").write_text(prog)
        subprocess.run(["gcc", "-I", str(HEADER.parent), str(Path(tmp) / "p.c"), "-o", str(Path(tmp) / "p")], check=True)
        out = subprocess.run([str(Path(tmp) / "p")], check=True, capture_output=True, text=True).stdout.split()
    got = [int(x) for x in out]
    cfg = _capi.OprlLearnerConfig
    want = [C.sizeof(_capi.OprlNet), C.sizeof(_capi.OprlHparams), C.sizeof(cfg), cfg.actor.offset,
            cfg.critics.offset, cfg.log_alpha.offset, cfg.hp.offset]
    assert got == want


def test_null_and_invalid_arguments_fail_loudly_without_a_gpu():
    from oprl_amd import _capi
    lib = _capi.load()
    assert lib.oprl_learner_create(None, None) != 0
    assert b"null" in lib.oprl_last_error()
    with pytest.raises(RuntimeError, match="status"):
        _capi.check(lib.oprl_replay_create(0, 0, 0, 0, None, None, None, None, None), "oprl_replay_create")
    assert lib.oprl_learner_update(None, None, None, None, None, None, 8, None, None, None) != 0


def test_learner_refuses_cpu_device():
    from oprl_amd.algos.ddpg import DDPG
    from oprl_amd.logging import NullLogger
    algo = DDPG(logger=NullLogger(), state_dim=4, action_dim=2, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU path"):
        algo.create()
    with pytest.raises(RuntimeError, match="create"):
        algo.check_created()


def test_replay_bookkeeping_matches_reference_trace_on_cpu():
    """Ring pointer / eviction / counters are host logic: identical to the trace
    recorded from the reference buffer (data movement is checked on the GPU)."""
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    from oprl_amd.buffers.protocols import ReplayBufferProtocol
    gold = sc.load_golden("replay_script")
    cap, S, A, L = (int(x) for x in gold["meta"])
    buf = EpisodicReplayBuffer(buffer_size_transitions=cap, state_dim=S, action_dim=A,
                               max_episode_lenth=L, device="cpu").create()
    assert isinstance(buf, ReplayBufferProtocol)
    assert buf.states.shape == (cap // L, L + 1, S)

    def snap(b):
        return [len(b), b.episodes_counter, b._ep_pointer, b.last_episode_length, *b.ep_lens], {}

    got = sc.replay_scenario(buf, S, A, snap)
    assert np.array_equal(got["trace"], gold["trace"])
    with pytest.raises(RuntimeError, match="no CPU sampler"):
        buf.sample(4)


def test_pack_index_is_a_bijection_and_matches_the_layout_formula():
    from oprl_amd.algos.nn_models import pack_floats
    # python restatement of csrc/engine.h pack_index
    def pack_index(r, c, NS):
        tile, i, s, kk, tt = r >> 4, r & 15, c >> 4, (c & 15) >> 2, c & 3
        return ((tile * NS + s) * 64 + (kk * 16 + i)) * 4 + tt
    for rows, cols in ((256, 30), (1, 256), (25, 512), (42, 256)):
        NS = -(-cols // 16)
        idx = {pack_index(r, c, NS) for r in range(rows) for c in range(cols)}
        assert len(idx) == rows * cols
        assert max(idx) < (-(-rows // 16)) * NS * 256
    assert pack_floats([30, 256, 256, 1]) == 2 * (16 * 2 + 16 * 16 + 1 * 16) * 256


# ------------------------------------------------------------------ callers (G7)
class _FakeAlgo:
    def __init__(self):
        self.calls = []
        self.logger = None
        self._created = True
        self.actor = self

    def check_created(self): pass
    def explore(self, s): self.calls.append("explore"); return np.zeros(6, np.float32)
    def exploit(self, s): return np.zeros(6, np.float32)
    def update(self, *batch): self.calls.append("update")
    def get_policy_state_dict(self): return {"w": t.zeros(1)}
    def state_dict(self): return {"w": t.zeros(1)}


class _FakeBuffer:
    def __init__(self): self.n, self.calls, self.episodes_counter, self.last_episode_length = 0, [], 1, 0
    def check_created(self): pass
    def add_transition(self, *a, **k): self.n += 1; self.calls.append("add")
    def add_episode(self, ep): self.n += len(ep); self.calls.append("add_episode")
    def sample(self, B): self.calls.append("sample"); return tuple(t.zeros(B, 1) for _ in range(5))
    def __len__(self): return self.n


def test_trainer_call_sequence_and_warmup_gating():
    from oprl_amd.environment import make_env
    from oprl_amd.logging import NullLogger
    from oprl_amd.trainers.base_trainer import BaseTrainer
    algo, buf = _FakeAlgo(), _FakeBuffer()
    tr = BaseTrainer(logger=NullLogger("/tmp/oprl_amd_test"), env=make_env("walker-walk", 0),
                     make_env_test=lambda s: make_env("walker-walk", s), replay_buffer=buf, algo=algo,
                     num_steps=40, start_steps=20, batch_size=8, eval_interval=10 ** 9,
                     save_policy_every=0, stdout_log_every=10 ** 9)
    tr.train()
    # num_steps + 1 iterations; sample/update only once len(buffer) >= batch_size (step 8 onwards)
    assert buf.calls.count("add") == 41
    assert buf.calls.count("sample") == algo.calls.count("update") == 41 - 7
    # random actions for env_step <= start_steps, policy afterwards
    assert algo.calls.count("explore") == 40 - 20
    first_sample = buf.calls.index("sample")
    assert buf.calls[:first_sample] == ["add"] * 8


def test_distributed_learner_loop_counts():
    from oprl_amd.distrib.policy_update_worker import run_policy_update_worker
    from oprl_amd.distrib.queue import Queue, QueueHub
    from oprl_amd.logging import NullLogger
    from oprl_amd.runners.config import DistribConfig
    import multiprocessing as mp
    cfg = DistribConfig(batch_size=4, num_env_workers=2, warmup_epochs=1, episode_length=5, learner_num_waits=1)
    hub = QueueHub([f"{k}_{i}" for i in range(2) for k in ("env", "policy")], mp.get_context("spawn"))
    ep = [[np.zeros(3, np.float32), np.zeros(2, np.float32), 0.0, False, np.zeros(3, np.float32)]] * 5
    n_epochs = 4
    for i in range(2):
        q = Queue(f"env_{i}", hub)
        for _ in range(n_epochs):
            q.push(pickle.dumps(ep))
    import time; time.sleep(0.3)          # let the feeder threads flush
    algo, buf = _FakeAlgo(), _FakeBuffer()
    algo.logger = NullLogger("/tmp/oprl_amd_test")
    run_policy_update_worker(lambda lg: algo, None, lambda: buf, lambda: algo.logger, cfg, hub,
                             max_epochs=n_epochs, wait_s=0.01)
    assert buf.calls.count("add_episode") == n_epochs * 2
    # updates only for epochs > warmup_epochs (2 and 3): episode_length * num_env_workers each
    assert algo.calls.count("update") == 2 * 5 * 2
    time.sleep(0.3)
    pushed = 0
    qp = Queue("policy_0", hub)
    while True:
        d = qp.pop()
        if d is None:
            break
        pushed += 1
    assert pushed == n_epochs + 1          # one state_dict per epoch + the final STOP


def test_alias_package_serves_the_reference_config_scripts():
    """Every name the reference's config scripts import (configs/{ddpg,td3,sac,tqc}.py:1-13,
    configs/distrib_ddpg.py:1-22) resolves through the ``oprl`` alias package, and
    ``run_distrib_training`` takes the keyword arguments distrib_ddpg.py:80-90 passes."""
    import importlib
    import inspect
    names = {
        "oprl.algos.ddpg": ["DDPG"], "oprl.algos.td3": ["TD3"], "oprl.algos.sac": ["SAC"], "oprl.algos.tqc": ["TQC"],
        "oprl.algos.nn_models": ["DeterministicPolicy", "GaussianActor"],
        "oprl.algos.protocols": ["AlgorithmProtocol", "PolicyProtocol"],
        "oprl.buffers.protocols": ["ReplayBufferProtocol"],
        "oprl.buffers.episodic_buffer": ["EpisodicReplayBuffer"],
        "oprl.environment": ["make_env"], "oprl.environment.protocols": ["EnvProtocol"],
        "oprl.logging": ["LoggerProtocol", "FileTxtLogger", "get_logs_path", "make_text_logger_func"],
        "oprl.parse_args": ["parse_args", "parse_args_distrib"],
        "oprl.runners.config": ["CommonParameters", "DistribConfig"],
        "oprl.runners.train": ["run_training"], "oprl.runners.train_distrib": ["run_distrib_training"],
        "oprl.distrib.env_worker": ["run_env_worker"],
        "oprl.distrib.policy_update_worker": ["run_policy_update_worker"],
    }
    for mod, ns in names.items():
        m = importlib.import_module(mod)
        for n in ns:
            assert hasattr(m, n), f"{mod}.{n}"
    import oprl.environment
    from oprl.environment.make_env import make_env as by_module
    assert callable(oprl.environment.make_env) and oprl.environment.make_env is by_module   # not shadowed by the submodule
    from oprl.runners.train_distrib import run_distrib_training
    params = inspect.signature(run_distrib_training).parameters
    for k in ("run_env_worker", "run_policy_update_worker", "make_env", "make_algo", "make_policy",
              "make_replay_buffer", "make_logger", "config"):
        assert k in params, k
    from oprl.runners.train import run_training
    params = inspect.signature(run_training).parameters
    for k in ("make_algo", "make_env", "make_replay_buffer", "make_logger", "config", "seeds", "start_seed"):
        assert k in params, k


def test_config_script_factories_pickle_for_the_multi_seed_runner(monkeypatch):
    """``--seeds N`` spawns one process per seed and passes the script's factories: all of them must
    pickle (a closure-returning logger factory did not)."""
    import importlib
    import pickle
    import sys
    from pathlib import Path
    monkeypatch.setattr(sys, "argv", ["sac.py", "--env", "walker-walk", "--device", "cuda"])
    monkeypatch.syspath_prepend(str(Path(__file__).resolve().parents[1] / "configs"))
    sys.modules.pop("_common", None)
    mod = importlib.import_module("sac")
    try:
        for name in ("make_env", "make_algo", "make_replay_buffer", "make_logger", "config"):
            pickle.loads(pickle.dumps(getattr(mod, name)))
        assert mod.config.state_dim == 24 and mod.config.action_dim == 6
    finally:
        sys.modules.pop("sac", None)
        sys.modules.pop("_common", None)


def test_synthetic_env_dynamics_do_not_depend_on_the_interpreter_hash_seed():
    """Actors, learner and --seeds children are separate processes: the stand-in's dynamics must be the
    same in all of them (str hashes are salted per interpreter; the task name is hashed with crc32)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("from oprl_amd.environment import make_env; import zlib; "
            "e = make_env('synthetic:walker-walk', 3); print(zlib.crc32(e._F.tobytes() + e._G.tobytes() + e._goal.tobytes()))")
    outs = []
    for hs in ("1", "2"):
        env = dict(os.environ, PYTHONHASHSEED=hs)
        outs.append(subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True,
                                   text=True, check=True).stdout.strip())
    assert outs[0] == outs[1]
    from oprl_amd.environment import make_env
    e = make_env("walker-walk", 0)          # bare name: accepted (reference command lines), but labelled
    assert e.env_family == "synthetic"
    with pytest.raises(ValueError):
        make_env("Ant-v4", 0)


def test_add_transitions_block_equals_transition_by_transition():
    """The block write the distributed learner ranks use (one library call per actor episode) leaves the
    replay exactly as the same transitions added one by one: storage, lengths, pointer, eviction."""
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer

    def mk():
        return EpisodicReplayBuffer(buffer_size_transitions=40, state_dim=3, action_dim=2, max_episode_lenth=10,
                                    device="cpu").create()
    a, b = mk(), mk()
    rs = np.random.RandomState(0)
    for _ep in range(9):                       # 4 slots: wraps and evicts
        n = int(rs.randint(1, 11))
        rows = rs.standard_normal((n, 3 + 2 + 3)).astype(np.float32)
        rows[:, 6] = rs.rand(n) < 0.1
        for k, r in enumerate(rows):
            a.add_transition(r[:3], r[3:5], float(r[5]), bool(r[6]), episode_done=(k == n - 1))
        b.add_transitions(rows, episode_done=True)
        assert (a.ep_lens, len(a), a.episodes_counter, a._ep_pointer) == (b.ep_lens, len(b), b.episodes_counter, b._ep_pointer)
    for k in ("states", "actions", "rewards", "dones"):
        assert t.equal(a._tensors[k], b._tensors[k]), k
    with pytest.raises(IndexError):
        b.add_transitions(np.zeros((11, 8), np.float32))


def test_bench_failure_line_keeps_the_rccl_probe():
    """bench.py at N > 1: once the RCCL probe has produced its number, a failure of anything later (a peer-window probe
    that hangs into the watchdog, the rebuild) still reports that number — with its own step count and a note — not
    value null (the first multi-GPU run must produce the whole answer)."""
    import argparse
    import json
    import bench
    args = argparse.Namespace(steps=20, warmup=5, precision="f32")
    try:
        d = json.loads(bench.failure_line(args, 8, "watchdog: phase 'x' exceeded 900 s on rank 0"))
        assert d["value"] is None and d["steps"] == 20 and d["n_gpus"] == 8 and d["fallback"] is None
        bench.FALLBACK.update(value=123456.7, steps=1000, warmup=300, ms_per_step=0.0648, note="the RCCL probe's rate",
                              data_parallel_check={"exchange": "rccl", "probe_us_per_step": {"rccl": 64.8}})
        d = json.loads(bench.failure_line(args, 8, "watchdog: phase 'data-parallel probe, exchange level 1' exceeded 900 s"))
        assert d["value"] == 123456.7 and d["steps"] == 1000 and d["data_parallel_check"]["exchange"] == "rccl"
        assert "watchdog" in d["error"] and d["fallback"]
    finally:
        bench.FALLBACK.clear()
