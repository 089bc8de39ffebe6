"""Pins the CPU oracle (oracle/oprl_oracle.py) against golden vectors produced
by running the reference itself (oracle/gen_golden.py).  CPU only."""
import numpy as np
import torch as t

from oracle import fixtures as fx
from oracle import oprl_oracle as orc
from tests import scenarios as sc

# fp32 summation-order noise between the reference's autograd graph and the
# explicit restatement; measured <= 1.9e-6 (worst: SAC actor weights after 2 updates); gate at 1e-5.
TOL = 1e-5


def test_ddpg_matches_reference():
    t.set_num_threads(1)
    got = sc.ddpg_scenario(sc.OracleDDPG)
    worst = sc.compare(got, sc.load_golden("ddpg_walker_b256"), TOL)
    print("ddpg worst", worst)


def test_td3_matches_reference():
    t.set_num_threads(1)
    got = sc.td3_scenario(sc.OracleTD3)
    print("td3 worst", sc.compare(got, sc.load_golden("td3_cheetah_b256"), TOL))


def test_sac_fixed_alpha_matches_reference():
    t.set_num_threads(1)
    got = sc.sac_scenario(sc.OracleSAC, "humanoid", 1024, 300, False, 2)
    print("sac worst", sc.compare(got, sc.load_golden("sac_humanoid_b1024"), TOL))


def test_sac_tuned_alpha_matches_reference():
    t.set_num_threads(1)
    got = sc.sac_scenario(sc.OracleSAC, "walker", 256, 350, True, 3)
    print("sac-tune worst", sc.compare(got, sc.load_golden("sac_walker_tune_b256"), TOL))


def test_tqc_matches_reference():
    t.set_num_threads(1)
    got = sc.tqc_scenario(sc.OracleTQC)
    gold = sc.load_golden("tqc_walker_b256")
    print("tqc worst", sc.compare(got, gold, TOL, skip=("qh.",)))


def test_quantile_huber_known_answer():
    gold = sc.load_golden("tqc_walker_b256")
    rs = np.random.RandomState(400 + 500)
    z = t.from_numpy((rs.standard_normal((8, 5, 25)) * 1.5).astype(np.float32))
    y = t.from_numpy((rs.standard_normal((8, 123)) * 1.5).astype(np.float32))
    loss, dz = orc.quantile_huber_loss(z, y)
    assert sc.rel_dev(loss.numpy(), gold["qh.loss"]) < 1e-6
    assert sc.rel_dev(dz.numpy(), gold["qh.dz"]) < 1e-6


def _snapshot_oracle(buf):
    row = [len(buf), buf.episodes_counter, buf.ep_pointer, buf.last_episode_length, *buf.ep_lens]
    g = {}
    n = len(buf)
    if n > 0:
        inds = np.arange(n)
        e, st = buf.inds_to_episodic(inds)
        s, a, r, d, s2 = buf.gather(inds)
        g = dict(ep=e, step=st, s=s, a=a, r=r, d=d, s2=s2)
    return row, g


def test_replay_matches_reference_bitexact():
    gold = sc.load_golden("replay_script")
    cap, S, A, L = (int(x) for x in gold["meta"])
    buf = orc.ReplayOracle(cap, S, A, max_episode_lenth=L, fill=-99.0)
    got = sc.replay_scenario(buf, S, A, _snapshot_oracle)
    for k, w in gold.items():
        if k == "meta":
            continue
        assert np.array_equal(np.asarray(got[k]), w), k


def test_replay_known_answer_from_survey():
    """SURVEY.md §8(c) G5 table (probe of the reference buffer)."""
    buf = orc.ReplayOracle(12, 2, 1, max_episode_lenth=4, fill=-99.0)
    k = -1
    for n in (4, 3, 4, 2):
        for i in range(n):
            k += 1
            buf.add_transition(np.full(2, k, np.float32), np.zeros(1), float(k), False,
                               episode_done=(i == n - 1))
    k += 1
    buf.add_transition(np.full(2, k, np.float32), np.zeros(1), float(k), False)
    assert len(buf) == 7 and buf.episodes_counter == 3 and buf.ep_pointer == 1
    assert buf.ep_lens == [2, 1, 4] and buf.last_episode_length == 1
    s, a, r, d, s2 = buf.gather(np.arange(7))
    assert s[:, 0].tolist() == [11, 12, 13, 7, 8, 9, 10]
    assert s2[:, 0].tolist() == [12, 2, 5, 8, 9, 10, -99]


def test_policy_io_matches_reference():
    gold = sc.load_golden("policy_io")
    S, A, seed = (int(x) for x in gold["meta"])
    det = fx.make_net(seed + 1, fx.actor_dims(S, A))
    ga = fx.make_net(seed + 2, fx.actor_dims(S, A, gaussian=True))
    obs = t.from_numpy(np.random.RandomState(seed + 3).standard_normal(S).astype(np.float32))[None]
    assert sc.rel_dev(orc.det_policy_forward(det, obs)[0][0].numpy(), gold["det.exploit"]) < 1e-6
    # explore = clip(mlp(s) + 0.1*noise): NO tanh (nn_models.py:144-150)
    raw = orc.mlp_forward(det, obs)[-1][0] + 0.1 * fx.make_noise(seed + 4, (A,))
    assert sc.rel_dev(raw.clamp(-1, 1).numpy(), gold["det.explore"]) < 1e-6
    mu = orc.mlp_forward(ga, obs)[-1][:, :A]
    assert sc.rel_dev(t.tanh(mu)[0].numpy(), gold["ga.exploit"]) < 1e-6
    a, _, _ = orc.gaussian_forward(ga, obs, fx.make_noise(seed + 5, (1, A)), A)
    assert sc.rel_dev(a[0].numpy(), gold["ga.explore"]) < 1e-6
    assert list(gold["det.keys"]) == [f"mlp.nn.{i}.{w}" for i in (0, 2, 4) for w in ("weight", "bias")]
    assert list(gold["ga.keys"]) == [f"net.nn.{i}.{w}" for i in (0, 2, 4) for w in ("weight", "bias")]
