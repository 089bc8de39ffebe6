"""GPU parity of the HBM replay buffer: ring/eviction bookkeeping + gather
kernel, bit-exact against the golden trace recorded from the reference buffer."""
import numpy as np
import pytest
import torch as t

from tests import scenarios as sc

pytestmark = pytest.mark.gpu


def _snapshot(buf):
    row = [len(buf), buf.episodes_counter, buf._ep_pointer, buf.last_episode_length, *buf.ep_lens]
    g = {}
    n = len(buf)
    if n > 0:
        (s, a, r, d, s2), (ep, st) = buf.sample(n, inds=np.arange(n), return_indices=True)
        g = dict(ep=ep.cpu().numpy(), step=st.cpu().numpy(), s=s.cpu().numpy(), a=a.cpu().numpy(),
                 r=r.cpu().numpy(), d=d.cpu().numpy(), s2=s2.cpu().numpy())
    return row, g


def test_replay_script_bitexact_vs_reference():
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    gold = sc.load_golden("replay_script")
    cap, S, A, L = (int(x) for x in gold["meta"])
    buf = EpisodicReplayBuffer(buffer_size_transitions=cap, state_dim=S, action_dim=A,
                               max_episode_lenth=L, device="cuda").create()
    for k in ("states", "actions", "rewards", "dones"):
        buf._tensors[k].fill_(-99.0)
    got = sc.replay_scenario(buf, S, A, _snapshot)
    for k, w in gold.items():
        if k == "meta":
            continue
        assert np.array_equal(np.asarray(got[k]), w), k


def test_full_size_gather_properties():
    """1e6-transition buffer (BASELINE sizes): every sampled row must equal the
    storage row its (episode, step) names; device-drawn indices are in range and
    roughly uniform."""
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    S, A, E, L = 24, 6, 1000, 1000
    buf = EpisodicReplayBuffer(buffer_size_transitions=E * L, state_dim=S, action_dim=A,
                               device="cuda", seed=123).create()
    gen = t.Generator(device="cuda").manual_seed(0)
    buf._tensors["states"].copy_(t.randn((E, L + 1, S), device="cuda", generator=gen))
    buf._tensors["actions"].copy_(t.rand((E, L, A), device="cuda", generator=gen) * 2 - 1)
    buf._tensors["rewards"].copy_(t.rand((E, L, 1), device="cuda", generator=gen))
    lens = np.random.RandomState(0).randint(1, L + 1, size=E)
    buf.ep_lens = [int(x) for x in lens]
    buf.episodes_counter = E
    buf._number_transitions = int(lens.sum())
    buf._lens_dirty = True
    B = 4096
    (s, a, r, d, s2), (ep, st) = buf.sample(B, return_indices=True)
    ep_l, st_l = ep.long(), st.long()
    assert int(st_l.min()) >= 0 and bool((st_l < t.as_tensor(lens, device="cuda")[ep_l]).all())
    assert t.equal(s, buf.states[ep_l, st_l]) and t.equal(s2, buf.states[ep_l, st_l + 1])
    assert t.equal(a, buf.actions[ep_l, st_l]) and t.equal(r, buf.rewards[ep_l, st_l])
    assert t.equal(d, buf.dones[ep_l, st_l])
    # flat index reconstructed from (ep, step) must be uniform over [0, N)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    flat = starts[ep.cpu().numpy()] + st.cpu().numpy()
    N = int(lens.sum())
    assert flat.min() >= 0 and flat.max() < N
    hist, _ = np.histogram(flat, bins=16, range=(0, N))
    assert hist.min() > B / 16 * 0.6 and hist.max() < B / 16 * 1.4
    # two draws differ, same (seed, counter) would repeat
    (s_b, *_), _ = buf.sample(B, return_indices=True)
    assert not t.equal(s, s_b)


def test_empty_and_ragged():
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    buf = EpisodicReplayBuffer(buffer_size_transitions=40, state_dim=3, action_dim=2,
                               max_episode_lenth=10, device="cuda").create()
    with pytest.raises(ValueError):
        buf.sample(4)
    buf.add_transition(np.ones(3, np.float32), np.ones(2), 1.0, False)
    out = buf.sample(5)        # batch larger than the buffer: with replacement
    assert out[0].shape == (5, 3) and bool((out[0] == 1).all())
    with pytest.raises(IndexError):
        for _ in range(11):
            buf.add_transition(np.ones(3, np.float32), np.ones(2), 1.0, False)


def test_block_write_equals_transition_writes_on_device():
    """oprl_replay_write_block (whole actor episodes, staging overflow included) against oprl_replay_write:
    identical HBM storage and identical gathers, across ring wrap-around."""
    import numpy as np
    import torch as t
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer

    def mk():
        return EpisodicReplayBuffer(buffer_size_transitions=6000, state_dim=7, action_dim=3, max_episode_lenth=500,
                                    device="cuda", seed=5).create()
    a, b = mk(), mk()
    rs = np.random.RandomState(1)
    for _ep in range(30):                      # 12 slots of 500: wraps; 30 x ~400 rows overflow the 4096-row staging
        n = int(rs.randint(300, 501))
        rows = rs.standard_normal((n, 7 + 3 + 3)).astype(np.float32)
        rows[:, 11] = rs.rand(n) < 0.01
        for k, r in enumerate(rows):
            a.add_transition(r[:7], r[7:10], float(r[10]), bool(r[11]), episode_done=(k == n - 1))
        b.add_transitions(rows, episode_done=True)
    assert a.ep_lens == b.ep_lens and len(a) == len(b)
    for k in ("states", "actions", "rewards", "dones"):
        assert t.equal(getattr(a, k), getattr(b, k)), k
    inds = np.random.RandomState(2).randint(0, len(a), 512)
    for x, y in zip(a.sample(512, inds=inds), b.sample(512, inds=inds)):
        assert t.equal(x, y)


def test_eager_flush_equals_batched_writes_on_device():
    """The trainer loop's add_transition (EpisodicReplayBuffer.eager_flush: one library call per transition —
    oprl_replay_write_flush — whose ingest kernel reads the pinned staging row and the changed tail of the episode table)
    against the batched path (rows staged, table uploaded, one scatter at the next sample): identical storage, episode
    table and draws, across ring wrap-around, with samples taken in between."""
    import numpy as np
    import torch as t
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer

    def mk(eager):
        b = EpisodicReplayBuffer(buffer_size_transitions=600, state_dim=5, action_dim=2, max_episode_lenth=50,
                                 device="cuda", seed=9).create()
        b.eager_flush = eager
        return b
    a, b = mk(True), mk(False)
    rs = np.random.RandomState(3)
    for ep in range(30):                       # 12 slots of 50: wraps twice
        n = int(rs.randint(20, 51))
        for k in range(n):
            s, ac, r = rs.standard_normal(5).astype(np.float32), rs.uniform(-1, 1, 2), float(rs.uniform())
            for buf in (a, b):
                buf.add_transition(s, ac, r, False, episode_done=(k == n - 1))
            if (ep * 50 + k) % 37 == 0:        # a draw in the middle of an episode: the table's tail just changed
                for x, y in zip(a.sample(16), b.sample(16)):
                    assert t.equal(x, y)
    assert a.ep_lens == b.ep_lens and len(a) == len(b)
    inds = np.random.RandomState(2).randint(0, len(a), 256)
    for x, y in zip(a.sample(256, inds=inds), b.sample(256, inds=inds)):
        assert t.equal(x, y)
    for k in ("states", "actions", "rewards", "dones"):
        assert t.equal(getattr(a, k), getattr(b, k)), k


def test_index_map_with_more_episodes_than_the_lds_table_holds():
    """5000 short episodes (> the 2048-entry LDS table: coarse table + global finish, csrc/replay_index.h), ragged
    lengths: injected flat indices must map to the (episode, step) numpy's cumulative-ends rule gives — the gather
    kernel and the fused update kernels' own sampler (step_n) share the routine."""
    import numpy as np
    import torch as t
    from oprl_amd.buffers.episodic_buffer import EpisodicReplayBuffer
    E, L, S, A = 5000, 40, 4, 2
    buf = EpisodicReplayBuffer(buffer_size_transitions=E * L, state_dim=S, action_dim=A, max_episode_lenth=L,
                               device="cuda").create()
    rs = np.random.RandomState(0)
    lens = rs.randint(1, L + 1, size=E)
    buf._tensors["states"].copy_(t.arange(E * (L + 1) * S, dtype=t.float32, device="cuda").view(E, L + 1, S))
    buf.ep_lens = [int(x) for x in lens]
    buf.episodes_counter = E
    buf._number_transitions = int(lens.sum())
    buf._lens_dirty = True
    ends = np.cumsum(lens)
    inds = np.concatenate([rs.randint(0, ends[-1], 2000), [0, ends[-1] - 1], ends[:50] - 1, ends[:50]])
    inds = inds[inds < ends[-1]]
    (s, _a, _r, _d, s2), (ep, st) = buf.sample(len(inds), inds=inds, return_indices=True)
    want_ep = np.searchsorted(ends, inds, side="right")
    want_st = inds - np.where(want_ep > 0, ends[np.maximum(want_ep - 1, 0)], 0)
    assert np.array_equal(ep.cpu().numpy(), want_ep) and np.array_equal(st.cpu().numpy(), want_st)
    base = (want_ep * (L + 1) + want_st) * S
    assert np.array_equal(s[:, 0].cpu().numpy(), base.astype(np.float32))
    assert np.array_equal(s2[:, 0].cpu().numpy(), (base + S).astype(np.float32))
