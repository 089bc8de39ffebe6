"""Debug tool: a chain learner against a one-update-per-launch learner of the same stream, call by call; at every mismatch
say WHAT differs (which net, which 16 x 64 tiles), then resynchronise and go on."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
from oprl_amd.algos.ddpg import DDPG
from oprl_amd.logging import NullLogger
from tests.test_gpu_callers import _filled_buffer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
prec = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else "x2"
buf = _filled_buffer()


def make(env):
    for k in ("OPRL_AMD_CHAIN", "OPRL_AMD_FORM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t.manual_seed(0)
    return DDPG(logger=NullLogger(), state_dim=24, action_dim=6, device="cuda", max_batch=256, precision=prec).create()


EPIS = max(1, 4000 // K)          # calls per pair of learners (random data: the critic diverges after ~10k updates)
R = T = None
names = ("actor", "critic", "actor_target", "critic_target")
dims = {"actor": [(256, 24), (256,), (256, 256), (256,), (6, 256), (6,)], "critic": [(256, 30), (256,), (256, 256), (256,), (1, 256), (1,)]}
events = 0
for c in range(N):
    if c % EPIS == 0:
        R = T = None
        R = make({"OPRL_AMD_CHAIN": "1"})
        T = make({})
    T.learner.step_n(buf.handle, K, 256, seed=21 + c)      # (first: on an idle GPU)
    t.cuda.synchronize()
    R.learner.step_n(buf.handle, K, 256, seed=21 + c)
    t.cuda.synchronize()
    bad = [m for m in names if not t.equal(getattr(R, m)._oprl_arena, getattr(T, m)._oprl_arena)]
    if bad:
        events += 1
        if "--quiet" not in sys.argv:
            print(f"call {c}: differ: {bad}", flush=True)
        if "--quiet" not in sys.argv:
            qr, yr = R.learner.debug_q_y(256)
            qt, yt = T.learner.debug_q_y(256)
            print(f"   last update's in-kernel rows: q differs in {int((qr != qt).sum())} rows (max {float((qr - qt).abs().max()):.2e}), y in {int((yr != yt).sum())} rows (max {float((yr - yt).abs().max()):.2e})", flush=True)
        if "--views" in sys.argv:
            import ctypes as C
            hip = C.cdll.LoadLibrary("libamdhip64.so")
            vn = ["aX0", "aX1", "aX2", "pi", "gu", "du", "cX0", "cX1", "cX2", "cdY0", "cdY1", "seeds", "set0.s", "set1.s"]
            for w, nm in enumerate(vn):
                got = []
                for L in (R.learner, T.learner):
                    p_, n_ = C.c_void_p(), C.c_int64()
                    L.lib.oprl_learner_debug_view(L.handle, w, C.byref(p_), C.byref(n_))
                    b_ = t.empty(max(n_.value // 4, 1), dtype=t.float32, device="cuda")
                    if n_.value:
                        hip.hipMemcpy(C.c_void_p(b_.data_ptr()), p_, C.c_size_t(n_.value), C.c_int(3))
                    got.append(b_)
                d_ = (got[0].view(t.int32) != got[1].view(t.int32))
                print(f"   view {nm}: {int(d_.sum())} of {d_.numel()} words differ", flush=True)
                if nm in ("aX1", "aX2") and int(d_.sum()):
                    m2 = d_.view(256, 256)
                    cols = m2.any(0).nonzero().flatten().tolist()
                    rows = m2.any(1).nonzero().flatten().tolist()
                    dv = (got[0] - got[1]).abs().view(256, 256)
                    print(f"      rows with a difference: {len(rows)}; columns: {len(cols)} {cols[:40]}; max |diff| {float(dv.max()):.2e}; per-column counts of the first columns: {[int(m2[:, c].sum()) for c in cols[:12]]}", flush=True)
        for m in (bad if "--quiet" not in sys.argv else []):
            d = (getattr(R, m)._oprl_arena - getattr(T, m)._oprl_arena).abs().cpu()
            print(f"   {m}: non-finite in chain learner {int((~t.isfinite(getattr(T, m)._oprl_arena)).sum())}, in reference {int((~t.isfinite(getattr(R, m)._oprl_arena)).sum())}", flush=True)
            off = 0
            for shp in dims[m.split("_")[0]]:
                n = 1
                for s in shp:
                    n *= s
                blk = d[off:off + n].reshape(shp)
                off += n
                nz = int((blk > 0).sum())
                if nz:
                    if len(shp) == 2:
                        rows = sorted(set((blk > 0).nonzero()[:, 0].tolist()))
                        cols = sorted(set((blk > 0).nonzero()[:, 1].tolist()))
                        print(f"   {m} W{shp}: {nz} elements, max {float(blk.max()):.2e}, n rows {len(rows)} [{rows[0]}..{rows[-1]}], k cols {len(cols)} [{cols[0]}..{cols[-1]}]", flush=True)
                    else:
                        print(f"   {m} b{shp}: {nz} elements, max {float(blk.max()):.2e}", flush=True)
        T.learner.load_state_dict(R.learner.state_dict())
        t.cuda.synchronize()
print(f"{events} events in {N} calls of {K} updates ({prec})")
