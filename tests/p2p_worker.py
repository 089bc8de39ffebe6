"""Worker of tests/test_gpu_p2p.py: one data-parallel rank.  Both ranks of the test share GPU 0 (the
peer windows are ordinary IPC mappings, so two processes on one device exercise the whole protocol:
handle exchange, pushes into the other process's window, flags, window-half reuse); rendezvous and the
handle exchange go over gloo."""
import os
import sys

import torch as t
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> None:
    rank, world, rdv, out, algo_name, K, B, level = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4],
                                                      sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8]))
    prec = sys.argv[9] if len(sys.argv) > 9 else "f32"
    if level == 2 and algo_name == "ddpg":
        # two ranks share this GPU: one update per launch (k_ddpg_chain with several would queue the NEXT update's
        # workgroups of the rank that launched first into every free compute unit, and they wait — through its tiles'
        # exchange — for the other rank's workgroups, which then find none: ranks of a real job own their GPU)
        os.environ["OPRL_AMD_CHAIN"] = "1"
    dist.init_process_group("gloo", init_method=f"file://{rdv}", rank=rank, world_size=world)
    from oprl_amd.logging import NullLogger
    from oprl_amd.parallel import DataParallelLearner
    from tests.test_gpu_p2p import make_algo, make_shard
    algo = make_algo(algo_name, B, export_grads=True, precision=prec)
    buf = make_shard(rank)
    dp = DataParallelLearner(algo, dist.group.WORLD)
    ok = dp.init_p2p(level)
    if ok:
        dp.step_n(buf.handle, K, B, seed=5)
        t.cuda.synchronize()
    arenas = {m: getattr(algo, m)._oprl_arena.cpu() for m in ("actor", "critic")}
    t.save({"ok": ok, "why": dp.p2p_error, "arenas": arenas, "alpha": getattr(algo, "alpha", None),
            "form": algo.learner.debug_form(B)}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
