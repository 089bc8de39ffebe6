"""Each BASELINE config in each precision mode through step_n, one at a time (which one trips a bounded wait?)."""
import sys
sys.path.insert(0, "/root/repo")
import torch as t
import bench

dev = t.device("cuda", 0)
t.cuda.set_device(0)
only = sys.argv[1:] or None
for name, (cls_name, S_, A_, B_, extras, gflop, mbytes) in bench.BASELINE_CONFIGS.items():
    if only and not any(o in name for o in only):
        continue
    replay = bench.make_replay(dev, seed=0, S=S_, A=A_)
    for prec in ("x2", "f32", "bf16"):
        try:
            algo = bench._make_algo(cls_name, S_, A_, B_, extras, dev, prec)
            for n in (50, 200, 2000 if cls_name != "TQC" else 300):
                sec = bench._time_step_n(algo, replay, B_, n, dev)
            algo.learner.check()
            print(f"{name:32s} {prec:5s} {sec * 1e6:8.2f} us  {1 / sec:9.1f} steps/s", flush=True)
        except Exception as exc:  # noqa: BLE001
            print(f"{name:32s} {prec:5s} FAILED: {str(exc)[:200]}", flush=True)
        del algo
    del replay
