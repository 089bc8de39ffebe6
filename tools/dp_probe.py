"""Single-rank data-parallel rates (needs a GPU): RCCL path and the in-tile exchange path of an x2 learner."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch as t
import bench
dev = t.device("cuda", 0)
replay = bench.make_replay(dev, 0)
print(bench.dp_single_rank(dev, 0, replay, sys.argv[1] if len(sys.argv) > 1 else "x2", 3000))
