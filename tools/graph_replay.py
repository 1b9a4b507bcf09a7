"""Config 2's step (plan + forward + backward) captured in a hipGraph and replayed 60 times -- for rocprofv3 --kernel-trace +
tools/timeline.py, next to the eager launches of `bench.py --no-graph` (why is the replay slower?)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from benchlib import Step  # noqa: E402

a = argparse.Namespace(config="c2", dmax=0.1, cutoff=0.0, fwd_only=False, force_dist=False, exchange="halo", overlap=False)
dev = torch.device("cuda:0")
step = Step(a, dev, 0, 1)
for _ in range(3):
    step()
torch.cuda.synchronize()
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    step()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        step()
torch.cuda.current_stream(dev).wait_stream(side)
for _ in range(60):
    g.replay()
torch.cuda.synchronize()
print("replayed")
