#!/usr/bin/env python
"""CPU-side cost of the host API calls (enqueue time only, GPU running asynchronously). Development aid."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import gaussian_splatting as gsp, synthetic  # noqa: E402

dev = torch.device("cuda:0")
p = synthetic.gs_parameters(256, 256, seed=0).to(dev)
H = W = 1024
sm = torch.tensor([4.0, 4.0], device=dev)
wgt = torch.rand(3, H, W, device=dev)


def step():
    pa = p.detach().requires_grad_(True)
    out = gsp.generate_2D_gaussian_splatting_step((H, W), pa, 4.0, sm, dmax=0.1)
    (out * wgt).sum().backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
