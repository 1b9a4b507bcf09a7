"""gsasr_amd.tune on Gaussians that are NOT "about one LR pixel": for a grid of shapes x size distributions, what the library's own
rule costs, which combination tune() picks, and what it gains.   python tools/tune_demo.py > gpurun_out/tune_demo.txt   (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gsasr_amd import synthetic, tune  # noqa: E402

SHAPES = [("x4 1024^2", 256, 4, 1), ("x4 512^2 16/LR px", 128, 4, 16), ("x4 1024^2 16/LR px", 256, 4, 16), ("x2 1024^2", 512, 2, 1),
          ("x8 3072^2", 384, 8, 1), ("x12 3072^2", 256, 12, 1)]
DISTS = ["SURVEY 8(d)", "small (x0.4)", "saturated (x1.9)", "log-uniform", "needles", "bimodal"]


def reshape(sig, dist, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = sig.shape[0]
    r = torch.rand(n, 3, generator=g).to(sig.device)
    s = sig.clone()
    if dist == 1:
        s[:, :2] *= 0.4
    elif dist == 2:
        s[:, :2] *= 1.9
    elif dist == 3:
        s[:, :2] *= 0.1 * torch.pow(torch.tensor(20.0, device=sig.device), r[:, :2])
    elif dist == 4:
        flip = r[:, 2] < 0.5
        s[:, 0] *= torch.where(flip, 1.8, 0.2)
        s[:, 1] *= torch.where(flip, 0.2, 1.8)
        s[:, 2] = torch.where(r[:, 0] < 0.5, -0.9, 0.9) * (0.5 + 0.5 * r[:, 1])
    elif dist == 5:
        s[:, :2] *= torch.where(r[:, 2] < 0.9, 0.3, 1.9)[:, None]
    return s.contiguous()


def main():
    dev = torch.device("cuda:0")
    print(f"{'shape':22s} {'distribution':18s} {'default ms':>10s}  {'tune() picks':28s} {'ms':>8s} {'gain':>7s}")
    for name, lr, scale, gpp in SHAPES:
        sig0, xy, col, H, W = synthetic.kernel_inputs(lr, lr, scale, seed=0, gpp=gpp, device=dev)
        for di, dn in enumerate(DISTS):
            sig = reshape(sig0, di, 11 + di)
            res = tune.tune(sig, xy, col, H, W, 0.1, register=False)
            d = res.ms["default"]
            print(f"{name:22s} {dn:18s} {d:10.4f}  {res.name:28s} {res.ms[res.name]:8.4f} {100 * (1 - res.ms[res.name] / d):6.1f}%", flush=True)
        tune.reset()


if __name__ == "__main__":
    main()
