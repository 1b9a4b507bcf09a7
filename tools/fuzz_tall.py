#!/usr/bin/env python
"""The 32x64-px forward tile against the 32x16-px one on random large-scale shapes (GPU, development aid):
   python tools/fuzz_tall.py            # runs itself twice (GSASR_SPLAT_FWD_TALL=0 / 1) and compares the images"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = 10


def render(out_dir):
    import torch
    sys.path.insert(0, ROOT)
    from gsasr_amd import _cabi, synthetic
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    for case in range(CASES):
        h_lr, w_lr = int(rng.integers(60, 90)), int(rng.integers(60, 90))
        scale = float(rng.choice([24.0, 32.0, 40.0]))
        sig, xy, col, H, W = synthetic.kernel_inputs(h_lr, w_lr, scale, seed=100 + case)
        dmax = [0.1, None, 0.02][case % 3]
        r0 = int(rng.integers(0, H // 3)) if case % 2 else 0
        r1 = int(rng.integers(2 * H // 3, H)) if case % 2 else H
        a, b, c = sig.to(dev), xy.to(dev), col.to(dev)
        plan = _cabi.plan(a, b, c, H, W, dmax, rows=(r0, r1), flags=_cabi.FLAG_FORWARD_ONLY)
        img = torch.full((r1 - r0, W, 3), float("nan"), device=dev)
        _cabi.forward(plan, img, overwrite=True)
        np.save(os.path.join(out_dir, f"c{case}.npy"), img.cpu().numpy())
        print(f"case {case}: {H}x{W} rows [{r0},{r1}) N={sig.shape[0]} dmax={dmax} sub-tiles {((W + 7) // 8) * ((r1 - r0 + 15) // 16)}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        render(sys.argv[1])
        sys.exit(0)
    dirs = []
    for tall in ("0", "1"):
        d = f"/tmp/fuzz_tall_{tall}"
        os.makedirs(d, exist_ok=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), d], check=True, env=dict(os.environ, GSASR_SPLAT_FWD_TALL=tall))
        dirs.append(d)
    worst = 0.0
    for case in range(CASES):
        x, y = (np.load(os.path.join(d, f"c{case}.npy")) for d in dirs)
        assert np.isfinite(x).all() and np.isfinite(y).all(), case
        worst = max(worst, float(np.abs(x - y).max()) / max(1.0, float(np.abs(x).max())))
    print(f"{CASES} cases: worst relative difference tall vs plain tiles {worst:.2e}")
    assert worst <= 1e-5
