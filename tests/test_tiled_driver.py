"""Tiled-inference driver: gsasr_amd.split_and_joint_image against golden outputs of the
REFERENCE's utils/split_and_joint_image.py:98-232 (tests/golden/make_golden.py, stand-in models of tiled_models.py)."""
import glob
import os

import numpy as np
import pytest
import torch

import tiled_models
from gsasr_amd.split_and_joint_image import _paste_rule, split_and_joint_image

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiled_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[6:-4] for p in GOLDEN])
def test_tiled_driver_matches_reference_cpu(path):
    """`cuda_rendering=False` (the reference's own CPU branch): tiling, reflect padding, per-tile models and the
    pasting rule -- including the fractional-scale irregularity -- reproduce the reference's canvas"""
    z = np.load(path)
    sc = float(z["scale"])
    out = split_and_joint_image(torch.from_numpy(z["lq"]), sc, int(z["split_size"]), int(z["overlap_size"]),
                                tiled_models.model_g, tiled_models.model_fea2gs, torch.tensor([sc, sc]),
                                crop_size=int(z["crop_size"]), cuda_rendering=False)
    assert tuple(out.shape) == z["out"].shape
    np.testing.assert_allclose(out.numpy(), z["out"], rtol=1e-5, atol=1e-6)


def test_paste_rule_and_argument_checks():
    assert len(GOLDEN) == 8
    assert _paste_rule(0, 0, 3, 4, 2, True) == (0, 0) and _paste_rule(0, 2, 3, 4, 2, False) == (0, 2)
    assert _paste_rule(1, 1, 3, 4, 2, True) == (2, 2) and _paste_rule(2, 3, 3, 4, 2, True) == (2, 2)
    # the reference's irregularity (fractional scale only): last column / last row, interior otherwise
    assert _paste_rule(1, 3, 3, 4, 2, True) == (0, 2) and _paste_rule(2, 1, 3, 4, 2, True) == (2, 0)
    assert _paste_rule(1, 3, 3, 4, 2, False) == (2, 2) and _paste_rule(2, 1, 3, 4, 2, False) == (2, 2)
    lq = torch.rand(1, 3, 20, 20)
    with pytest.raises(AssertionError, match="overlap size is wrong"):
        split_and_joint_image(lq, 2.0, 8, 4, tiled_models.model_g, tiled_models.model_fea2gs, torch.tensor([2.0, 2.0]),
                              cuda_rendering=False)
    with pytest.raises(AssertionError, match="please decrease the split_size"):
        split_and_joint_image(torch.rand(1, 3, 5, 20), 2.0, 16, 2, tiled_models.model_g, tiled_models.model_fea2gs,
                              torch.tensor([2.0, 2.0]), cuda_rendering=False)


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [2.0, 2.5])
def test_tiled_driver_on_gpu_batches_the_rasterizer(scale):
    """on the GPU all tiles go through batched canvases; the result equals pasting per-tile renders"""
    from gsasr_amd import gaussian_splatting as gsp
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    lq = torch.rand(1, 3, 40, 52, device=dev)
    sm = torch.tensor([scale, scale], device=dev)
    kw = dict(if_dmax=True, dmax_mode="fix", dmax=0.4)
    out = split_and_joint_image(lq, scale, 12, 3, tiled_models.model_g, tiled_models.model_fea2gs, sm, crop_size=2, **kw)
    calls = []
    orig = gsp.generate_2D_gaussian_splatting_batch

    def one_by_one(sizes, params, scales, sms, **k):      # the same driver with the batch taken apart
        calls.append(len(sizes))
        return torch.stack([gsp.generate_2D_gaussian_splatting_step(sizes[b], params[b], scales[b], sms[b], **k)
                            for b in range(len(sizes))])
    import gsasr_amd.split_and_joint_image as drv
    drv.generate_2D_gaussian_splatting_batch = one_by_one
    try:
        ref = split_and_joint_image(lq, scale, 12, 3, tiled_models.model_g, tiled_models.model_fea2gs, sm, crop_size=2, **kw)
    finally:
        drv.generate_2D_gaussian_splatting_batch = orig
    assert calls and sum(calls) == 5 * 6 and out.shape == ref.shape      # ceil(37/9) x ceil(49/9) tiles
    assert float((out - ref).abs().max()) <= 2e-6
