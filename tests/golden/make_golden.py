"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (read-only, /root/reference).

Run in the authoring container only (the reference tree does not exist on the GPU box):

    python tests/golden/make_golden.py

What is captured (data only -- inputs and the reference's outputs; no reference source is copied):

 * raster_*.npz       `torch_version` from utils/gs_cuda/check.py:4-27 (unbounded) and
                      utils/gs_cuda_dmax/check.py:4-31 (dmax box): image, and d(sum(weight*img))/d{sigmas,
                      coords,colors} by autograd through it.  fp32 run and fp64-input run of the same case.
 * prologue_*.npz     the tensors `generate_2D_gaussian_splatting_step` hands to `GSCUDA.apply`
                      (utils/gaussian_splatting.py:158-213 -> :86-98 / :119-131), captured by replacing the
                      two `gswrapper` modules in sys.modules with a recorder, so the reference host code
                      runs unmodified on CPU.
 * rendering_python_*.npz  output of the `cuda_rendering=False` branch (utils/gaussian_splatting.py:11-84).
 * tiled_*.npz        `split_and_joint_image` (utils/split_and_joint_image.py:98-232) with the deterministic
                      stand-in encoder/decoder of tests/tiled_models.py and `cuda_rendering=False`: LR input,
                      arguments and the stitched SR output, for an integer and a fractional scale factor.

Stubs needed because this image has no torchvision / CUDA: dummy `torchvision`, `torchvision.utils`
(imported but unused by utils/gaussian_splatting.py:7-8) and a dummy `gswrapper` (check.py:2 would
otherwise JIT-compile CUDA).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("GSASR_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub_modules():
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tvu.save_image = lambda *a, **k: None
    tv.utils = tvu
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.utils", tvu)
    gw = types.ModuleType("gswrapper")
    gw.gaussiansplatting_render = None
    sys.modules["gswrapper"] = gw


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def raster_case(tv_fn, name, sigmas, coords, colors, hw, dmax=None, wseed=1):
    """Run torch_version in fp32 and with fp64 inputs; store image + autograd grads."""
    out = {"sigmas": sigmas.numpy(), "coords": coords.numpy(), "colors": colors.numpy(),
           "h": hw[0], "w": hw[1], "dmax": np.float64(-1.0 if dmax is None else dmax)}
    torch.manual_seed(wseed)
    wgt = torch.rand(hw[0], hw[1], 3)
    out["weight"] = wgt.numpy()
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        s = sigmas.to(dt).clone().requires_grad_(True)
        c = coords.to(dt).clone().requires_grad_(True)
        k = colors.to(dt).clone().requires_grad_(True)
        args = (s, c, k, hw) if dmax is None else (s, c, k, hw, dmax)
        img = tv_fn(*args)
        (wgt.to(img.dtype) * img).sum().backward()
        out[f"img_{tag}"] = img.detach().numpy()
        out[f"g_sigmas_{tag}"] = s.grad.numpy()
        out[f"g_coords_{tag}"] = c.grad.numpy()
        out[f"g_colors_{tag}"] = k.grad.numpy()
    np.savez_compressed(os.path.join(OUT, f"raster_{name}.npz"), **out)
    print(name, "img.sum", float(out["img_f32"].sum()), "gs.sum", float(out["g_sigmas_f32"].sum()))


def tiled_section():
    """the tiled-inference goldens only (`python make_golden.py --tiled`)"""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for modname in ("utils.gs_cuda.gswrapper", "utils.gs_cuda_dmax.gswrapper"):   # (never called with cuda_rendering=False)
        if modname not in sys.modules:
            m = types.ModuleType(modname)
            m.GSCUDA = None
            sys.modules[modname] = m
    # ---- tiled inference driver (utils/split_and_joint_image.py) -----------------------------------------
    import utils.split_and_joint_image as reftile  # noqa: E402  (the reference, unmodified)
    sys.path.insert(0, os.path.dirname(OUT))
    import tiled_models  # noqa: E402
    # (round 3: six more shapes -- other tile grids incl. a single row / column of tiles, crop 0, scales whose ceil()
    # rounding of the SR tile and overlap sizes differs; `--tiled` regenerates this section only)
    for name, (hl, wl), sc, split, overlap, crop in (("int_s2_20x26", (20, 26), 2.0, 8, 2, 2),
                                                     ("frac_s2p5_18x22", (18, 22), 2.5, 8, 3, 1),
                                                     ("int_s3_17x31", (17, 31), 3.0, 7, 2, 1),
                                                     ("frac_s1p7_25x19", (25, 19), 1.7, 9, 3, 2),
                                                     ("frac_s3p3_30x16", (30, 16), 3.3, 10, 3, 0),
                                                     ("int_s4_12x40_onerow", (12, 40), 4.0, 11, 2, 3),
                                                     ("frac_s2p25_40x13_onecol", (40, 13), 2.25, 12, 5, 1),
                                                     ("frac_s1p2_33x35", (33, 35), 1.2, 6, 1, 0)):
        torch.manual_seed(21)
        lq = torch.rand(1, 3, hl, wl)
        out = reftile.split_and_joint_image(lq, sc, split, overlap, tiled_models.model_g, tiled_models.model_fea2gs,
                                            torch.tensor([sc, sc]), crop_size=crop, cuda_rendering=False)
        np.savez_compressed(os.path.join(OUT, f"tiled_{name}.npz"), lq=lq.numpy(), scale=np.float64(sc),
                            split_size=split, overlap_size=overlap, crop_size=crop, out=out.numpy())
        print("tiled", name, tuple(out.shape), float(out.mean()))


def main():
    _stub_modules()
    if "--tiled" in sys.argv:
        return tiled_section()
    only_python = "--python" in sys.argv
    case = (lambda *a, **k: None) if only_python else raster_case
    chk0 = _load(os.path.join(REF, "utils/gs_cuda/check.py"), "ref_check_unbounded")
    chk1 = _load(os.path.join(REF, "utils/gs_cuda_dmax/check.py"), "ref_check_dmax")

    def rnd(s, seed=0, sig_scale=1.0):
        torch.manual_seed(seed)
        sigmas = 0.999 * torch.rand(s, 3)
        coords = 2 * torch.rand(s, 2) - 1.0
        colors = torch.rand(s, 3)
        sigmas[:, :2] *= sig_scale
        return sigmas, coords, colors

    # the reference's own check.py sizes (SURVEY.md 8c known answers)
    case(chk0.torch_version, "unbounded_s40_49x49", *rnd(40), (49, 49))
    s, c, k = rnd(4, sig_scale=5.0)
    case(chk1.torch_version, "dmax0p5_s4_10x10_sig5", s, c, k, (10, 10), dmax=0.5)
    case(chk1.torch_version, "dmax0p5_s40_49x49", *rnd(40), (49, 49), dmax=0.5)
    # box edges active: small dmax, non-square image
    case(chk1.torch_version, "dmax0p1_s40_49x49", *rnd(40, seed=2), (49, 49), dmax=0.1)
    case(chk1.torch_version, "dmax0p25_s24_31x47", *rnd(24, seed=3, sig_scale=0.3), (31, 47), dmax=0.25)
    case(chk0.torch_version, "unbounded_s24_31x47", *rnd(24, seed=3, sig_scale=0.3), (31, 47))
    # degenerate / edge inputs: a Gaussian outside the image, |rho| near the activation limit, tiny sigma,
    # a Gaussian exactly on a pixel, one exactly dmax away from a pixel column
    s, c, k = rnd(12, seed=4, sig_scale=0.2)
    c[0] = torch.tensor([1.7, -1.4])                # outside the image
    c[1] = torch.tensor([-1.0, 1.0])                # on the corner pixel
    s[2, 2] = 0.9990234375                          # rho with exactly representable square is not needed; large |rho|
    s[3, 2] = -0.99609375
    s[4, :2] = torch.tensor([1e-3, 2e-3])           # very narrow
    s[5, :2] = torch.tensor([0.9, 0.02])            # very anisotropic
    c[6] = torch.tensor([2 * 10 / 32 - 1.0 + 0.25, 0.0])   # column 10 of a 33-wide grid is exactly dmax away
    case(chk1.torch_version, "dmax0p25_edge_s12_29x33", s, c, k, (29, 33), dmax=0.25)
    case(chk0.torch_version, "unbounded_edge_s12_29x33", s, c, k, (29, 33))
    # (sizes below 50: check.py:9 logs through an undefined `logger` from 50 px up)
    # round 3: a thin wide image, a thin tall one, strongly correlated Gaussians, a sub-pixel dmax box, negative sigmas
    # (only sigma^2 and 1/(sx sy) enter the reference's formulas)
    case(chk1.torch_version, "dmax0p3_thin_s30_5x49", *rnd(30, seed=5, sig_scale=0.15), (5, 49), dmax=0.3)
    case(chk0.torch_version, "unbounded_tall_s30_49x4", *rnd(30, seed=6, sig_scale=0.2), (49, 4))
    s, c, k = rnd(24, seed=7, sig_scale=0.25)
    s[:, 2] = torch.tensor([0.9999, -0.9999, 0.999, -0.999, 0.99, -0.99] * 4)
    case(chk1.torch_version, "dmax0p4_rho_s24_33x37", s, c, k, (33, 37), dmax=0.4)
    case(chk0.torch_version, "unbounded_rho_s24_33x37", s, c, k, (33, 37))
    case(chk1.torch_version, "dmax0p02_s40_41x43", *rnd(40, seed=8, sig_scale=0.5), (41, 43), dmax=0.02)
    s, c, k = rnd(16, seed=9, sig_scale=0.3)
    s[::2, 0] *= -1.0
    s[::3, 1] *= -1.0
    case(chk1.torch_version, "dmax0p5_negsigma_s16_31x29", s, c, k, (31, 29), dmax=0.5)

    # ---- L2 prologue captures -------------------------------------------------------------------------
    sys.path.insert(0, REF)
    rec = {}

    class _Rec:
        @staticmethod
        def apply(sigmas, coords, colors, rendered_img, dmax=None):
            rec["sigmas"], rec["coords"], rec["colors"] = sigmas.clone(), coords.clone(), colors.clone()
            rec["shape"], rec["dmax"] = tuple(rendered_img.shape), dmax
            return rendered_img

    for modname in ("utils.gs_cuda.gswrapper", "utils.gs_cuda_dmax.gswrapper"):
        m = types.ModuleType(modname)
        m.GSCUDA = _Rec
        sys.modules[modname] = m
    import utils.gaussian_splatting as refgs  # noqa: E402  (the reference, unmodified)

    torch.manual_seed(7)
    gsp = torch.randn(64, 9)
    gsp[:, 7:9] = torch.rand(64, 2)
    cases = [("s4_40x52_fix", (40, 52), 4.0, True, "fix", 0.1),
             ("s2p5_40x52_dyn", (40, 52), 2.5, True, "dynamic", 25),
             ("s4_40x52_unbounded", (40, 52), 4.0, False, "fix", 25),
             ("s3_33x33_fix", [33, 33], 3.0, True, "fix", 0.5),
             # round 3: a wide and a tall image (the align-corners shift of the centres at other magnitudes), a fractional
             # scale whose step is not exactly representable, sr_size as the tensor the tiled driver passes
             ("s3p3_37x1500_fix", (37, 1500), 3.3, True, "fix", 0.05),
             ("s6p6_2000x24_dyn", torch.tensor([2000, 24]), 6.6, True, "dynamic", 40),
             ("s1p2_64x64_unbounded", (64, 64), 1.2, False, "fix", 25)]
    for name, sr, sc, if_dmax, mode, dmax in ([] if only_python else cases):
        rec.clear()
        out = refgs.generate_2D_gaussian_splatting_step(sr, gsp.clone(), sc, torch.tensor([sc, sc]),
                                                        cuda_rendering=True, if_dmax=if_dmax,
                                                        dmax_mode=mode, dmax=dmax)
        np.savez_compressed(os.path.join(OUT, f"prologue_{name}.npz"), gs_parameters=gsp.numpy(),
                            sr_size=np.array([int(v) for v in sr]), scale=np.float64(sc), if_dmax=if_dmax, dmax_mode=mode,
                            dmax_in=np.float64(dmax), sigmas=rec["sigmas"].numpy(), coords=rec["coords"].numpy(),
                            colors=rec["colors"].numpy(), img_shape=np.array(rec["shape"]),
                            dmax_out=np.float64(-1.0 if rec["dmax"] is None else float(rec["dmax"])),
                            out_shape=np.array(out.shape))
        print("prologue", name, rec["shape"], rec["dmax"], tuple(out.shape))

    # ---- rendering_python (cuda_rendering=False) --------------------------------------------------------
    # (round 3: scales 2.7 and 6.6, where `int(10 * 2 / step_size)` on the fp32 step TENSOR -- 45 and 110 -- differs from
    # the same expression on a Python float -- 44 and 109; `--python` regenerates this section only)
    for name, sr, sc, n in (("n64_40x40_s4", (40, 40), 4.0, 64), ("n48_36x44_s3", (36, 44), 3.0, 48),
                            ("n24_27x33_s2p7", (27, 33), 2.7, 24), ("n20_33x40_s6p6", (33, 40), 6.6, 20)):
        torch.manual_seed(11)
        g = torch.randn(n, 9)
        g[:, 7:9] = torch.rand(n, 2)
        out = refgs.generate_2D_gaussian_splatting_step(sr, g.clone(), sc, torch.tensor([sc, sc]),
                                                        cuda_rendering=False)
        np.savez_compressed(os.path.join(OUT, f"rendering_python_{name}.npz"), gs_parameters=g.numpy(),
                            sr_size=np.array(sr), scale=np.float64(sc), out=out.numpy())
        print("rendering_python", name, float(out.mean()), float(out.max()))

    if only_python:
        return
    tiled_section()

    # BASELINE.json config 1 known answer: statistics only (the tensor is 768 KB)
    torch.manual_seed(0)
    g = torch.randn(4096, 9)
    g[:, 7:9] = torch.rand(4096, 2)
    out = refgs.generate_2D_gaussian_splatting_step((256, 256), g.clone(), 4.0, torch.tensor([4.0, 4.0]),
                                                    cuda_rendering=False)
    np.savez_compressed(os.path.join(OUT, "rendering_python_config1_stats.npz"), mean=np.float64(out.mean()),
                        max=np.float64(out.max()), sum_rows=out.sum(dim=(0, 2)).numpy(),
                        probe=out[:, ::37, ::41].numpy())
    print("config1", float(out.mean()), float(out.max()))


if __name__ == "__main__":
    main()
