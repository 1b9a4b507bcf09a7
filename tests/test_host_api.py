"""CPU tests of the host side: L2 prologue vs reference captures, C-ABI exports, error behaviour."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest
import torch

from gsasr_amd import _cabi, gaussian_splatting as gsp, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
PROLOGUE = sorted(glob.glob(os.path.join(GOLDEN, "prologue_*.npz")))


@pytest.mark.parametrize("path", PROLOGUE, ids=[os.path.basename(p)[9:-4] for p in PROLOGUE])
def test_prologue_matches_reference_capture(path):
    """gsasr_amd's activations + kernel-frame conversion == what the reference hands to GSCUDA.apply."""
    z = np.load(path)
    sc = float(z["scale"])
    p = torch.from_numpy(z["gs_parameters"])
    step = gsp._step_size(sc, torch.tensor([sc, sc]), 1.2, "scale_modify")
    sx, sy, rho, xy, col = gsp._activate(p)
    sig, coords, colors, H, W = gsp._to_kernel_frame(sx, sy, rho, xy, col, z["sr_size"].tolist(), step)
    assert (H, W, 3) == tuple(z["img_shape"])
    np.testing.assert_allclose(sig.numpy(), z["sigmas"], rtol=1e-6)
    np.testing.assert_allclose(coords.numpy(), z["coords"], atol=1e-6)
    np.testing.assert_allclose(colors.numpy(), z["colors"], rtol=1e-6)
    if bool(z["if_dmax"]):
        got = gsp._resolve_dmax(float(z["dmax_in"]), str(z["dmax_mode"]), z["sr_size"].tolist())
        assert abs(got - float(z["dmax_out"])) < 1e-12


def test_prologue_accepts_tensor_sr_size_and_bad_modes():
    assert gsp._hw(torch.tensor([40, 52])) == (40, 52)
    with pytest.raises(ValueError):
        gsp._resolve_dmax(0.1, "bogus", (8, 8))
    with pytest.raises(AssertionError):
        gsp._step_size(4.0, torch.tensor([4.0, 3.0]), 1.2, "scale_modify")
    assert gsp._step_size(3.0, None, 1.2, "scale") == pytest.approx(0.4)


RP = sorted(glob.glob(os.path.join(GOLDEN, "rendering_python_n*.npz")))


@pytest.mark.parametrize("path", RP, ids=[os.path.basename(p)[17:-4] for p in RP])
def test_rendering_python_branch_matches_reference(path):
    z = np.load(path)
    sc = float(z["scale"])
    out = gsp.generate_2D_gaussian_splatting_step(z["sr_size"].tolist(), torch.from_numpy(z["gs_parameters"]), sc,
                                                  torch.tensor([sc, sc]), cuda_rendering=False)
    np.testing.assert_allclose(out.numpy(), z["out"], rtol=1e-4, atol=2e-5)


def test_sample_coords_gather():
    z = np.load(os.path.join(GOLDEN, "rendering_python_n48_36x44_s3.npz"))
    sc = float(z["scale"])
    pts = [(0, 0), (3, 7), (35, 43)]
    out = gsp.generate_2D_gaussian_splatting_step(z["sr_size"].tolist(), torch.from_numpy(z["gs_parameters"]), sc,
                                                  torch.tensor([sc, sc]), sample_coords=pts, cuda_rendering=False)
    assert out.shape == (3, 3)
    np.testing.assert_allclose(out[:, 1].numpy(), z["out"][:, 3, 7], rtol=1e-4, atol=2e-5)
    # the datasets hand an [S,2] integer tensor (continuous_bicubic_downsample_dataset.py:87-88): one gather,
    # same values and the same gradient scatter (repeated coordinates accumulate) as the reference's loop
    img = torch.rand(3, 20, 30, requires_grad=True)
    pts_t = torch.stack([torch.randint(0, 20, (50,)), torch.randint(0, 30, (50,))], dim=1)
    pts_t[5] = pts_t[3]
    a = gsp._sample(img, pts_t)
    b = torch.stack([img[:, c[0], c[1]] for c in pts_t], dim=1)
    assert torch.equal(a, b)
    assert torch.equal(torch.autograd.grad(a.sum(), img)[0], torch.autograd.grad(b.sum(), img)[0])


def test_cabi_exports_every_declared_symbol():
    """The shared library loads on a CPU-only box and exports exactly what include/*.h declares."""
    hdr = open(os.path.join(ROOT, "include", "gsasr_splat.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gsasr_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_cabi.EXPORTS), declared ^ set(_cabi.EXPORTS)
    L = ctypes.CDLL(_cabi.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _cabi.lib().gsasr_abi_version() == 7


def test_workspace_bytes_and_bad_dims_no_gpu():
    d = _cabi.make_dims(65536, 1024, 1024, 0.1, list_cap=-1)
    n = _cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(d))
    assert 65536 * 120 <= n <= 65536 * 160      # 136 B/Gaussian of scratch + per-cell tables
    # the tile lists are the default for DENSE plans only (>= 1 Gaussian per 4 pixels): none at one Gaussian per 16 pixels ...
    d = _cabi.make_dims(65536, 1024, 1024, 0.1)
    assert _cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(d)) == n
    # ... 2 048 tiles of 32 x 16 px, a 64-byte cursor line and `cap` 8-byte entries each, at 16 Gaussians per LR pixel
    dd = [_cabi.make_dims(1048576, 1024, 1024, 0.1, list_cap=c) for c in (-1, 0)]
    n16, nl = (_cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(x)) for x in dd)
    cap = (nl - n16 - 2048 * 64) // (2048 * 8)
    assert nl > n16 and 2048 <= cap <= 8192 and cap % 64 == 0
    d = _cabi.make_dims(65536, 1024, 1024, 0.1, list_cap=100)       # explicit capacity: rounded up to 64 entries
    assert _cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(d)) == n + 2048 * 64 + 2048 * 128 * 8
    bad = _cabi.make_dims(10, 1, 8, 0.1)        # h < 2: the grid 2*i/(h-1)-1 is undefined
    assert _cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(bad)) == 0
    bad = _cabi.make_dims(10, 8, 8, 0.1, rows=(4, 2))
    assert _cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(bad)) == 0
    # default support cutoff is adaptive: tau = ln(N / 1e-5) clamped to [16, 104]; explicit values pass through
    import math
    assert _cabi.get_default_cutoff() == 0.0
    assert abs(_cabi.resolve_cutoff(0.0, 65536) - math.log(65536 / 1e-5)) < 1e-4
    assert _cabi.resolve_cutoff(0.0, 1) == 16.0 and _cabi.resolve_cutoff(0.0, 2 ** 31 - 1) < 104.0
    assert _cabi.resolve_cutoff(32.0, 10) == 32.0 and _cabi.resolve_cutoff(-1.0, 10) == -1.0


def test_tensor_checks_raise_runtimeerror_like_reference():
    s = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="CUDA"):       # reference: CHECK_CUDA (gswrapper.cpp:5)
        _cabi.plan(s, torch.zeros(4, 2), torch.zeros(4, 3), 8, 8, 0.1)
    from gsasr_amd import gscuda
    with pytest.raises(RuntimeError, match="CUDA"):
        gscuda.gs_render(s, torch.zeros(4, 2), torch.zeros(4, 3), torch.zeros(8, 8, 3), 4, 8, 8, 3, 0.1)


def test_synthetic_inputs_shape_and_determinism():
    a = synthetic.kernel_inputs(16, 16, 4.0, seed=3)
    b = synthetic.kernel_inputs(16, 16, 4.0, seed=3)
    assert a[3:] == (64, 64) and a[0].shape == (256, 3)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)
    # means follow the LR raster: Gaussian k sits in LR pixel (k // 16, k % 16)
    xy = (a[1] + 1) / 2 * 64
    assert ((xy[:, 0] / 4).floor().clamp(0, 15).long() == torch.arange(256) % 16).float().mean() > 0.9


def test_compat_install_registers_reference_module_names():
    import sys
    from gsasr_amd import compat
    compat.install()
    from utils.gs_cuda_dmax.gswrapper import GSCUDA as A  # noqa: E402
    from basicsr.utils.gs_cuda.gswrapper import GSCUDA as B  # noqa: E402
    import gscuda  # noqa: E402
    from gsasr_amd.gs_cuda_dmax.gswrapper import GSCUDA as A0
    from gsasr_amd.gs_cuda.gswrapper import GSCUDA as B0
    assert A is A0 and B is B0 and hasattr(gscuda, "gs_render") and hasattr(gscuda, "gs_render_backward")
    compat.install(also_gaussian_splatting=True)       # host API + tiled-inference driver as well
    from utils.gaussian_splatting import generate_2D_gaussian_splatting_step as S  # noqa: E402
    from basicsr.utils.split_and_joint_image import split_and_joint_image as T  # noqa: E402
    import gsasr_amd.split_and_joint_image as tiled
    assert S is gsp.generate_2D_gaussian_splatting_step and T is tiled.split_and_joint_image
    for k in [k for k in sys.modules if k.split(".")[0] in ("utils", "basicsr", "gscuda")]:
        del sys.modules[k]


def test_config1_known_answer():
    """BASELINE.json config 1 in full: `rendering_python` on torch.manual_seed(0) inputs, 4 096 Gaussians ->
    256^2, x4, forward only -- mean 0.21793251, max 1.79732013 and the probes of tests/golden/
    rendering_python_config1_stats.npz, captured from the imported reference (make_golden.py:176-184).  Both this
    package's `cuda_rendering=False` branch and the oracle's restatement (the one bench.py times as `pytorch_path`)."""
    from oracle import host_ref
    z = np.load(os.path.join(GOLDEN, "rendering_python_config1_stats.npz"))
    torch.manual_seed(0)
    g = torch.randn(4096, 9)
    g[:, 7:9] = torch.rand(4096, 2)
    sm = torch.tensor([4.0, 4.0])
    for out in (gsp.generate_2D_gaussian_splatting_step((256, 256), g.clone(), 4.0, sm, cuda_rendering=False),
                host_ref.rendering_python(g.clone(), (256, 256), sm)):
        assert tuple(out.shape) == (3, 256, 256)
        assert float(out.mean()) == pytest.approx(0.21793251, abs=2e-6) and float(out.mean()) == pytest.approx(float(z["mean"]), abs=2e-6)
        assert float(out.max()) == pytest.approx(1.79732013, abs=2e-5) and float(out.max()) == pytest.approx(float(z["max"]), abs=2e-5)
        np.testing.assert_allclose(out.sum(dim=(0, 2)).numpy(), z["sum_rows"], rtol=2e-5, atol=1e-3)
        np.testing.assert_allclose(out[:, ::37, ::41].numpy(), z["probe"], rtol=1e-4, atol=2e-5)


def test_compat_install_keeps_real_packages_importable(tmp_path):
    """install() must not shadow the reference's own packages -- `utils.rdn`, `basicsr.models` ... keep
    importing after it, only the rasterizer leaves are replaced"""
    import subprocess
    import sys
    (tmp_path / "utils").mkdir()
    (tmp_path / "utils" / "__init__.py").write_text("")
    (tmp_path / "utils" / "rdn.py").write_text("MARK = 'real rdn'\n")
    (tmp_path / "utils" / "gs_cuda_dmax").mkdir()
    (tmp_path / "utils" / "gs_cuda_dmax" / "gswrapper.py").write_text("raise RuntimeError('the reference wrapper must not be imported')\n")
    (tmp_path / "basicsr").mkdir()
    (tmp_path / "basicsr" / "__init__.py").write_text("")
    (tmp_path / "basicsr" / "models.py").write_text("MARK = 'real models'\n")
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from gsasr_amd import compat; compat.install()\n"
        "import utils.rdn, basicsr.models\n"
        "assert utils.rdn.MARK == 'real rdn' and basicsr.models.MARK == 'real models'\n"
        "from utils.gs_cuda_dmax.gswrapper import GSCUDA as A\n"
        "from basicsr.utils.gs_cuda.gswrapper import GSCUDA as B\n"
        "import gsasr_amd.gs_cuda_dmax.gswrapper as m1, gsasr_amd.gs_cuda.gswrapper as m0, gscuda\n"
        "assert A is m1.GSCUDA and B is m0.GSCUDA and hasattr(gscuda, 'gs_render')\n"
        "print('ok')\n" % (ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr


def test_compat_install_rebinds_models_that_import_at_package_import(tmp_path):
    """the real `basicsr/__init__.py` does `from .models import *`, which imports gsasr_model.py, which binds
    `from basicsr.utils.gaussian_splatting import generate_2D_gaussian_splatting_step` at import time.  install() must have
    the leaf registered BEFORE it imports the parent packages, or the model keeps the reference's function.  A tree of
    that shape (TrainTestGSASR/basicsr/__init__.py, models/__init__.py, models/gsasr_model.py:10), and a package whose
    import fails for another reason (missing dependency), which must surface instead of being replaced by a stub."""
    import subprocess
    import sys
    b = tmp_path / "basicsr"
    (b / "models").mkdir(parents=True)
    (b / "utils").mkdir()
    (b / "__init__.py").write_text("from .models import *\n")
    (b / "models" / "__init__.py").write_text("from .gsasr_model import GSASRModel\n__all__ = ['GSASRModel']\n")
    (b / "models" / "gsasr_model.py").write_text(
        "from basicsr.utils.gaussian_splatting import generate_2D_gaussian_splatting_step\n"
        "from basicsr.utils.split_and_joint_image import split_and_joint_image\n"
        "class GSASRModel:\n    step = staticmethod(generate_2D_gaussian_splatting_step)\n")
    (b / "utils" / "__init__.py").write_text("")
    (b / "utils" / "gaussian_splatting.py").write_text("def generate_2D_gaussian_splatting_step(*a, **k):\n    raise RuntimeError('reference host path')\n")
    (b / "utils" / "split_and_joint_image.py").write_text("def split_and_joint_image(*a, **k):\n    raise RuntimeError('reference tiled driver')\n")
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from gsasr_amd import compat, gaussian_splatting as gsp; compat.install(also_gaussian_splatting=True)\n"
        "import basicsr.models.gsasr_model as m\n"
        "assert m.generate_2D_gaussian_splatting_step is gsp.generate_2D_gaussian_splatting_step, m.generate_2D_gaussian_splatting_step\n"
        "assert m.GSASRModel.step is gsp.generate_2D_gaussian_splatting_step\n"
        "import basicsr.utils as u; assert u.gaussian_splatting is gsp\n"
        "print('ok')\n" % (ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr
    # a parent package that exists but cannot be imported: the real error, not a silent empty stand-in
    (b / "utils" / "__init__.py").write_text("import a_dependency_that_is_not_installed\n")
    code2 = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "from gsasr_amd import compat\n"
             "try:\n    compat.install()\nexcept ModuleNotFoundError as e:\n    assert e.name == 'a_dependency_that_is_not_installed', e.name\n    print('ok')\n"
             % (ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
    # ... and nothing of the failed install() stays behind: no rasterizer leaf of this package under the
    # reference's names, no partly imported parent, `gscuda` not registered
    code3 = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "from gsasr_amd import compat\n"
             "try:\n    compat.install(also_gaussian_splatting=True)\nexcept ModuleNotFoundError:\n    pass\n"
             "left = [k for k in sys.modules if k == 'gscuda' or k.startswith(('basicsr', 'utils'))]\n"
             "assert not left, left\nprint('ok')\n" % (ROOT, str(tmp_path)))
    r = subprocess.run([sys.executable, "-c", code3], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


def test_default_backward_rule_by_workspace_size():
    """which plans carry slots for the tile-stationary backward by default (bwd_wants_tile, gsasr_amd/csrc/splat_common.h): x8 and up
    from 8 Mpx; since round 5 one Gaussian per 2..4 px from 1 Mpx (a dense plan whose tile lists that kernel reads).  Seen from the
    host as the workspace a plan asks for, against the same dims with GSASR_FLAG_BWD_GAUSSIAN."""
    import ctypes
    from gsasr_amd import _cabi

    def slots(s, h, w, **kw):
        a = _cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(_cabi.make_dims(s, h, w, 0.1, **kw)))
        b = _cabi.lib().gsasr_splat_workspace_bytes(ctypes.byref(_cabi.make_dims(s, h, w, 0.1, flags=_cabi.FLAG_BWD_GAUSSIAN, **kw)))
        assert a >= b
        return a - b >= s * 32 * 8
    assert slots(512 * 512, 1024, 1024)                  # x2, one Gaussian per LR pixel: 4 px per Gaussian on 1 Mpx
    assert slots(4 * 256 * 256, 1024, 1024)              # x4 at four per LR pixel
    assert not slots(256 * 256, 512, 512)                # x2 on 512^2: too few tiles
    assert not slots(256 * 256, 1024, 1024)              # config 2: 16 px per Gaussian
    assert not slots(16 * 256 * 256, 1024, 1024)         # 16 per LR pixel at x4: 1 px per Gaussian
    assert not slots(512 * 512, 1024, 1024, rows=(0, 512))      # a row band: the caller decides (shard.py sets the flag)
    assert not slots(512 * 512, 1024, 1024, list_cap=-1)        # no lists, no tile kernel at this density
    assert slots(1024 * 1024, 8192, 8192)                # config 4


def test_deferred_assert_exit_runs_later_handlers_once():
    """A deferred check that fails at interpreter exit fails the PROCESS (exit status 1) after every atexit handler registered
    later than the package's import has run exactly once (round 5 re-ran atexit's list from inside the handler: ADVICE r5)."""
    import subprocess
    import sys
    code = (
        "import atexit, sys\n"
        "sys.path.insert(0, %r)\n"
        "from gsasr_amd import _deferred\n"
        "def boom():\n"
        "    raise AssertionError('scale_modify is not the same')\n"
        "_deferred.deferred_asserts.flush = boom\n"
        "atexit.register(lambda: print('later-handler', flush=True))\n"
        "print('main-done', flush=True)\n"
    ) % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stdout, r.stderr)
    assert r.stdout.count("later-handler") == 1 and "main-done" in r.stdout
    assert "deferred check failed at exit" in r.stderr
    env = dict(os.environ, GSASR_AMD_DEFERRED_EXIT="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and r.stdout.count("later-handler") == 1
