"""Make the reference's import paths resolve to this package (drop-in switch).

The reference imports the rasterizer as (utils/gaussian_splatting.py:87,101,120,134 and the BasicSR copy)

    from utils.gs_cuda.gswrapper import GSCUDA              from utils.gs_cuda_dmax.gswrapper import GSCUDA
    from basicsr.utils.gs_cuda.gswrapper import GSCUDA      from basicsr.utils.gs_cuda_dmax.gswrapper import GSCUDA
    import gscuda                                            (gs_cuda_dmax/gswrapper.py:19)

`install()` registers exactly those LEAF module names in sys.modules, pointing at gsasr_amd's implementations, so
GSASR's encoder -> fea2gs -> splat -> HR-image code runs unchanged on MI355X.  The parent packages (`utils`,
`basicsr`, `basicsr.utils`, `utils.gs_cuda*`) are the REAL ones whenever they can be imported -- every other reference
module (`utils.rdn`, `utils.fea2gs`, `basicsr.models`, ...) keeps importing normally; an empty stand-in package is
created only for a parent that does not exist on sys.path (e.g. when only the rasterizer is wanted).  Call it after
sys.path is set up for the reference tree and before the reference's modules import the rasterizer.
"""
import importlib
import sys
import types


def _parent(name: str):
    """the real package `name` if importable, else an empty stand-in (registered in sys.modules)"""
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except ImportError:
        m = types.ModuleType(name)
        m.__path__ = []            # a package with nothing in it besides what install() attaches
        m.__gsasr_amd_stub__ = True
        sys.modules[name] = m
        if "." in name:
            setattr(sys.modules[name.rsplit(".", 1)[0]], name.rsplit(".", 1)[1], m)
        return m


def install(also_gaussian_splatting: bool = False) -> None:
    from . import gscuda
    from .gs_cuda import gswrapper as unbounded
    from .gs_cuda_dmax import gswrapper as bounded

    sys.modules["gscuda"] = gscuda
    for root in ("utils", "basicsr.utils"):
        parts = root.split(".")
        for k in range(1, len(parts) + 1):
            _parent(".".join(parts[:k]))
        for sub, mod in (("gs_cuda", unbounded), ("gs_cuda_dmax", bounded)):
            pkg = _parent(f"{root}.{sub}")
            sys.modules[f"{root}.{sub}.gswrapper"] = mod      # the leaf: never the reference's JIT-compiling CUDA wrapper
            setattr(pkg, "gswrapper", mod)
        if also_gaussian_splatting:   # the host API and the tiled-inference driver built on it
            from . import gaussian_splatting, split_and_joint_image
            for name, mod in (("gaussian_splatting", gaussian_splatting), ("split_and_joint_image", split_and_joint_image)):
                sys.modules[f"{root}.{name}"] = mod
                setattr(sys.modules[root], name, mod)
