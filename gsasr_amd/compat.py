"""Make the reference's import paths resolve to this package (drop-in switch).

The reference imports the rasterizer as (utils/gaussian_splatting.py:87,101,120,134 and the BasicSR copy)

    from utils.gs_cuda.gswrapper import GSCUDA              from utils.gs_cuda_dmax.gswrapper import GSCUDA
    from basicsr.utils.gs_cuda.gswrapper import GSCUDA      from basicsr.utils.gs_cuda_dmax.gswrapper import GSCUDA
    import gscuda                                            (gs_cuda_dmax/gswrapper.py:19)

`install()` registers those module names in sys.modules, pointing at gsasr_amd's implementations, so
GSASR's encoder -> fea2gs -> splat -> HR-image code runs unchanged on MI355X.
"""
import sys
import types


def install(also_gaussian_splatting: bool = False) -> None:
    from . import gscuda
    from .gs_cuda import gswrapper as unbounded
    from .gs_cuda_dmax import gswrapper as bounded

    def _ensure_pkg(name: str):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []  # mark as package
            sys.modules[name] = m
        return sys.modules[name]

    sys.modules["gscuda"] = gscuda
    for root in ("utils", "basicsr.utils"):
        parts = root.split(".")
        for k in range(1, len(parts) + 1):
            _ensure_pkg(".".join(parts[:k]))
        for sub, mod in (("gs_cuda", unbounded), ("gs_cuda_dmax", bounded)):
            pkg = _ensure_pkg(f"{root}.{sub}")
            sys.modules[f"{root}.{sub}.gswrapper"] = mod
            setattr(pkg, "gswrapper", mod)
            setattr(sys.modules[root], sub, pkg)
        if also_gaussian_splatting:   # the host API and the tiled-inference driver built on it
            from . import gaussian_splatting, split_and_joint_image
            for name, mod in (("gaussian_splatting", gaussian_splatting), ("split_and_joint_image", split_and_joint_image)):
                sys.modules[f"{root}.{name}"] = mod
                setattr(sys.modules[root], name, mod)
