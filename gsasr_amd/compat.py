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
    """the real package `name` if it exists, else an empty stand-in (registered in sys.modules).  Only "there is no such
    package" makes a stand-in: any other failure while importing a package that DOES exist (a missing dependency such
    as cv2 inside basicsr.utils, a syntax error, ...) is the user's real problem and is raised as it is."""
    if name in sys.modules:
        return sys.modules[name]
    try:
        return importlib.import_module(name)
    except ModuleNotFoundError as e:
        # e.name is the module that could not be found: the package itself (or one of its parents) => stand-in
        if e.name is None or not (name == e.name or name.startswith(e.name + ".")):
            raise
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

    # 1. The LEAF names first, before any parent package is imported: importing the real `basicsr` runs
    #    `from .models import *`, which imports gsasr_model.py, which does `from basicsr.utils.gaussian_splatting import
    #    generate_2D_gaussian_splatting_step` -- the import machinery imports the parents and then finds the leaf already
    #    registered here, so the model classes bind THIS package's functions (registered afterwards, they would have bound
    #    the reference's unfused host path for good).
    leaves = {}
    for root in ("utils", "basicsr.utils"):
        leaves[f"{root}.gs_cuda.gswrapper"] = unbounded      # never the reference's JIT-compiling CUDA wrappers
        leaves[f"{root}.gs_cuda_dmax.gswrapper"] = bounded
        if also_gaussian_splatting:   # the host API and the tiled-inference driver built on it
            from . import gaussian_splatting, split_and_joint_image
            leaves[f"{root}.gaussian_splatting"] = gaussian_splatting
            leaves[f"{root}.split_and_joint_image"] = split_and_joint_image
    before = {k: sys.modules.get(k) for k in ("gscuda", *leaves)}
    known = set(sys.modules)
    sys.modules["gscuda"] = gscuda
    sys.modules.update(leaves)
    # 2. The parents: the real packages where they exist, and the leaves attached to them as attributes.  If a parent
    #    that exists fails to import (a missing dependency inside basicsr.utils, say) nothing of this call stays behind: a
    #    caller that catches the error must not be left with this package's leaves under half-imported parents.
    try:
        for name, mod in leaves.items():
            parts = name.split(".")
            for k in range(1, len(parts)):
                _parent(".".join(parts[:k]))
            setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
            sys.modules[name] = mod       # (a parent's own __init__ may have imported and re-registered its leaf meanwhile)
    except BaseException:
        for k, v in before.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        roots = ("utils", "basicsr")
        for k in [k for k in sys.modules if k not in known and (k in roots or k.startswith(tuple(r + "." for r in roots)))]:
            sys.modules.pop(k, None)      # stand-ins and partly imported parents this call brought in
        raise
