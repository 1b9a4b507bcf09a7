"""gsasr_amd -- MI355X-native 2D Gaussian-splatting rasterizer for GSASR (the one hot path of
ChrisDud0257/GSASR, behind the reference's own autograd/operator surface).

    from gsasr_amd.gaussian_splatting import generate_2D_gaussian_splatting_step   # utils/gaussian_splatting.py
    from gsasr_amd.gs_cuda_dmax.gswrapper import GSCUDA, gaussiansplatting_render  # utils/gs_cuda_dmax/gswrapper.py
    from gsasr_amd.gs_cuda.gswrapper import GSCUDA                                 # utils/gs_cuda/gswrapper.py
    import gsasr_amd.gscuda                                                        # pybind module `gscuda`
    from gsasr_amd.shard import splat_band                                         # multi-GPU row-band shard

The compute lives in gsasr_amd/csrc/splat_{plan,forward,backward,backward_home,step,sampled,shard,api}.hip (hand-written HIP
for gfx950; csrc/gsasr_splat.hip is the same code as one translation unit for the micro-benchmark) behind the C ABI of
include/gsasr_splat.h; Python only moves pointers.  Build with `python -m gsasr_amd.build`.
"""
__version__ = "0.1.0"
