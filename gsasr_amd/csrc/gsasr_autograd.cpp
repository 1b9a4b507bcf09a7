// gsasr_autograd.cpp -- the drop-in `GSCUDA` autograd node as a C++ torch::autograd::Function.
//
// Why: the reference's calling convention is one autograd node per rendered image (utils/gs_cuda_dmax/gswrapper.py:22-44,
// sixteen per training step in basicsr/models/gsasr_model.py:191-233).  A PYTHON torch.autograd.Function makes the
// engine's worker thread take the GIL to run `backward`; a C++ node is called from the engine directly: measured 54 vs 66 us
// for a null node (profiles/history/r03_bwd_experiments.txt (4)).  What it does is exactly what gsasr_amd/gs_cuda*/gswrapper.py
// does through ctypes: plan + splat in forward (reference contract: `rendered_img` is accumulated into and returned),
// one backward into three fresh gradient tensors, `None` for `rendered_img` and `dmax`.
//
// No HIP headers, no device code: the kernels live in libgsasr_splat.so behind the C ABI of include/gsasr_splat.h, which this
// file calls through function pointers handed over by Python (`bind`).  The stream is the one Python read when it called
// `forward` -- the engine runs a node's backward on the stream its forward ran on.
#include <torch/extension.h>

#include <cstdint>
#include <mutex>
#include <vector>

#include "gsasr_splat.h"

namespace {

using plan_fn = int (*)(const float *, const float *, const float *, const gsasr_dims *, void *, size_t, void *);
using fwd_fn = int (*)(const gsasr_dims *, const void *, size_t, float *, void *);
using bwd_fn = int (*)(const float *, const float *, const float *, const float *, float *, float *, float *, const gsasr_dims *,
                       const void *, size_t, void *);
using bytes_fn = size_t (*)(const gsasr_dims *);
using err_fn = const char *(*)(void);
using step_fwd_fn = int (*)(const float *, const float *, const gsasr_dims *, void *, size_t, float *, void *);
using step_fwd_sm_fn = int (*)(const float *, const float *, int, float, int *, const gsasr_dims *, void *, size_t, float *, void *);
using step_bwd_fn = int (*)(const float *, const float *, const float *, float *, const gsasr_dims *, void *, size_t, void *);

plan_fn p_plan = nullptr;
fwd_fn p_fwd = nullptr;
bwd_fn p_bwd = nullptr;
bytes_fn p_bytes = nullptr;
err_fn p_err = nullptr;
bytes_fn p_step_bytes = nullptr;
step_fwd_fn p_step_fwd = nullptr;
step_fwd_sm_fn p_step_fwd_sm = nullptr;
step_bwd_fn p_step_bwd = nullptr;

// Plan workspaces kept between calls, per (device, stream, size, shape): a workspace that comes back from a finished node
// has the OTHER parity's cell counters zeroed by its last plan (GSASR_FLAG_COUNTERS_CLEAN / _PARITY), so the next plan on it
// launches no memset -- the same bookkeeping as gsasr_amd._cabi._WorkspacePool.
struct Pooled {
    int dev;
    int64_t stream;
    int64_t bytes;
    int s, h, w;
    at::Tensor ws;
    unsigned parity;
    unsigned layout = 0;   // layout flags + batch geometry of the plan (step nodes): a workspace is reused only by the same layout
    int batch = 0, slot = 0;
};
std::mutex g_mu;
std::vector<Pooled> g_pool;
constexpr size_t POOL_MAX = 64;

void check(int rc, const char *what)
{
    TORCH_CHECK(rc == 0, what, " failed (status ", rc, "): ", p_err ? p_err() : "?");
}

const float *fptr(const at::Tensor &t, const char *name, int64_t last)
{
    // same failure mode as the reference's CHECK_INPUT (gswrapper.cpp:5-7)
    TORCH_CHECK(t.is_cuda(), name, " must be a CUDA tensor");
    TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32; the kernels read raw fp32");
    TORCH_CHECK(t.dim() >= 1 && t.size(-1) == last, name, " has the wrong last dimension");
    return t.data_ptr<float>();
}

// What a node keeps of its plan.  The workspace goes back to the pool when the NODE dies (not when its backward has run: a
// graph kept with retain_graph=True may run it again), clean for the flipped parity.
struct PlanState : torch::CustomClassHolder {
    at::Tensor ws;
    int64_t stream = 0;
    int s = 0, h = 0, w = 0;
    float dmax = 0.f;
    int parity = -1;   // -1: not pooled (planned under graph capture)
    unsigned flags = 0, layout = 0;      // (step nodes) the dims' flags; their layout part
    int batch = 0, slot = 0, grad_rows = 0;
    std::vector<int> sample_hw;          // (batched canvas) the host array gsasr_dims points to
    ~PlanState() override
    {
        if (parity < 0 || !ws.defined()) return;
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_pool.size() < POOL_MAX)
            g_pool.push_back(Pooled{(int)ws.get_device(), stream, (int64_t)ws.numel(), s, h, w, ws, (unsigned)parity ^ 1u, layout, batch, slot});
    }
    gsasr_dims dims() const
    {
        gsasr_dims d{};
        d.s = s; d.h = h; d.w = w; d.c = 3; d.dmax = dmax; d.row0 = 0; d.row1 = h; d.cutoff = 0.f; d.flags = flags;
        d.batch = batch; d.slot = slot; d.sample_hw = batch > 1 ? sample_hw.data() : nullptr; d.grad_rows = grad_rows;
        return d;
    }
};
static auto plan_state_registration = torch::class_<PlanState>("gsasr_amd", "PlanState").def(torch::init<>());

class GSCudaNode : public torch::autograd::Function<GSCudaNode> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &sigmas, const at::Tensor &coords,
                              const at::Tensor &colors, at::Tensor rendered_img, double dmax, int64_t stream, bool capturing)
    {
        TORCH_CHECK(p_plan, "gsasr_autograd: bind() has not been called");
        const float *ps = fptr(sigmas, "sigmas", 3), *pc = fptr(coords, "coords", 2), *pk = fptr(colors, "colors", 3);
        TORCH_CHECK(rendered_img.dim() == 3 && rendered_img.size(2) == 3, "rendered_img must be [H,W,3]");
        float *pi = const_cast<float *>(fptr(rendered_img, "rendered_img", 3));
        const int64_t s = sigmas.size(0);
        TORCH_CHECK(coords.size(0) == s && colors.size(0) == s, "sigmas, coords, colors disagree on the number of Gaussians");
        TORCH_CHECK(rendered_img.get_device() == sigmas.get_device(), "rendered_img does not match the plan (shape / device)");
        gsasr_dims d{};
        d.s = (int)s; d.h = (int)rendered_img.size(0); d.w = (int)rendered_img.size(1); d.c = 3;
        d.dmax = dmax < 0.0 ? -1.f : (float)dmax;
        d.row0 = 0; d.row1 = d.h; d.cutoff = 0.f; d.flags = 0u;
        const size_t bytes = p_bytes(&d);
        TORCH_CHECK(bytes != 0, "gsasr_splat_workspace_bytes failed: ", p_err());
        at::Tensor ws;
        unsigned parity = 0u;
        bool clean = false;
        if (!capturing) {   // (a captured plan is replayed on the same workspace with the same parity: it zeroes its own counters)
            std::lock_guard<std::mutex> lk(g_mu);
            for (size_t i = g_pool.size(); i-- > 0;) {
                const Pooled &e = g_pool[i];
                if (e.dev == (int)sigmas.get_device() && e.stream == stream && e.bytes == (int64_t)bytes && e.s == d.s && e.h == d.h && e.w == d.w && e.layout == 0u && e.batch == 0) {
                    ws = e.ws;
                    parity = e.parity;
                    clean = true;
                    g_pool.erase(g_pool.begin() + (long)i);
                    break;
                }
            }
        }
        if (!ws.defined()) ws = at::empty({(int64_t)bytes}, sigmas.options().dtype(at::kByte));
        gsasr_dims dp = d;
        if (clean) dp.flags |= GSASR_FLAG_COUNTERS_CLEAN | (parity ? GSASR_FLAG_PARITY : 0u);
        check(p_plan(ps, pc, pk, &dp, ws.data_ptr(), bytes, (void *)stream), "gsasr_splat_plan");
        check(p_fwd(&d, ws.data_ptr(), bytes, pi, (void *)stream), "gsasr_splat_forward");
        ctx->save_for_backward({sigmas, coords, colors});
        auto st = c10::make_intrusive<PlanState>();
        st->ws = ws; st->stream = stream; st->s = d.s; st->h = d.h; st->w = d.w; st->dmax = d.dmax;
        st->parity = capturing ? -1 : (int)parity;
        ctx->saved_data["plan"] = c10::IValue(st);
        ctx->mark_dirty({rendered_img});
        return rendered_img;
    }

    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::tensor_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &sigmas = saved[0], &coords = saved[1], &colors = saved[2];
        const auto st = ctx->saved_data["plan"].toCustomClass<PlanState>();
        // (the Python Functions are @once_differentiable: a differentiable backward -- create_graph=True, gradient penalties --
        // must raise there and here alike instead of silently dropping the second-order terms)
        TORCH_CHECK(!torch::autograd::GradMode::is_enabled() || !grads[0].requires_grad(),
                    "gsasr_amd: the rasterizer's backward is not differentiable (create_graph=True is not supported)");
        const at::Tensor &ws = st->ws;
        const int64_t stream = st->stream;
        at::Tensor g = grads[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        at::Tensor gs = at::empty_like(sigmas), gc = at::empty_like(coords), gk = at::empty_like(colors);
        gsasr_dims d{};
        d.s = st->s; d.h = st->h; d.w = st->w; d.c = 3;
        d.dmax = st->dmax;
        d.row0 = 0; d.row1 = d.h; d.cutoff = 0.f;
        d.flags = GSASR_FLAG_OVERWRITE_GRADS;   // (the reference zero-fills three tensors and adds into them: stored instead)
        check(p_bwd(sigmas.data_ptr<float>(), coords.data_ptr<float>(), colors.data_ptr<float>(), g.data_ptr<float>(),
                    gs.data_ptr<float>(), gc.data_ptr<float>(), gk.data_ptr<float>(), &d, ws.data_ptr(), (size_t)ws.numel(),
                    (void *)stream),
              "gsasr_splat_backward");
        return {gs, gc, gk, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

// The fused step (gsasr_amd.gaussian_splatting._FusedStep / _FusedBatch): raw decoder output [N,9] (or [B,N,9]) -> planar
// image [3,H,W] (or [B,3,slot,W]) and back, one C call each way (prologue + plan + splat; splat backward + chain rule).  What
// `generate_2D_gaussian_splatting_step` -- the function the reference's training loop calls per sample
// (basicsr/models/gsasr_model.py:191-233) -- costs on the host is this node.
constexpr unsigned LAYOUT_FLAGS = GSASR_FLAG_FORWARD_ONLY | GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_GAUSSIAN | GSASR_FLAG_BWD_ATOMIC |
                                  GSASR_FLAG_CHW_GRAD | GSASR_FLAG_STRIDE8;

class StepNode : public torch::autograd::Function<StepNode> {
public:
    static at::Tensor forward(torch::autograd::AutogradContext *ctx, const at::Tensor &gs_parameters, const c10::optional<at::Tensor> &step,
                              int64_t h, int64_t w, double dmax, int64_t flags, const c10::optional<at::Tensor> &scale_modify,
                              int64_t sm_stride, double default_step, int64_t mismatch_ptr, std::vector<int64_t> sizes, int64_t slot,
                              int64_t h_max, int64_t stream, bool capturing)
    {
        TORCH_CHECK(p_step_fwd, "gsasr_autograd: bind_step() has not been called");
        const float *pp = fptr(gs_parameters, "gs_parameters", 9);
        const int batch = (int)(sizes.size() / 2);
        TORCH_CHECK(batch <= 1 ? gs_parameters.dim() == 2 : (gs_parameters.dim() == 3 && gs_parameters.size(0) == batch),
                    "gs_parameters must be [N,9] (one image) or [B,N,9] (batched canvas)");
        auto st = c10::make_intrusive<PlanState>();
        st->s = (int)(gs_parameters.numel() / 9);
        st->h = (int)h; st->w = (int)w; st->dmax = dmax < 0.0 ? -1.f : (float)dmax;
        st->flags = (unsigned)flags | GSASR_FLAG_OVERWRITE_IMAGE | GSASR_FLAG_CHW_IMAGE;
        st->layout = st->flags & LAYOUT_FLAGS;
        st->stream = stream;
        if (batch > 1) {
            st->batch = batch; st->slot = (int)slot;
            st->sample_hw.assign(sizes.begin(), sizes.end());
        }
        gsasr_dims d = st->dims();
        const size_t bytes = p_step_bytes(&d);
        TORCH_CHECK(bytes != 0, "gsasr_step_workspace_bytes failed: ", p_err());
        unsigned parity = 0u;
        bool clean = false;
        if (!capturing) {
            std::lock_guard<std::mutex> lk(g_mu);
            for (size_t i = g_pool.size(); i-- > 0;) {
                const Pooled &e = g_pool[i];
                if (e.dev == (int)gs_parameters.get_device() && e.stream == stream && e.bytes == (int64_t)bytes && e.s == d.s && e.h == d.h &&
                    e.w == d.w && e.layout == (st->layout | 0x80000000u) && e.batch == st->batch && e.slot == st->slot) {
                    st->ws = e.ws;
                    parity = e.parity;
                    clean = true;
                    g_pool.erase(g_pool.begin() + (long)i);
                    break;
                }
            }
        }
        st->layout |= 0x80000000u;     // (a step workspace is never handed to a GSCUDA node and vice versa)
        if (!st->ws.defined()) st->ws = at::empty({(int64_t)bytes}, gs_parameters.options().dtype(at::kByte));
        st->parity = capturing ? -1 : (int)parity;
        gsasr_dims dp = d;
        if (clean) dp.flags |= GSASR_FLAG_COUNTERS_CLEAN | (parity ? GSASR_FLAG_PARITY : 0u);
        at::Tensor img = batch > 1 ? at::empty({batch, 3, slot, w}, gs_parameters.options()) : at::empty({3, h, w}, gs_parameters.options());
        const float *ps = nullptr;
        if (scale_modify.has_value() && scale_modify->defined()) {
            TORCH_CHECK(scale_modify->is_cuda() && scale_modify->scalar_type() == at::kFloat, "scale_modify must be a float32 CUDA tensor");
            TORCH_CHECK(scale_modify->get_device() == gs_parameters.get_device(), "scale_modify lives on another device than gs_parameters");
            check(p_step_fwd_sm(pp, scale_modify->data_ptr<float>(), (int)sm_stride, (float)default_step, (int *)mismatch_ptr, &dp,
                                st->ws.data_ptr(), bytes, img.data_ptr<float>(), (void *)stream),
                  "gsasr_step_forward_sm");
        } else {
            TORCH_CHECK(step.has_value() && step->defined(), "step size missing");
            ps = fptr(*step, "step_size", step->size(-1));
            TORCH_CHECK(step->get_device() == gs_parameters.get_device(), "step_size lives on another device than gs_parameters");
            check(p_step_fwd(pp, ps, &dp, st->ws.data_ptr(), bytes, img.data_ptr<float>(), (void *)stream), "gsasr_step_forward");
        }
        ctx->save_for_backward({gs_parameters, (step.has_value() && step->defined()) ? *step : at::Tensor()});
        ctx->saved_data["plan"] = c10::IValue(st);
        // (batched canvas: the slot is h_max rounded up to whole 16-row tiles; the caller sees [B,3,h_max,W], and the backward
        // reads the gradient of exactly that shape in place)
        return batch > 1 && h_max < slot ? img.slice(2, 0, h_max) : img;
    }

    static torch::autograd::tensor_list backward(torch::autograd::AutogradContext *ctx, torch::autograd::tensor_list grads)
    {
        const auto saved = ctx->get_saved_variables();
        const at::Tensor &gs_parameters = saved[0], &step = saved[1];
        const auto st = ctx->saved_data["plan"].toCustomClass<PlanState>();
        // (the Python Functions are @once_differentiable: a differentiable backward -- create_graph=True, gradient penalties --
        // must raise there and here alike instead of silently dropping the second-order terms)
        TORCH_CHECK(!torch::autograd::GradMode::is_enabled() || !grads[0].requires_grad(),
                    "gsasr_amd: the rasterizer's backward is not differentiable (create_graph=True is not supported)");
        at::Tensor g = grads[0];
        if (g.scalar_type() != at::kFloat) g = g.to(at::kFloat);
        g = g.contiguous();
        at::Tensor gp = at::empty_like(gs_parameters);
        gsasr_dims d = st->dims();
        d.flags |= GSASR_FLAG_CHW_GRAD;      // the planar gradient autograd hands back is read as it is
        if (st->batch > 1) d.grad_rows = (int)g.size(2);     // [B,3,Hmax,W]: rows per plane
        check(p_step_bwd(gs_parameters.data_ptr<float>(), step.defined() ? step.data_ptr<float>() : nullptr, g.data_ptr<float>(),
                         gp.data_ptr<float>(), &d, st->ws.data_ptr(), (size_t)st->ws.numel(), (void *)st->stream),
              "gsasr_step_backward");
        return {gp, at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(),
                at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor(), at::Tensor()};
    }
};

at::Tensor step_apply(const at::Tensor &gs_parameters, const c10::optional<at::Tensor> &step, int64_t h, int64_t w, double dmax, int64_t flags,
                      const c10::optional<at::Tensor> &scale_modify, int64_t sm_stride, double default_step, int64_t mismatch_ptr,
                      std::vector<int64_t> sizes, int64_t slot, int64_t h_max, int64_t stream, bool capturing)
{
    return StepNode::apply(gs_parameters, step, h, w, dmax, flags, scale_modify, sm_stride, default_step, mismatch_ptr, sizes, slot, h_max,
                           stream, capturing);
}

void bind_step(int64_t bytes, int64_t fwd, int64_t fwd_sm, int64_t bwd)
{
    p_step_bytes = (bytes_fn)bytes;
    p_step_fwd = (step_fwd_fn)fwd;
    p_step_fwd_sm = (step_fwd_sm_fn)fwd_sm;
    p_step_bwd = (step_bwd_fn)bwd;
}

at::Tensor gscuda_apply(const at::Tensor &sigmas, const at::Tensor &coords, const at::Tensor &colors, at::Tensor rendered_img,
                        double dmax, int64_t stream, bool capturing)
{
    return GSCudaNode::apply(sigmas, coords, colors, rendered_img, dmax, stream, capturing);
}

void bind(int64_t plan, int64_t fwd, int64_t bwd, int64_t bytes, int64_t err)
{
    p_plan = (plan_fn)plan;
    p_fwd = (fwd_fn)fwd;
    p_bwd = (bwd_fn)bwd;
    p_bytes = (bytes_fn)bytes;
    p_err = (err_fn)err;
}

void clear_pool()
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_pool.clear();
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("bind", &bind, "hand over the C-ABI entry points of libgsasr_splat.so (addresses from ctypes)");
    m.def("gscuda_apply", &gscuda_apply, "GSCUDA.apply(sigmas, coords, colors, rendered_img, dmax) as a C++ autograd node; dmax < 0: gs_cuda");
    m.def("bind_step", &bind_step, "hand over gsasr_step_workspace_bytes / _forward / _forward_sm / _backward");
    m.def("step_apply", &step_apply, "the fused step (prologue + plan + splat, and back) as a C++ autograd node");
    m.def("clear_pool", &clear_pool, "drop the pooled plan workspaces");
}
