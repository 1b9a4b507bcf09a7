// splat_api.hip -- the library's state (error message, default cutoff, plan notes), workspace checks, the small entry points and the reference-shaped launchers
// (one translation unit of libgsasr_splat.so; gsasr_splat.hip has the overview of the whole pipeline)
#include "splat_common.h"

using namespace gsasr_detail;

namespace gsasr_detail {

// process default of the support cutoff: 0 = adaptive; first read from the environment (GSASR_SPLAT_CUTOFF), then
// whatever gsasr_set_default_cutoff stored.  One atomic word: setting and planning from different threads is a benign
// race on WHICH value a plan sees, never a torn one.
static std::atomic<float> g_default_cutoff{-12345.f};

float default_cutoff()
{
    float v = g_default_cutoff.load(std::memory_order_relaxed);
    if (v == -12345.f) {
        const char *e = getenv("GSASR_SPLAT_CUTOFF");
        float init = e ? (float)atof(e) : 0.f, expected = -12345.f;
        g_default_cutoff.compare_exchange_strong(expected, init, std::memory_order_relaxed);
        v = g_default_cutoff.load(std::memory_order_relaxed);
    }
    return v;
}

void store_default_cutoff(float tau) { g_default_cutoff.store(tau, std::memory_order_relaxed); }

// Kernel choices registered per shape (gsasr_set_kernel_choice).  A handful of entries, looked up by the policy functions of
// splat_common.h on every call: one relaxed load says "nothing registered" (the normal case), else a linear search under a mutex.
struct ChoiceEntry { int s, h, w, row0, row1, batch, slot; float dmax, cutoff; bool fwd_only; KernelChoice c; };
static std::mutex g_choice_mu;
static std::vector<ChoiceEntry> g_choices;
static std::atomic<int> g_nchoices{0};

static ChoiceEntry choice_key(const gsasr_dims *d)
{
    ChoiceEntry e{};
    e.s = d->s; e.h = d->h; e.w = d->w; e.row0 = d->row0; e.row1 = d->row1;
    e.batch = batch_of(d); e.slot = d->batch > 1 ? d->slot : 0;
    e.dmax = d->dmax >= 0.f ? d->dmax : -1.f; e.cutoff = d->cutoff;
    e.fwd_only = (d->flags & GSASR_FLAG_FORWARD_ONLY) != 0;
    return e;
}

static bool same_shape(const ChoiceEntry &a, const ChoiceEntry &b)
{
    return a.s == b.s && a.h == b.h && a.w == b.w && a.row0 == b.row0 && a.row1 == b.row1 && a.batch == b.batch && a.slot == b.slot &&
           a.dmax == b.dmax && a.cutoff == b.cutoff && a.fwd_only == b.fwd_only;
}

KernelChoice registered_choice(const gsasr_dims *d)
{
    if (g_nchoices.load(std::memory_order_relaxed) == 0) return KernelChoice{0u, 0};
    const ChoiceEntry k = choice_key(d);
    std::lock_guard<std::mutex> lk(g_choice_mu);
    for (const ChoiceEntry &e : g_choices)
        if (same_shape(e, k)) return e.c;
    return KernelChoice{0u, 0};
}

// Which plans carry slots.  The slot count of a workspace follows from the flags of the dims the PLAN was made with; a
// backward (or the gather of a step call) that derives it from its OWN flags would, when the two disagree, read slots and
// spans the plan never wrote.  The plan therefore leaves a note {workspace -> slots per Gaussian} here and every later
// call on that workspace lays it out from the note (a small direct-mapped table: a lost note only means the old
// behaviour, trusting the caller's flags).
// One 64-bit word per note, written and read atomically (no lock on the plan / forward / backward path): the workspace
// address (256-byte aligned: bits 8..47), a 16-bit hash of the shape the plan was made for, and the slot count.  A note
// whose shape hash differs from the caller's dims -- a freed workspace address reused for another shape without a new plan,
// a note overwritten by a colliding workspace -- is ignored.
constexpr int NOTES = 1024;
static std::atomic<unsigned long long> g_notes[NOTES];
// ... and a second word per note: the capacity of the plan's tile lists (the stride of tl_entries), tagged with the same shape
// hash.  The capacity follows from gsasr_dims.list_cap OR from the kernel choice registered for the shape at PLAN time; a
// forward / backward that re-derived it from the registry would, after a tune() / reset between the plan and the render, read
// the lists with another stride than k_bin wrote them with.  The plan's value is frozen here instead.
static std::atomic<unsigned long long> g_note_caps[NOTES];

static unsigned note_slot(const void *ws) { return (unsigned)(((uintptr_t)ws >> 8) * 2654435761u >> 22) & (NOTES - 1); }

static unsigned long long note_shape(const gsasr_dims *d)
{
    unsigned long long x = (unsigned long long)(unsigned)d->s * 0x9E3779B97F4A7C15ull;
    x ^= ((unsigned long long)(unsigned)d->h << 32 | (unsigned)d->w) * 0xC2B2AE3D27D4EB4Full;
    x ^= (unsigned long long)(unsigned)batch_of(d) * 0x27D4EB2F165667C5ull;
    return (x >> 40) & 0xffffull;
}

// (payload byte: slots per Gaussian in bits 0..4, the tile lists' tile height in bits 5..6: 0 none, 1 = 16 rows, 2 = 32)
void note_plan(const void *ws, const gsasr_dims *d, int part_k, int tl_hlog, int tl_cap)
{
    // (the capacity first: a reader that sees the new first word also sees a capacity of the same plan or a newer one of the
    // same workspace and shape -- concurrent plans on ONE workspace are the caller's race either way)
    g_note_caps[note_slot(ws)].store((note_shape(d) << 32) | (unsigned long long)(unsigned)tl_cap, std::memory_order_release);
    const unsigned long long w = ((unsigned long long)(uintptr_t)ws & 0x0000ffffffffff00ull) << 16 | note_shape(d) << 8 |
                                 (unsigned long long)((part_k & 0x1f) | (tl_hlog ? (tl_hlog - 3) << 5 : 0));
    g_notes[note_slot(ws)].store(w, std::memory_order_release);
}

// layout of the plan in `ws`: from the note its plan left, else from these dims -- except the tile lists, which a call uses
// only on the note's word (a lost note means the search, never a list nobody wrote)
Layout plan_layout(const gsasr_dims *d, const void *ws)
{
    int part_k = -1, tl_hlog = -1, tl_cap = -1;
    bool cap_lost = false;
    if (ws) {
        const unsigned long long w = g_notes[note_slot(ws)].load(std::memory_order_acquire);
        const unsigned long long key = ((unsigned long long)(uintptr_t)ws & 0x0000ffffffffff00ull) << 16 | note_shape(d) << 8;
        if ((w & ~0xffull) == key) {
            part_k = (int)(w & 0x1full);
            tl_hlog = (int)((w >> 5) & 3ull) ? (int)((w >> 5) & 3ull) + 3 : 0;
            const unsigned long long c = g_note_caps[note_slot(ws)].load(std::memory_order_acquire);
            if ((c >> 32) == note_shape(d)) tl_cap = (int)(unsigned)(c & 0xffffffffull);
            else cap_lost = true;  // (the capacity word belongs to another plan: the lists are not read -- never with a guessed stride)
        }
    }
    // (without a note the list region is still SIZED from the dims -- the step entry points place their scratch behind the
    // plan -- but nothing reads it)
    Layout L = make_layout(d, part_k, tl_hlog, tl_cap);
    if (tl_hlog < 0 || cap_lost) L.tl_ok = false;
    return L;
}

static thread_local char tl_err[256] = "";

int fail(int code, const char *msg)
{
    snprintf(tl_err, sizeof tl_err, "%s", msg);
    return code;
}

int hip_fail(hipError_t e, const char *where)
{
    snprintf(tl_err, sizeof tl_err, "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}


const char *last_error_message() { return tl_err; }

int check_ws(const gsasr_dims *dims, const void *ws, size_t ws_bytes, Layout &L, bool planning)
{
    if (!dims_ok(dims)) return fail(GSASR_ERR_ARG, "bad dims (need c==3, 2<=h,w<=32767, 0<=row0<=row1<=h)");
    if (dims->batch > 1 && dims->grad_rows != 0 && (dims->flags & GSASR_FLAG_CHW_GRAD)) {
        // planar gradient of a batched canvas [B, 3, grad_rows, w]: every sample's rows must lie inside its planes
        int hmax = 0;
        for (int b = 0; b < dims->batch; ++b) hmax = dims->sample_hw[2 * b] > hmax ? dims->sample_hw[2 * b] : hmax;
        if (dims->grad_rows < hmax) return fail(GSASR_ERR_ARG, "grad_rows is smaller than a sample's height");
    } else if (dims->batch <= 1 && dims->grad_rows != 0 && dims->grad_rows != dims->row1 - dims->row0) {
        return fail(GSASR_ERR_ARG, "grad_rows applies to a batched canvas only (leave it 0)");
    }
    L = planning ? make_layout(dims) : plan_layout(dims, ws);
    if (!ws || ((uintptr_t)ws & 255u)) return fail(GSASR_ERR_WORKSPACE, "workspace null or not 256-byte aligned");
    if (ws_bytes < L.total) return fail(GSASR_ERR_WORKSPACE, "workspace smaller than gsasr_splat_workspace_bytes()");
    return GSASR_OK;
}

}  // namespace gsasr_detail

extern "C" {


int gsasr_abi_version(void) { return GSASR_SPLAT_ABI_VERSION; }

const char *gsasr_last_error(void) { return last_error_message(); }

void gsasr_set_default_cutoff(float tau) { store_default_cutoff(tau); }

float gsasr_get_default_cutoff(void) { return default_cutoff(); }

int gsasr_set_kernel_choice(const gsasr_dims *shape, unsigned flags, int list_cap)
{
    constexpr unsigned ALLOWED = GSASR_FLAG_FWD_WIDE | GSASR_FLAG_FWD_NARROW | GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_GAUSSIAN | GSASR_FLAG_BWD_HOME;
    if (!shape) return fail(GSASR_ERR_ARG, "null shape");
    if (flags & ~ALLOWED) return fail(GSASR_ERR_ARG, "a kernel choice holds GSASR_FLAG_FWD_WIDE | _FWD_NARROW | _BWD_TILE | _BWD_GAUSSIAN | _BWD_HOME only");
    if ((flags & GSASR_FLAG_FWD_WIDE) && (flags & GSASR_FLAG_FWD_NARROW)) return fail(GSASR_ERR_ARG, "GSASR_FLAG_FWD_WIDE and _FWD_NARROW exclude each other");
    if (__builtin_popcount(flags & (GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_GAUSSIAN | GSASR_FLAG_BWD_HOME)) > 1)
        return fail(GSASR_ERR_ARG, "GSASR_FLAG_BWD_TILE, _BWD_GAUSSIAN and _BWD_HOME exclude each other");
    ChoiceEntry k = choice_key(shape);
    k.c = KernelChoice{flags, list_cap};
    std::lock_guard<std::mutex> lk(g_choice_mu);
    for (ChoiceEntry &e : g_choices)
        if (same_shape(e, k)) { e.c = k.c; return GSASR_OK; }
    if (flags == 0u && list_cap == 0) return GSASR_OK;      // "no choice" for a shape without an entry: nothing to store
    if (g_choices.size() >= 256) {      // full: the oldest registration goes (a training run over ragged crop sizes must not
        static unsigned oldest = 0;     // start failing at its 257th shape; that shape falls back to the library's rule)
        g_choices[oldest++ & 255u] = k;
        return GSASR_OK;
    }
    g_choices.push_back(k);
    g_nchoices.store((int)g_choices.size(), std::memory_order_relaxed);
    return GSASR_OK;
}

int gsasr_get_kernel_choice(const gsasr_dims *shape, unsigned *flags, int *list_cap)
{
    if (!shape || g_nchoices.load(std::memory_order_relaxed) == 0) return 0;
    const ChoiceEntry k = choice_key(shape);
    std::lock_guard<std::mutex> lk(g_choice_mu);
    for (const ChoiceEntry &e : g_choices)
        if (same_shape(e, k)) {
            if (flags) *flags = e.c.flags;
            if (list_cap) *list_cap = e.c.list_cap;
            return 1;
        }
    return 0;
}

void gsasr_clear_kernel_choices(void)
{
    std::lock_guard<std::mutex> lk(g_choice_mu);
    g_choices.clear();
    g_nchoices.store(0, std::memory_order_relaxed);
}

float gsasr_resolve_cutoff(float cutoff, int s) { return resolve_cutoff(cutoff, s); }

int gsasr_plan_cutoff(const gsasr_dims *dims, const void *workspace, size_t workspace_bytes, void *stream, float *tau,
                      unsigned *k_box)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L)) return rc;
    const PlanView V = make_view(L, const_cast<void *>(workspace));
    unsigned h[HDR_WORDS];
    HIP_TRY(hipMemcpyAsync(h, V.hdr, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    float t;
    memcpy(&t, &h[4], 4);
    if (tau) *tau = t;
    if (k_box) *k_box = h[5];
    return GSASR_OK;
}

size_t gsasr_splat_workspace_bytes(const gsasr_dims *dims)
{
    if (!dims_ok(dims)) {
        fail(GSASR_ERR_ARG, "bad dims");
        return 0;
    }
    return make_layout(dims).total;
}


// ---- reference-shaped launchers -------------------------------------------------------------------
// Scratch of the reference-shaped launchers below.  The reference's gs.h launchers take no workspace, so these keep one
// per (device, stream) between calls -- stream-ordered reuse, exactly like the caching allocator behind the reference's
// own torch.zeros -- instead of a hipMallocAsync / hipFreeAsync pair per call; a workspace that is planned again for the
// same shape also skips the memset of its cell counters (GSASR_FLAG_COUNTERS_CLEAN / _PARITY: every plan zeroes the other
// parity's counters on the side).  gsasr_release_launcher_scratch() frees them.
struct LauncherScratch {
    int dev;
    hipStream_t st;
    void *ptr;
    size_t bytes;
    int s, h, w;          // shape of the last plan made in it (the counters' layout)
    unsigned plans;       // plans made for that shape so far
    unsigned long long used;
};
constexpr int LAUNCHER_SLOTS = 8;
static LauncherScratch g_scratch[LAUNCHER_SLOTS];
static std::mutex g_scratch_mu;
static unsigned long long g_scratch_clock = 0;

// (the caller holds g_scratch_mu from here until its kernels are enqueued: a second host thread can then neither evict the entry
// nor take the next parity before the first thread's plan sits in the stream)
static int launcher_scratch(const gsasr_dims &d, size_t bytes, hipStream_t st, void **ws, unsigned *flags)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    LauncherScratch *e = nullptr;
    for (LauncherScratch &c : g_scratch)
        if (c.ptr && c.dev == dev && c.st == st) { e = &c; break; }
    if (!e) {   // a free slot, else the least recently used one
        for (LauncherScratch &c : g_scratch)
            if (!c.ptr) { e = &c; break; }
        if (!e) {
            e = &g_scratch[0];
            for (LauncherScratch &c : g_scratch)
                if (c.used < e->used) e = &c;
        }
        if (e->ptr) {   // evict: freed in the order of ITS stream
            int cur = dev;
            if (e->dev != cur) HIP_TRY(hipSetDevice(e->dev));
            hipError_t fe = hipFreeAsync(e->ptr, e->st);
            if (e->dev != cur) HIP_TRY(hipSetDevice(cur));
            if (fe != hipSuccess) return hip_fail(fe, "hipFreeAsync");
        }
        *e = LauncherScratch{dev, st, nullptr, 0, 0, 0, 0, 0u, 0ull};
    }
    if (e->bytes < bytes) {
        if (e->ptr) HIP_TRY(hipFreeAsync(e->ptr, st));
        e->ptr = nullptr;
        e->bytes = 0;
        void *p = nullptr;
        HIP_TRY(hipMallocAsync(&p, bytes, st));
        e->ptr = p;
        e->bytes = bytes;
        e->plans = 0;
    }
    if (e->s != d.s || e->h != d.h || e->w != d.w) {
        e->s = d.s; e->h = d.h; e->w = d.w;
        e->plans = 0;
    }
    *flags = e->plans == 0 ? 0u : (GSASR_FLAG_COUNTERS_CLEAN | ((e->plans & 1u) ? GSASR_FLAG_PARITY : 0u));
    ++e->plans;
    e->used = ++g_scratch_clock;
    *ws = e->ptr;
    return GSASR_OK;
}

int gsasr_release_launcher_scratch(void)
{
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    int rc = GSASR_OK;
    for (LauncherScratch &c : g_scratch) {
        if (!c.ptr) continue;
        if (c.dev != cur) (void)hipSetDevice(c.dev);
        hipError_t e = hipFreeAsync(c.ptr, c.st);
        if (c.dev != cur) (void)hipSetDevice(cur);
        if (e != hipSuccess) rc = hip_fail(e, "hipFreeAsync");
        c = LauncherScratch{};
    }
    return rc;
}

static int render_common(const float *sigmas, const float *coords, const float *colors, float *img,
                         const float *grads, float *gs, float *gc, float *gk, int s, int h, int w, int c,
                         float dmax, bool backward, void *stream)
{
    gsasr_dims d{};   // (batch fields zero: one image)
    d.s = s; d.h = h; d.w = w; d.c = c; d.dmax = dmax; d.row0 = 0; d.row1 = h; d.cutoff = 0.f; d.flags = 0;
    const size_t bytes = gsasr_splat_workspace_bytes(&d);
    if (!bytes) return GSASR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    void *ws = nullptr;
    unsigned counter_flags = 0u;
    // Under stream capture the cached scratch must not be touched: a pointer that came from a captured hipMallocAsync is only
    // valid inside the graph, and a captured plan is replayed with ONE parity, so it must zero its own counters (flags = 0).
    // The captured call therefore allocates, plans and frees stream-ordered, all three as nodes of the graph.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing(st, &cap));
    const bool capturing = cap != hipStreamCaptureStatusNone;
    std::unique_lock<std::mutex> lk(g_scratch_mu, std::defer_lock);
    if (capturing) {
        HIP_TRY(hipMallocAsync(&ws, bytes, st));
    } else {
        lk.lock();      // held until the plan and the render are enqueued (launcher_scratch)
        if (int rc = launcher_scratch(d, bytes, st, &ws, &counter_flags)) return rc;
    }
    d.flags = counter_flags;
    int rc = gsasr_splat_plan(sigmas, coords, colors, &d, ws, bytes, stream);
    d.flags = 0;
    if (rc == GSASR_OK) {
        if (!backward) {
            rc = gsasr_splat_forward(&d, ws, bytes, img, stream);
        } else {
            if (dmax < 0.f) d.flags |= GSASR_FLAG_OVERWRITE_GRADS;  // gs_cuda backward overwrites (gs.cu:169-176)
            rc = gsasr_splat_backward(sigmas, coords, colors, grads, gs, gc, gk, &d, ws, bytes, stream);
        }
    }
    if (capturing) {
        const hipError_t fe = hipFreeAsync(ws, st);
        if (rc == GSASR_OK && fe != hipSuccess) rc = hip_fail(fe, "hipFreeAsync");
    }
    return rc;
}

int gsasr_gs_render(const float *sigmas, const float *coords, const float *colors, float *rendered_img, int s,
                    int h, int w, int c, void *stream)
{
    return render_common(sigmas, coords, colors, rendered_img, nullptr, nullptr, nullptr, nullptr, s, h, w, c,
                         -1.f, false, stream);
}

int gsasr_gs_render_backward(const float *sigmas, const float *coords, const float *colors, const float *grads,
                             float *grads_sigmas, float *grads_coords, float *grads_colors, int s, int h, int w,
                             int c, void *stream)
{
    return render_common(sigmas, coords, colors, nullptr, grads, grads_sigmas, grads_coords, grads_colors, s, h,
                         w, c, -1.f, true, stream);
}

int gsasr_gs_render_dmax(const float *sigmas, const float *coords, const float *colors, float *rendered_img,
                         int s, int h, int w, int c, float dmax, void *stream)
{
    if (!(dmax >= 0.f)) return fail(GSASR_ERR_ARG, "dmax must be >= 0");
    return render_common(sigmas, coords, colors, rendered_img, nullptr, nullptr, nullptr, nullptr, s, h, w, c,
                         dmax, false, stream);
}

int gsasr_gs_render_backward_dmax(const float *sigmas, const float *coords, const float *colors,
                                  const float *grads, float *grads_sigmas, float *grads_coords,
                                  float *grads_colors, int s, int h, int w, int c, float dmax, void *stream)
{
    if (!(dmax >= 0.f)) return fail(GSASR_ERR_ARG, "dmax must be >= 0");
    return render_common(sigmas, coords, colors, nullptr, grads, grads_sigmas, grads_coords, grads_colors, s, h,
                         w, c, dmax, true, stream);
}

}  // extern "C"
