// gsasr_splat.hip -- MI355X (gfx950 / CDNA4) 2D Gaussian-splatting rasterizer behind the C ABI of
// include/gsasr_splat.h.  Written for wave64 / 256 CUs / 8 XCDs; not a translation of the
// reference's CUDA (utils/gs_cuda*/gs.cu), whose semantics it reproduces:
//
//   forward : img[p,:] += sum_s [|dx|<=dmax & |dy|<=dmax] exp(w1_s * q_s(p)) * colors[s,:]
//   backward: analytic gradient of sum(grad_img * img) w.r.t. sigmas[s,3], coords[s,2], colors[s,3]
//
// Pipeline (all on the caller's stream, no host sync):
//   plan     k_classify  per Gaussian: pixel bounding box of (dmax box  ∩  sigma*sqrt(2 tau) support box),
//                        class {normal -> 16x16-px cell of its centre | large | dead}, cell histogram,
//                        max extent of the normal class; also the px/py pixel-coordinate tables
//                        (double expression rounded to float, as gs.cu:27-28 does per pixel).
//            k_scan      exclusive scan of the cell histogram (one workgroup).
//            k_scatter   counting-sort scatter of Gaussian indices into cell order.
//            k_pack      cell-ordered 32-byte records {x, y, A, B, C, r, g, b} (A,B,C = exponent
//                        coefficients with log2(e) folded, computed in double) + 8-byte pixel bboxes.
//   forward  k_render_fwd  PIXEL-stationary: one wave64 = one 8x8 pixel sub-tile, RGB accumulators in
//                        registers.  The wave walks the cell rows within the class' max extent; 64
//                        candidates are box-tested at once (one per lane, 8-byte bbox), the hit mask is
//                        a ballot in SGPRs, and each hit's record is fetched with SCALAR loads (wave-
//                        uniform data belongs in SGPRs on CDNA) -- no LDS, no atomics, one coalesced
//                        read-modify-write of the tile at the end.
//   backward k_render_bwd  GAUSSIAN-stationary: one wave64 = one Gaussian, lanes sweep the pixels of
//                        its box reading grad_img (L1/L2 resident), five moment sums + three colour sums
//                        in registers, ONE DPP wave reduction per Gaussian, plain store of the 8 grads
//                        (no atomics, deterministic).  Gaussians of the "large" class are split into
//                        row chunks spread over all waves and combined with fp32 atomics.
//
// No MFMA: this is gather/scatter-accumulate with one transcendental per pair, not a contraction.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gsasr_splat.h"

namespace {

constexpr int CELL = 16;        // binning cell side in pixels
constexpr int CELL_SHIFT = 4;
constexpr int SUB = 8;          // forward sub-tile side: 8x8 pixels = one wave64
constexpr int RCAP_PX = 128;    // half-extent (px) above which a Gaussian is binned as "large"
constexpr int NCH = 64;         // row chunks a large Gaussian is split into in backward
constexpr int HDR_WORDS = 64;   // plan header (uint32): [0]=max x half-extent of normals, [1]=max y
constexpr double LOG2E = 1.4426950408889634074;

struct Params {
    int s, h, w, row0, row1;
    int bounded;     // 1: gs_cuda_dmax box test, 0: gs_cuda (no test)
    float dmax;      // box half-size (normalised units); +inf when !bounded
    float kcut;      // sqrt(2 tau) or 0 when the support cutoff is disabled
    int ncx, ncy, ncells;
};

struct PlanView {
    unsigned *hdr;          // [HDR_WORDS]
    unsigned *cell_count;   // [ncells+2]   (ncells = "large" class, ncells+1 = "dead" class)
    unsigned *cell_cursor;  // [ncells+2]
    unsigned *cell_start;   // [ncells+3]   exclusive scan of cell_count, last = s
    float *px, *py;         // [w], [h]
    unsigned *key;          // [s] class/cell of Gaussian i
    unsigned *perm;         // [s] cell-ordered position -> Gaussian index
    float4 *rec;            // [2*s]
    short4 *bbox;           // [s]
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct Layout {
    size_t off_hdr, off_count, off_cursor, off_start, off_px, off_py, off_key, off_perm, off_rec, off_bbox;
    size_t zero_bytes;  // header + count + cursor are zeroed by one memset at the start of plan
    size_t total;
    int ncx, ncy, ncells;
};

bool dims_ok(const gsasr_dims *d)
{
    return d && d->s >= 0 && d->h >= 2 && d->w >= 2 && d->h <= 32767 && d->w <= 32767 && d->c == 3 &&
           d->row0 >= 0 && d->row0 <= d->row1 && d->row1 <= d->h && !(d->dmax != d->dmax);
}

Layout make_layout(const gsasr_dims *d)
{
    Layout L{};
    L.ncx = (d->w + CELL - 1) / CELL;
    L.ncy = (d->h + CELL - 1) / CELL;
    L.ncells = L.ncx * L.ncy;
    const size_t ncls = (size_t)L.ncells + 2, s = (size_t)d->s;
    size_t o = 0;
    L.off_hdr = o;    o += HDR_WORDS * 4;
    L.off_count = o;  o += align_up(ncls * 4, 256);
    L.off_cursor = o; o += align_up(ncls * 4, 256);
    L.zero_bytes = o;
    L.off_start = o;  o += align_up((ncls + 1) * 4, 256);
    L.off_px = o;     o += align_up((size_t)d->w * 4, 256);
    L.off_py = o;     o += align_up((size_t)d->h * 4, 256);
    L.off_key = o;    o += align_up(s * 4, 256);
    L.off_perm = o;   o += align_up(s * 4, 256);
    L.off_rec = o;    o += align_up(s * 32, 256);
    L.off_bbox = o;   o += align_up(s * 8, 256);
    L.total = o;
    return L;
}

PlanView make_view(const Layout &L, void *ws)
{
    char *b = (char *)ws;
    PlanView V;
    V.hdr = (unsigned *)(b + L.off_hdr);
    V.cell_count = (unsigned *)(b + L.off_count);
    V.cell_cursor = (unsigned *)(b + L.off_cursor);
    V.cell_start = (unsigned *)(b + L.off_start);
    V.px = (float *)(b + L.off_px);
    V.py = (float *)(b + L.off_py);
    V.key = (unsigned *)(b + L.off_key);
    V.perm = (unsigned *)(b + L.off_perm);
    V.rec = (float4 *)(b + L.off_rec);
    V.bbox = (short4 *)(b + L.off_bbox);
    return V;
}

float g_default_cutoff = -12345.f;  // resolved lazily (env GSASR_SPLAT_CUTOFF or the header default)

float default_cutoff()
{
    if (g_default_cutoff == -12345.f) {
        const char *e = getenv("GSASR_SPLAT_CUTOFF");
        g_default_cutoff = e ? (float)atof(e) : GSASR_SPLAT_DEFAULT_CUTOFF;
        if (g_default_cutoff == 0.f) g_default_cutoff = GSASR_SPLAT_DEFAULT_CUTOFF;
    }
    return g_default_cutoff;
}

Params make_params(const gsasr_dims *d, const Layout &L)
{
    Params P;
    P.s = d->s; P.h = d->h; P.w = d->w; P.row0 = d->row0; P.row1 = d->row1;
    P.bounded = d->dmax >= 0.f;
    P.dmax = P.bounded ? d->dmax : INFINITY;
    float tau = d->cutoff == 0.f ? default_cutoff() : d->cutoff;
    P.kcut = tau > 0.f ? (float)(std::sqrt(2.0 * (double)tau) * (1.0 + 1e-6)) : 0.f;
    P.ncx = L.ncx; P.ncy = L.ncy; P.ncells = L.ncells;
    return P;
}

thread_local char tl_err[256] = "";

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

// ---------------------------------------------------------------------------------------------------
// geometry shared by classify / pack / backward
// ---------------------------------------------------------------------------------------------------
struct Box {
    int c0, c1, r0, r1;  // inclusive pixel-index window, clipped to the image and the owned rows
    float ex, ey;        // half-extents in pixels (before clipping)
    int cls;             // 0 normal, 1 large, 2 dead
};

__device__ __forceinline__ Box gaussian_box(float sx, float sy, float x, float y, const Params &P)
{
    Box b;
    float ext_x = P.dmax, ext_y = P.dmax;
    if (P.kcut > 0.f) {  // marginal bound of the ellipse {exponent >= -tau}: |dx| <= sx*sqrt(2 tau), any rho
        ext_x = fminf(ext_x, P.kcut * sx);
        ext_y = fminf(ext_y, P.kcut * sy);
    }
    const float hx = 0.5f * (float)(P.w - 1), hy = 0.5f * (float)(P.h - 1);
    const float cxp = (x + 1.f) * hx, cyp = (y + 1.f) * hy;
    b.ex = ext_x * hx;
    b.ey = ext_y * hy;
    // +-1 px of slack covers every rounding between this window and the kernels' own float tests
    const float lox = floorf(cxp - b.ex) - 1.f, hix = ceilf(cxp + b.ex) + 1.f;
    const float loy = floorf(cyp - b.ey) - 1.f, hiy = ceilf(cyp + b.ey) + 1.f;
    const bool finite = (sx - sx == 0.f) && (sy - sy == 0.f) && (x - x == 0.f) && (y - y == 0.f);
    b.c0 = (int)fmaxf(lox, 0.f);
    b.c1 = (int)fminf(hix, (float)(P.w - 1));
    b.r0 = (int)fmaxf(loy, (float)P.row0);
    b.r1 = (int)fminf(hiy, (float)(P.row1 - 1));
    if (!finite || b.c0 > b.c1 || b.r0 > b.r1 || !(hix >= 0.f) || !(hiy >= 0.f))
        b.cls = 2;
    else if (!(b.ex <= (float)RCAP_PX && b.ey <= (float)RCAP_PX))
        b.cls = 1;
    else
        b.cls = 0;
    return b;
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    return v;
}

// wave64 sum; result valid in every lane (butterfly)
__device__ __forceinline__ float wave_sum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---------------------------------------------------------------------------------------------------
// plan kernels
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_classify(Params P, const float *__restrict__ sigmas,
                                                  const float *__restrict__ coords, PlanView V)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // pixel-centre tables: the reference's double expression, rounded to float (gs_cuda/gs.cu:27-28)
    if (i < P.w) V.px[i] = (float)(2.0 * (double)i / (double)(P.w - 1) - 1.0);
    if (i < P.h) V.py[i] = (float)(2.0 * (double)i / (double)(P.h - 1) - 1.0);
    unsigned rx = 0, ry = 0;
    if (i < P.s) {
        const float sx = sigmas[i * 3 + 0], sy = sigmas[i * 3 + 1];
        const float x = coords[i * 2 + 0], y = coords[i * 2 + 1];
        const Box b = gaussian_box(sx, sy, x, y, P);
        unsigned key;
        if (b.cls == 2) {
            key = (unsigned)P.ncells + 1u;
        } else if (b.cls == 1) {
            key = (unsigned)P.ncells;
        } else {
            const float hx = 0.5f * (float)(P.w - 1), hy = 0.5f * (float)(P.h - 1);
            int cx = (int)fminf(fmaxf(floorf((x + 1.f) * hx), 0.f), (float)(P.w - 1)) >> CELL_SHIFT;
            int cy = (int)fminf(fmaxf(floorf((y + 1.f) * hy), 0.f), (float)(P.h - 1)) >> CELL_SHIFT;
            key = (unsigned)(cy * P.ncx + cx);
            rx = (unsigned)ceilf(b.ex) + 2u;
            ry = (unsigned)ceilf(b.ey) + 2u;
        }
        V.key[i] = key;
        atomicAdd(&V.cell_count[key], 1u);
    }
    rx = wave_max_u32(rx);
    ry = wave_max_u32(ry);
    if ((threadIdx.x & 63) == 0) {
        if (rx) atomicMax(&V.hdr[0], rx);
        if (ry) atomicMax(&V.hdr[1], ry);
    }
}

__global__ __launch_bounds__(1024) void k_scan(int n, const unsigned *__restrict__ count,
                                               unsigned *__restrict__ start)
{
    __shared__ unsigned part[1024];
    const int t = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = t * per, e = min(n, b + per);
    unsigned sum = 0;
    for (int k = b; k < e; ++k) sum += count[k];
    part[t] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {  // Hillis-Steele inclusive scan of the 1024 partials
        unsigned v = t >= o ? part[t - o] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned run = part[t] - sum;
    for (int k = b; k < e; ++k) {
        start[k] = run;
        run += count[k];
    }
    if (t == 1023) start[n] = part[1023];
}

__global__ __launch_bounds__(256) void k_scatter(Params P, PlanView V)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.s) return;
    const unsigned key = V.key[i];
    const unsigned pos = V.cell_start[key] + atomicAdd(&V.cell_cursor[key], 1u);
    V.perm[pos] = (unsigned)i;
}

__global__ __launch_bounds__(256) void k_pack(Params P, const float *__restrict__ sigmas,
                                              const float *__restrict__ coords,
                                              const float *__restrict__ colors, PlanView V)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= P.s) return;
    const unsigned i = V.perm[j];
    const float sx = sigmas[i * 3 + 0], sy = sigmas[i * 3 + 1], rho = sigmas[i * 3 + 2];
    const float x = coords[i * 2 + 0], y = coords[i * 2 + 1];
    const Box b = gaussian_box(sx, sy, x, y, P);
    // exponent = w1*(dx^2/sx^2 - 2 rho dx dy/(sx sy) + dy^2/sy^2), w1 = -0.5/(1-rho^2)   (gs.cu:33-56)
    const double dr = rho, dsx = sx, dsy = sy;
    const double w1 = -0.5 / (1.0 - dr * dr) * LOG2E;
    const float A = (float)(w1 / (dsx * dsx));
    const float B = (float)(-2.0 * dr * w1 / (dsx * dsy));
    const float C = (float)(w1 / (dsy * dsy));
    V.rec[2 * j + 0] = make_float4(x, y, A, B);
    V.rec[2 * j + 1] = make_float4(C, colors[i * 3 + 0], colors[i * 3 + 1], colors[i * 3 + 2]);
    short4 bb;
    if (b.cls == 2) { bb.x = 1; bb.y = 0; bb.z = 1; bb.w = 0; }
    else { bb.x = (short)b.c0; bb.y = (short)b.c1; bb.z = (short)b.r0; bb.w = (short)b.r1; }
    V.bbox[j] = bb;
}

// ---------------------------------------------------------------------------------------------------
// forward: one wave64 per 8x8 pixel sub-tile, four sub-tiles side by side per workgroup (32x8 px)
// ---------------------------------------------------------------------------------------------------
template <bool BOUNDED>
__device__ __forceinline__ void fwd_segment(unsigned beg, unsigned end, int lane, int sx0, int sx1, int sy0,
                                            int sy1, float px, float py, float dmax,
                                            const float4 *__restrict__ rec, const short4 *__restrict__ bbox,
                                            float &ar, float &ag, float &ab)
{
    for (unsigned base = beg; base < end; base += 64) {
        const unsigned j = base + (unsigned)lane;
        bool hit = false;
        if (j < end) {  // one 8-byte load: {c0 | c1<<16, r0 | r1<<16} as signed 16-bit pixel indices
            const uint2 bb = reinterpret_cast<const uint2 *>(bbox)[j];
            const int c0 = (int)(short)(bb.x & 0xffffu), c1 = (int)bb.x >> 16;
            const int r0 = (int)(short)(bb.y & 0xffffu), r1 = (int)bb.y >> 16;
            hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
        }
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int k = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float4 *r = rec + 2 * (size_t)(base + (unsigned)k);
            const float4 r0 = r[0], r1 = r[1];  // wave-uniform address -> s_load_dwordx8
            const float dx = px - r0.x, dy = py - r0.y;
            const float t = fmaf(r0.w, dy, r0.z * dx);
            const float pw = fmaf(dx, t, r1.x * dy * dy);
            float v = __builtin_amdgcn_exp2f(pw);
            if (BOUNDED) v = (fabsf(dx) <= dmax && fabsf(dy) <= dmax) ? v : 0.f;
            ar = fmaf(v, r1.y, ar);
            ag = fmaf(v, r1.z, ag);
            ab = fmaf(v, r1.w, ab);
        }
    }
}

template <bool BOUNDED>
__global__ __launch_bounds__(256) void k_render_fwd(Params P, PlanView V, float *__restrict__ img, int tiles_x,
                                                    int tiles_y)
{
    // XCD-aware tile order: blocks are dealt round-robin to the 8 XCDs (block b -> XCD b%8), so give
    // each XCD a contiguous band of tile rows: neighbouring tiles then share records in ONE L2.
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u;
    const unsigned t = xcd * q + min(xcd, r) + (b >> 3);
    const int bx = (int)(t % (unsigned)tiles_x), by = (int)(t / (unsigned)tiles_x);
    (void)tiles_y;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave id is uniform: keep it in an SGPR
    const int sx0 = (bx * 4 + wv) * SUB, sy0 = P.row0 + by * SUB;
    if (sx0 >= P.w) return;  // wave-uniform
    const int sx1 = min(sx0 + SUB - 1, P.w - 1), sy1 = min(sy0 + SUB - 1, P.row1 - 1);
    const int X = sx0 + (lane & 7), Y = sy0 + (lane >> 3);
    const bool live = X < P.w && Y < P.row1;
    const float px = V.px[min(X, P.w - 1)], py = V.py[min(Y, P.h - 1)];
    float ar = 0.f, ag = 0.f, ab = 0.f;

    const float4 *__restrict__ rec = V.rec;
    const short4 *__restrict__ bbox = V.bbox;
    const unsigned *__restrict__ cs = V.cell_start;
    // normal class: cells whose Gaussians can reach this sub-tile (max half-extent from the plan header)
    const int rx = (int)V.hdr[0], ry = (int)V.hdr[1];
    if (rx > 0) {
        const int cx0 = max(sx0 - rx, 0) >> CELL_SHIFT, cx1 = min((sx1 + rx) >> CELL_SHIFT, P.ncx - 1);
        const int cy0 = max(sy0 - ry, 0) >> CELL_SHIFT, cy1 = min((sy1 + ry) >> CELL_SHIFT, P.ncy - 1);
        for (int cy = cy0; cy <= cy1; ++cy) {
            const unsigned beg = cs[cy * P.ncx + cx0], end = cs[cy * P.ncx + cx1 + 1];
            fwd_segment<BOUNDED>(beg, end, lane, sx0, sx1, sy0, sy1, px, py, P.dmax, rec, bbox, ar, ag, ab);
        }
    }
    // large class: every wave tests all of them
    fwd_segment<BOUNDED>(cs[P.ncells], cs[P.ncells + 1], lane, sx0, sx1, sy0, sy1, px, py, P.dmax, rec, bbox,
                         ar, ag, ab);
    if (live) {
        float *o = img + ((size_t)(Y - P.row0) * P.w + X) * 3;
        o[0] += ar;
        o[1] += ag;
        o[2] += ab;
    }
}

// ---------------------------------------------------------------------------------------------------
// backward: one wave64 per Gaussian (cell order, so neighbouring waves read neighbouring pixels)
// ---------------------------------------------------------------------------------------------------
template <bool BOUNDED>
__device__ __forceinline__ void bwd_item(unsigned j, int chunk, bool atomic, int lane, const Params &P,
                                         const PlanView &V, const float *__restrict__ sigmas,
                                         const float *__restrict__ coords, const float *__restrict__ colors,
                                         const float *__restrict__ grad, float *__restrict__ g_sigmas,
                                         float *__restrict__ g_coords, float *__restrict__ g_colors)
{
    const unsigned i = V.perm[j];
    const float sx = sigmas[i * 3 + 0], sy = sigmas[i * 3 + 1], rho = sigmas[i * 3 + 2];
    const float x = coords[i * 2 + 0], y = coords[i * 2 + 1];
    const float cr = colors[i * 3 + 0], cg = colors[i * 3 + 1], cb = colors[i * 3 + 2];
    const Box b = gaussian_box(sx, sy, x, y, P);
    if (b.cls == 2) return;
    int r0 = b.r0, r1 = b.r1;
    if (chunk >= 0) {
        const int rpc = (b.r1 - b.r0 + NCH) / NCH;
        r0 = b.r0 + chunk * rpc;
        r1 = min(b.r1, r0 + rpc - 1);
        if (r0 > r1) return;
    }
    const double dr = rho, dsx = sx, dsy = sy;
    const double w1d = -0.5 / (1.0 - dr * dr);
    const double w2d = 1.0 / (dsx * dsx), w3d = 1.0 / (dsx * dsy), w4d = 1.0 / (dsy * dsy);
    const float A = (float)(w1d * LOG2E * w2d), B = (float)(-2.0 * dr * w1d * LOG2E * w3d),
                C = (float)(w1d * LOG2E * w4d);

    const int bw = b.c1 - b.c0 + 1;
    const int npx = bw * (r1 - r0 + 1);
    const int kstep = 64 / bw, rstep = 64 % bw;
    int row = lane / bw, col = lane % bw;
    float Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, Cr = 0.f, Cg = 0.f, Cb = 0.f;
    const float *__restrict__ pxt = V.px;
    const float *__restrict__ pyt = V.py;
    for (int idx = lane; idx < npx; idx += 64) {
        const int X = b.c0 + col, Y = r0 + row;
        const float dx = pxt[X] - x, dy = pyt[Y] - y;
        const float *g = grad + ((size_t)(Y - P.row0) * P.w + X) * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        const float t = fmaf(B, dy, A * dx);
        const float pw = fmaf(dx, t, C * dy * dy);
        float v = __builtin_amdgcn_exp2f(pw);
        if (BOUNDED) v = (fabsf(dx) <= P.dmax && fabsf(dy) <= P.dmax) ? v : 0.f;
        const float gp = fmaf(g2, cb, fmaf(g1, cg, g0 * cr));  // dL/dv summed over channels (gs.cu:150)
        const float qv = gp * v, qdx = qv * dx, qdy = qv * dy;
        Sx += qdx;
        Sy += qdy;
        Sxx = fmaf(qdx, dx, Sxx);
        Sxy = fmaf(qdx, dy, Sxy);
        Syy = fmaf(qdy, dy, Syy);
        Cr = fmaf(v, g0, Cr);
        Cg = fmaf(v, g1, Cg);
        Cb = fmaf(v, g2, Cb);
        col += rstep;
        row += kstep;
        if (col >= bw) { col -= bw; ++row; }
    }
    Sx = wave_sum(Sx); Sy = wave_sum(Sy); Sxx = wave_sum(Sxx); Sxy = wave_sum(Sxy); Syy = wave_sum(Syy);
    Cr = wave_sum(Cr); Cg = wave_sum(Cg); Cb = wave_sum(Cb);
    if (lane == 0) {
        // the per-pixel partials of gs.cu:139-146 are linear in {q dx, q dy, q dx^2, q dx dy, q dy^2},
        // so the Gaussian-constant factors are applied once, in double, to the five moment sums
        const double rw3 = dr * w3d, two_w1 = 2.0 * w1d;
        const double gx = two_w1 * (-w2d * Sx + rw3 * Sy);
        const double gy = two_w1 * (-w4d * Sy + rw3 * Sx);
        const double gsx = two_w1 / dsx * (rw3 * Sxy - w2d * Sxx);
        const double gsy = two_w1 / dsy * (rw3 * Sxy - w4d * Syy);
        const double qd = w2d * Sxx - 2.0 * rw3 * Sxy + w4d * Syy;
        const double grho = -two_w1 * (two_w1 * dr * qd + w3d * Sxy);
        float *os = g_sigmas + (size_t)i * 3, *op = g_coords + (size_t)i * 2, *oc = g_colors + (size_t)i * 3;
        if (atomic) {
            atomicAdd(os + 0, (float)gsx); atomicAdd(os + 1, (float)gsy); atomicAdd(os + 2, (float)grho);
            atomicAdd(op + 0, (float)gx);  atomicAdd(op + 1, (float)gy);
            atomicAdd(oc + 0, Cr); atomicAdd(oc + 1, Cg); atomicAdd(oc + 2, Cb);
        } else {
            os[0] += (float)gsx; os[1] += (float)gsy; os[2] += (float)grho;
            op[0] += (float)gx;  op[1] += (float)gy;
            oc[0] += Cr; oc[1] += Cg; oc[2] += Cb;
        }
    }
}

template <bool BOUNDED>
__global__ __launch_bounds__(256) void k_render_bwd(Params P, PlanView V, const float *__restrict__ sigmas,
                                                    const float *__restrict__ coords,
                                                    const float *__restrict__ colors,
                                                    const float *__restrict__ grad, float *__restrict__ g_sigmas,
                                                    float *__restrict__ g_coords, float *__restrict__ g_colors)
{
    const int lane = threadIdx.x & 63;
    const unsigned gw = blockIdx.x * 4u + (unsigned)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned nwaves = gridDim.x * 4u;
    const unsigned large_beg = V.cell_start[P.ncells], large_end = V.cell_start[P.ncells + 1];
    if (gw < large_beg)
        bwd_item<BOUNDED>(gw, -1, false, lane, P, V, sigmas, coords, colors, grad, g_sigmas, g_coords, g_colors);
    else if (gw < large_end)
        bwd_item<BOUNDED>(gw, 0, true, lane, P, V, sigmas, coords, colors, grad, g_sigmas, g_coords, g_colors);
    // remaining row chunks of the large class, spread over all waves
    const unsigned extra = (large_end - large_beg) * (unsigned)(NCH - 1);
    for (unsigned it = gw; it < extra; it += nwaves) {
        const unsigned j = large_beg + it / (unsigned)(NCH - 1);
        const int chunk = 1 + (int)(it % (unsigned)(NCH - 1));
        bwd_item<BOUNDED>(j, chunk, true, lane, P, V, sigmas, coords, colors, grad, g_sigmas, g_coords, g_colors);
    }
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                    \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

int check_ws(const gsasr_dims *dims, const void *ws, size_t ws_bytes, Layout &L)
{
    if (!dims_ok(dims)) return fail(GSASR_ERR_ARG, "bad dims (need c==3, 2<=h,w<=32767, 0<=row0<=row1<=h)");
    L = make_layout(dims);
    if (!ws || ((uintptr_t)ws & 255u)) return fail(GSASR_ERR_WORKSPACE, "workspace null or not 256-byte aligned");
    if (ws_bytes < L.total) return fail(GSASR_ERR_WORKSPACE, "workspace smaller than gsasr_splat_workspace_bytes()");
    return GSASR_OK;
}

}  // namespace

extern "C" {

int gsasr_abi_version(void) { return GSASR_SPLAT_ABI_VERSION; }

const char *gsasr_last_error(void) { return tl_err; }

void gsasr_set_default_cutoff(float tau) { g_default_cutoff = tau == 0.f ? GSASR_SPLAT_DEFAULT_CUTOFF : tau; }

float gsasr_get_default_cutoff(void) { return default_cutoff(); }

size_t gsasr_splat_workspace_bytes(const gsasr_dims *dims)
{
    if (!dims_ok(dims)) {
        fail(GSASR_ERR_ARG, "bad dims");
        return 0;
    }
    return make_layout(dims).total;
}

int gsasr_splat_plan(const float *sigmas, const float *coords, const float *colors, const gsasr_dims *dims,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L)) return rc;
    if (dims->s > 0 && (!sigmas || !coords || !colors)) return fail(GSASR_ERR_ARG, "null input pointer");
    hipStream_t st = (hipStream_t)stream;
    const Params P = make_params(dims, L);
    const PlanView V = make_view(L, workspace);
    HIP_TRY(hipMemsetAsync(workspace, 0, L.zero_bytes, st));
    const int nthreads = dims->s > dims->w ? (dims->s > dims->h ? dims->s : dims->h)
                                           : (dims->w > dims->h ? dims->w : dims->h);
    hipLaunchKernelGGL(k_classify, dim3((nthreads + 255) / 256), dim3(256), 0, st, P, sigmas, coords, V);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, L.ncells + 2, V.cell_count, V.cell_start);
    if (dims->s > 0) {
        const int nb = (dims->s + 255) / 256;
        hipLaunchKernelGGL(k_scatter, dim3(nb), dim3(256), 0, st, P, V);
        hipLaunchKernelGGL(k_pack, dim3(nb), dim3(256), 0, st, P, sigmas, coords, colors, V);
    }
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_splat_forward(const gsasr_dims *dims, const void *workspace, size_t workspace_bytes, float *img,
                        void *stream)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L)) return rc;
    const int rows = dims->row1 - dims->row0;
    if (rows == 0) return GSASR_OK;
    if (!img) return fail(GSASR_ERR_ARG, "null image pointer");
    const Params P = make_params(dims, L);
    const PlanView V = make_view(L, const_cast<void *>(workspace));
    const int tiles_x = (dims->w + 4 * SUB - 1) / (4 * SUB), tiles_y = (rows + SUB - 1) / SUB;
    const dim3 grid((unsigned)tiles_x * (unsigned)tiles_y), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (P.bounded)
        hipLaunchKernelGGL(k_render_fwd<true>, grid, block, 0, st, P, V, img, tiles_x, tiles_y);
    else
        hipLaunchKernelGGL(k_render_fwd<false>, grid, block, 0, st, P, V, img, tiles_x, tiles_y);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_splat_backward(const float *sigmas, const float *coords, const float *colors, const float *grad_img,
                         float *g_sigmas, float *g_coords, float *g_colors, const gsasr_dims *dims,
                         const void *workspace, size_t workspace_bytes, void *stream)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L)) return rc;
    if (dims->s == 0 || dims->row1 == dims->row0) return GSASR_OK;
    if (!sigmas || !coords || !colors || !grad_img || !g_sigmas || !g_coords || !g_colors)
        return fail(GSASR_ERR_ARG, "null pointer");
    const Params P = make_params(dims, L);
    const PlanView V = make_view(L, const_cast<void *>(workspace));
    const dim3 grid((unsigned)((dims->s + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (P.bounded)
        hipLaunchKernelGGL(k_render_bwd<true>, grid, block, 0, st, P, V, sigmas, coords, colors, grad_img,
                           g_sigmas, g_coords, g_colors);
    else
        hipLaunchKernelGGL(k_render_bwd<false>, grid, block, 0, st, P, V, sigmas, coords, colors, grad_img,
                           g_sigmas, g_coords, g_colors);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

// ---- reference-shaped launchers -------------------------------------------------------------------
static int render_common(const float *sigmas, const float *coords, const float *colors, float *img,
                         const float *grads, float *gs, float *gc, float *gk, int s, int h, int w, int c,
                         float dmax, bool backward, void *stream)
{
    gsasr_dims d;
    d.s = s; d.h = h; d.w = w; d.c = c; d.dmax = dmax; d.row0 = 0; d.row1 = h; d.cutoff = 0.f; d.flags = 0;
    const size_t bytes = gsasr_splat_workspace_bytes(&d);
    if (!bytes) return GSASR_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    void *ws = nullptr;
    HIP_TRY(hipMallocAsync(&ws, bytes, st));
    int rc = gsasr_splat_plan(sigmas, coords, colors, &d, ws, bytes, stream);
    if (rc == GSASR_OK) {
        if (!backward) {
            rc = gsasr_splat_forward(&d, ws, bytes, img, stream);
        } else {
            if (dmax < 0.f && s > 0) {  // gs_cuda backward overwrites its outputs (gs.cu:169-176)
                (void)hipMemsetAsync(gs, 0, sizeof(float) * 3 * (size_t)s, st);
                (void)hipMemsetAsync(gc, 0, sizeof(float) * 2 * (size_t)s, st);
                (void)hipMemsetAsync(gk, 0, sizeof(float) * 3 * (size_t)s, st);
            }
            rc = gsasr_splat_backward(sigmas, coords, colors, grads, gs, gc, gk, &d, ws, bytes, stream);
        }
    }
    hipError_t e = hipFreeAsync(ws, st);
    if (rc == GSASR_OK && e != hipSuccess) return hip_fail(e, "hipFreeAsync");
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
