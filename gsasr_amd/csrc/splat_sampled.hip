// splat_sampled.hip -- sampled pixels (the reference's sample_coords): sort of the points, point-stationary forward, Gaussian-stationary backward
// (one translation unit of libgsasr_splat.so; gsasr_splat.hip has the overview of the whole pipeline)
#include "splat_common.h"

using namespace gsasr_detail;

namespace {

// ---------------------------------------------------------------------------------------------------
// sampled pixels.  The reference renders the whole image and then picks `sample_coords`
// out of it (utils/gaussian_splatting.py:214-216); here only the requested points are evaluated.
//   k_pts_count / k_pts_scan / k_pts_place   counting sort of the points into point-cells (8x8 px; coarser when the
//                 image has more than PT_CELLS of those); a sorted point carries its px, py; out-of-range points go to
//                 a last bucket.  k_pts_grads (backward) gathers the upstream gradient into the same order.
//   k_sample_fwd  POINT-stationary two-level walk: one workgroup = the points of a 16x16-px block; level 1 lists the
//                 Gaussians whose window meets the block, level 2 evaluates each of them (one per lane, loaded once) at
//                 all the block's points; no atomics on the output.
//   k_sample_bwd  GAUSSIAN-stationary, eight Gaussians per wave64: a Gaussian's 8 lanes stride over the sorted points
//                 of the point-cells its window touches; same sums and epilogue as k_render_bwd.
// ---------------------------------------------------------------------------------------------------
constexpr int PT_CELLS = 12288;                    // point-cells at most: their histogram + scan live in LDS (48 KB)
constexpr int PT_MIN_SHIFT = 3, PT_MAX_SHIFT = 9;  // point-cells are 8..512 px a side

struct PtView {
    unsigned *start;   // [ncx*ncy + 2] exclusive scan of the points per point-cell; [ncx*ncy] = first invalid point
    unsigned *cursor;  // [ncx*ncy + 1] fill positions of the counting sort
    float4 *sorted;    // [n] {px, py, X | canvas row << 16, original index}: everything a kernel needs about a point
    float4 *grads;     // [n] {g_r, g_g, g_b, -} of the sorted points (backward)
    int shx, shy, ncx, ncy;
};

__device__ __forceinline__ int2 point_rc(int2 raw, const Geo &g)
{
    int r = raw.x, c = raw.y;
    if (r < 0) r += g.h;   // Python's wrap-around of negative indices
    if (c < 0) c += g.w;
    return make_int2(r, c);
}

// The counting sort as three launches (the histogram is zeroed by a memset): count, scan (one workgroup), place.
__device__ __forceinline__ int point_cell(const Params &P, const PlanView &V, const PtView &S, const int *__restrict__ pts,
                                          int i, int n_per, int &X, int &Y, int &pxo)
{
    const Geo g = sample_geo(P, V, i / n_per);
    const int2 rc = point_rc(reinterpret_cast<const int2 *>(pts)[i], g);
    const bool ok = rc.x >= 0 && rc.x < g.h && rc.y >= 0 && rc.y < g.w;
    X = rc.y; Y = g.base + rc.x; pxo = g.pxo;
    return ok ? (Y >> S.shy) * S.ncx + (X >> S.shx) : S.ncx * S.ncy;
}

__global__ __launch_bounds__(256) void k_pts_count(Params P, PlanView V, PtView S, const int *__restrict__ pts,
                                                   int n_total, int n_per)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    int X, Y, pxo;
    atomicAdd(&S.start[point_cell(P, V, S, pts, i, n_per, X, Y, pxo)], 1u);
}

__global__ __launch_bounds__(1024) void k_pts_scan(PtView S, int n_total)
{
    __shared__ unsigned s_c[PT_CELLS + 1], s_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int npc = S.ncx * S.ncy;
    // exclusive scan of npc + 1 counts in place, through LDS so that global memory is read and written coalesced:
    // consecutive entries per thread, wave scan, 16 wave totals
    for (int e = tid; e <= npc; e += 1024) s_c[e] = S.start[e];
    __syncthreads();
    constexpr int PER = (PT_CELLS + 1 + 1023) / 1024;
    unsigned loc[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = tid * PER + k;
        loc[k] = e <= npc ? s_c[e] : 0u;
        sum += loc[k];
    }
    unsigned inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)inc, o);
        if (lane >= o) inc += v;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    unsigned run = inc - sum;
    for (int k = 0; k < wv; ++k) run += s_wave[k];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = tid * PER + k;
        if (e <= npc) s_c[e] = run;
        run += loc[k];
    }
    __syncthreads();
    for (int e = tid; e <= npc; e += 1024) {
        const unsigned v = s_c[e];
        S.start[e] = v;
        S.cursor[e] = v;
    }
    if (tid == 0) S.start[npc + 1] = (unsigned)n_total;
}

__global__ __launch_bounds__(256) void k_pts_place(Params P, PlanView V, PtView S, const int *__restrict__ pts,
                                                   int n_total, int n_per)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    int X, Y, pxo;
    const int cell = point_cell(P, V, S, pts, i, n_per, X, Y, pxo);
    const bool ok = cell < S.ncx * S.ncy;
    const float px = ok ? V.px[pxo + X] : 0.f, py = ok ? V.py[Y] : 0.f;
    const unsigned pos = atomicAdd(&S.cursor[cell], 1u);
    S.sorted[pos] = make_float4(px, py, ok ? __uint_as_float((unsigned)X | ((unsigned)Y << 16)) : 0.f, __uint_as_float((unsigned)i));
}

// backward: the upstream gradient [B, 3, n_per] gathered into the sorted order, so that a candidate is two
// 16-byte loads at one index
__global__ __launch_bounds__(256) void k_pts_grads(PtView S, const float *__restrict__ grad_out, int n_total, int n_per)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_total) return;
    const int idx = (int)__float_as_uint(S.sorted[i].w), smp = idx / n_per;
    const float *g = grad_out + (size_t)smp * 3 * n_per + (idx - smp * n_per);
    S.grads[i] = make_float4(g[0], g[(size_t)n_per], g[2 * (size_t)n_per], 0.f);
}

// wave64 sum without LDS traffic: four DPP row shifts leave each row's total in its lane 0
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v += dpp_row_shl<8>(v);
    v += dpp_row_shl<4>(v);
    v += dpp_row_shl<2>(v);
    v += dpp_row_shl<1>(v);
    const int i = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(i, 48)));
}

// Forward at the points of ONE 16x16-px block of point-cells per workgroup, as a two-level walk (cf. fwd_block).
// Level 1: the four waves stride through the Gaussians binned within reach of the block's rectangle (row of plan cells by
// row, SAMPLE_CHUNKS dense windows in flight per lane) and append those whose window meets the rectangle to a survivor
// list in LDS.  Level 2: every lane takes a survivor, loads its record ONCE (the next one prefetched) and evaluates it at
// all the block's points -- staged in LDS, read as broadcasts, two points per packed-fp32 instruction -- into per-lane
// accumulators that live in registers across the whole walk; one DPP reduction over the wave and an LDS combine over
// the waves at the end.  128 VGPRs (the accumulators), 33 KB of LDS (the list): four workgroups per CU.  The last
// workgroup zeroes the outputs of the out-of-range points.
constexpr int SAMPLE_WAVES = 4;
constexpr int SAMPLE_CHUNKS = 8;      // candidate windows in flight per lane (level 1)
constexpr int SAMPLE_LIST = 8192;     // capacity of the survivor list = candidates tested between two level-2 passes
constexpr int SAMPLE_BLOCK = 24;      // points evaluated per walk: 72 accumulator VGPRs (28: 22 spills at 4 waves per SIMD; 16: 45% of the blocks walk twice)

// One Gaussian per lane against the points staged in LDS (broadcast reads), two points per packed-fp32 operation.
template <bool TEST>
__device__ __forceinline__ void sample_eval(const float4 *s_pt, int npb, const float4 a, const float4 b, float dmax,
                                            v2f (&acc)[SAMPLE_BLOCK / 2][3])
{
#pragma unroll
    for (int k = 0; k < SAMPLE_BLOCK / 2; ++k) {
        if (2 * k >= npb) continue;   // (uniform; an odd last point pairs with a stale entry that is never written out)
        const float4 p0 = s_pt[2 * k], p1 = s_pt[2 * k + 1];
        const v2f dx = (v2f){p0.x, p1.x} - a.x, dy = (v2f){p0.y, p1.y} - a.y;
        const v2f u = a.z * dx;
        const v2f bq = b.x * dy + a.w * u;
        const v2f pw = -(u * u) - bq * bq;
        v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
        if (TEST) {   // (dmax = +inf for the lanes whose Gaussian needs no test)
            v.x = fmaxf(fabsf(dx.x), fabsf(dy.x)) <= dmax ? v.x : 0.f;
            v.y = fmaxf(fabsf(dx.y), fabsf(dy.y)) <= dmax ? v.y : 0.f;
        }
        acc[k][0] += v * b.y;
        acc[k][1] += v * b.z;
        acc[k][2] += v * b.w;
    }
}

template <bool BOUNDED>
__global__ __launch_bounds__(64 * SAMPLE_WAVES) __attribute__((amdgpu_waves_per_eu(4))) void k_sample_fwd(Params P, PlanView V, PtView S, int n_per,
                                                                 float *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the workgroup's block of point-cells: 16x16 px (2x2 cells of 8 px), or one coarser cell
    const int fsx = max(S.shx, CELL_SHIFT), fsy = max(S.shy, CELL_SHIFT);
    const int nbx = ((P.w - 1) >> fsx) + 1, nby = ((P.h - 1) >> fsy) + 1, npc = S.ncx * S.ncy;
    // (XCD-aware order, as in the full forward: each XCD takes a contiguous band of blocks, whose survivors' records then
    // sit in ONE L2; the last workgroup handles the out-of-range points)
    const int blk = (int)blockIdx.x == nbx * nby ? nbx * nby : (int)xcd_swizzle(blockIdx.x, (unsigned)(nbx * nby));
    if (blk == nbx * nby) {   // the out-of-range points: nothing is rendered there
        const unsigned pbeg = S.start[npc], pend = S.start[npc + 1];
        for (unsigned i = pbeg + threadIdx.x; i < pend; i += 64 * SAMPLE_WAVES) {
            const int idx = (int)__float_as_uint(S.sorted[i].w), smp = idx / n_per;
            float *o = out + (size_t)smp * 3 * n_per + (idx - smp * n_per);
            o[0] = 0.f; o[(size_t)n_per] = 0.f; o[2 * (size_t)n_per] = 0.f;
        }
        return;
    }
    const int bxi = blk % nbx, byi = blk / nbx;
    const int px0 = bxi << (fsx - S.shx), px1 = min(px0 + (1 << (fsx - S.shx)), S.ncx);   // point-cell columns [px0, px1)
    const int py0 = byi << (fsy - S.shy), two = (fsy > S.shy && py0 + 1 < S.ncy) ? 1 : 0;  // one or two rows of them
    const unsigned beg0 = S.start[py0 * S.ncx + px0], n0 = S.start[py0 * S.ncx + px1] - beg0;
    const unsigned beg1 = two ? S.start[(py0 + 1) * S.ncx + px0] : 0u;
    const unsigned n1 = two ? S.start[(py0 + 1) * S.ncx + px1] - beg1 : 0u;
    const unsigned pbeg = 0u, pend = n0 + n1;   // the block's points, numbered through both rows
    if (pend == 0u) return;
    __shared__ unsigned s_list[SAMPLE_LIST];
    __shared__ unsigned s_cnt[2];
    __shared__ float4 s_pt[SAMPLE_BLOCK];      // {px, py, X | Y << 16, original index}
    __shared__ float s_acc[3 * SAMPLE_BLOCK];
    const int bx0 = bxi << fsx, by0 = byi << fsy;
    const int bx1 = min(bx0 + (1 << fsx), P.w) - 1, by1 = min(by0 + (1 << fsy), P.h) - 1;
    const float4 *__restrict__ rec = V.rec;
    const unsigned *__restrict__ cs = V.cell_start;

    // segment table of the rectangle (every wave builds the same one)
    const int rx = (int)V.hdr[8], ry = (int)V.hdr[9];
    int nseg = 0;
    unsigned sbeg = 0, send = 0;
    if (rx > 0) {
        const int cx0 = max(bx0 - rx, 0) >> CELL_SHIFT, cx1 = min((bx1 + rx) >> CELL_SHIFT, P.ncx - 1);
        const int cy0 = max(by0 - ry, 0) >> CELL_SHIFT, cy1 = min((by1 + ry) >> CELL_SHIFT, P.ncy - 1);
        nseg = cy1 - cy0 + 1;     // <= (512 + 2*128)/16 + 1 = 49 rows of plan cells
        if (lane < nseg) {
            sbeg = cs[(cy0 + lane) * P.ncx + cx0];
            send = cs[(cy0 + lane) * P.ncx + cx1 + 1];
        }
    }
    if (lane == nseg) {
        sbeg = cs[P.ncells];
        send = cs[P.ncells + 1];
    }
    ++nseg;
    const unsigned long long below = (1ull << lane) - 1ull;

    for (unsigned pb = pbeg; pb < pend; pb += SAMPLE_BLOCK) {   // (more than SAMPLE_BLOCK points in a cell: walk again)
        const int npb = (int)min((unsigned)SAMPLE_BLOCK, pend - pb);
        if ((int)threadIdx.x < npb) {
            const unsigned pi = pb + threadIdx.x;
            s_pt[threadIdx.x] = S.sorted[pi < n0 ? beg0 + pi : beg1 + (pi - n0)];
        }
        if (threadIdx.x < 3 * SAMPLE_BLOCK) s_acc[threadIdx.x] = 0.f;
        if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
        __syncthreads();
        v2f acc[SAMPLE_BLOCK / 2][3];   // per lane: colour sums of the block's points, two points per register pair
#pragma unroll
        for (int k = 0; k < SAMPLE_BLOCK / 2; ++k) acc[k][0] = acc[k][1] = acc[k][2] = (v2f){0.f, 0.f};
        // Level 1 fills the survivor list with the candidates row by row (cell rows within reach, then the large
        // class) until the next batch might overflow it or the candidates are exhausted; level 2 empties it.  (One
        // level-2 site in the code: inlined twice it spills 22 accumulator registers.)
        int r = 0;
        unsigned i0 = (unsigned)__builtin_amdgcn_readlane((int)sbeg, 0), se = (unsigned)__builtin_amdgcn_readlane((int)send, 0);
        constexpr unsigned BATCH = 64u * SAMPLE_WAVES * SAMPLE_CHUNKS;
        for (unsigned round = 0;; ++round) {
            unsigned *cnt = s_cnt + (round & 1u);
            // ---- level 1: the workgroup strides through a row, SAMPLE_CHUNKS windows in flight per lane, against
            // the block's rectangle
            for (unsigned proc = 0; r < nseg && proc + BATCH <= (unsigned)SAMPLE_LIST;) {
                if (i0 >= se) {
                    if (++r < nseg) {
                        i0 = (unsigned)__builtin_amdgcn_readlane((int)sbeg, r);
                        se = (unsigned)__builtin_amdgcn_readlane((int)send, r);
                    }
                    continue;
                }
                uint2 cw[SAMPLE_CHUNKS];
#pragma unroll
                for (int k = 0; k < SAMPLE_CHUNKS; ++k) {
                    const unsigned i = i0 + 64u * SAMPLE_WAVES * (unsigned)k + (unsigned)threadIdx.x;
                    cw[k] = make_uint2(0x7fffu, 0x7fffu);   // a window that overlaps nothing
                    if (i < se) cw[k] = V.win[i];
                }
#pragma unroll
                for (int k = 0; k < SAMPLE_CHUNKS; ++k) {
                    if (i0 + 64u * SAMPLE_WAVES * (unsigned)k >= se) continue;   // (uniform)
                    const int c0 = (int)(cw[k].x & 0x7fffu), c1 = (int)(cw[k].x >> 16);
                    const int r0 = (int)(cw[k].y & 0x7fffu), r1 = (int)(cw[k].y >> 16);
                    const bool hit = (c0 <= bx1) & (c1 >= bx0) & (r0 <= by1) & (r1 >= by0);
                    const unsigned long long m = __ballot(hit);
                    if (m) {
                        unsigned at = 0;
                        if (lane == 0) at = atomicAdd(cnt, (unsigned)__builtin_popcountll(m));
                        at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
                        // entry = index | "needs the dmax test" (window word bit 15) << 31
                        if (hit) s_list[at + (unsigned)__builtin_popcountll(m & below)] =
                            (i0 + 64u * SAMPLE_WAVES * (unsigned)k + (unsigned)threadIdx.x) | ((cw[k].x & 0x8000u) << 16);
                    }
                }
                i0 += BATCH;
                proc += BATCH;
            }
            __syncthreads();
            if (threadIdx.x == 0) s_cnt[(round + 1u) & 1u] = 0u;
            const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)*cnt);
            // ---- level 2: a survivor per lane, loaded once, evaluated at every point of the block -------------
            // No window test per point: a Gaussian's terms outside its window are below exp(-tau) (that is what the
            // window means), so adding them is as exact as skipping them; only the dmax box must be honoured.
            // (the next survivor's record is in flight while the current one is evaluated)
            unsigned q = (unsigned)wv * 64u;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;   // {x, y, IX, NR}, {IY, r, g, b}; dead lanes add 0 * v
            bool test = false;
            if (q + (unsigned)lane < n) {
                const unsigned e = s_list[q + lane], j = e & 0x7fffffffu;
                a = rec[2 * (size_t)j];
                b = rec[2 * (size_t)j + 1];
                test = (e >> 31) != 0u;
            }
            while (q < n) {
                asm volatile("" ::: "memory");   // re-read the points from LDS every trip: hoisted, they cost 96 VGPRs
                const unsigned nq = q + 64u * SAMPLE_WAVES;
                float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
                bool ntest = false;
                if (nq + (unsigned)lane < n) {
                    const unsigned e = s_list[nq + lane], j = e & 0x7fffffffu;
                    na = rec[2 * (size_t)j];
                    nb = rec[2 * (size_t)j + 1];
                    ntest = (e >> 31) != 0u;
                }
                // (one code path: an `if (any lane needs the test)` around two instantiations makes the compiler keep
                // two copies of the accumulators -- 200 spilled dwords; the test is 3 instructions per point)
                sample_eval<BOUNDED>(s_pt, npb, a, b, test ? P.dmax : INFINITY, acc);
                q = nq; a = na; b = nb; test = ntest;
            }
            if (r >= nseg) break;   // (uniform)
            __syncthreads();        // the list is rewritten by the next round
        }
        // one reduction per block: over the lanes with DPP, over the waves in LDS
#pragma unroll
        for (int k = 0; k < SAMPLE_BLOCK / 2; ++k) {
            if (2 * k >= npb) continue;   // (uniform)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t0 = wave_sum_dpp(acc[k][c].x), t1 = wave_sum_dpp(acc[k][c].y);
                if (lane == 0) {
                    atomicAdd(&s_acc[3 * (2 * k) + c], t0);
                    atomicAdd(&s_acc[3 * (2 * k + 1) + c], t1);
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 3 * npb) {
            const int k = threadIdx.x / 3, c = threadIdx.x - 3 * k;
            const int idx = (int)__float_as_uint(s_pt[k].w), smp = idx / n_per;
            out[((size_t)smp * 3 + c) * n_per + (idx - smp * n_per)] = s_acc[threadIdx.x];   // [B, 3, n_per]
        }
        __syncthreads();   // s_pt / s_acc are rewritten for the next block of points
    }
}

// Backward at the points: GAUSSIAN-stationary, SB_LANES lanes per Gaussian, i.e. eight Gaussians per wave64.  (With one
// wave per Gaussian the ~200 instructions of per-wave bookkeeping -- fetch, reduction, epilogue -- at 4 cycles each were
// the whole run time: 232 us for the 590 k Gaussians of config 5; 16 lanes: 104 us, 8: 85 us, 4: 80 us but a
// large-class Gaussian then walks every point with 4 lanes.)  A Gaussian's lanes stride over the sorted points of the
// point-cells its window touches, SB_ROWS rows of cells as one run of indices; DPP reduction inside the 16-lane row;
// the Gaussian's first lane writes the gradient.
constexpr int SB_LANES = 8;
constexpr int SB_ROWS = 4;

template <bool BOUNDED>
__global__ __launch_bounds__(256) void k_sample_bwd(Params P, PlanView V, PtView S,
                                                    float *__restrict__ g_sigmas, float *__restrict__ g_coords,
                                                    float *__restrict__ g_colors)
{
    constexpr float HALF_LOG2E = 0.72134752044448170368f;
    const int lane = threadIdx.x & 63, sl = lane & (SB_LANES - 1);
    // (XCD-banded order as in k_render_bwd: an XCD sweeps a contiguous run of the cell-ordered Gaussians = a band of points)
    const unsigned j = (xcd_swizzle(blockIdx.x, gridDim.x) * 256u + threadIdx.x) / SB_LANES;   // this lane's Gaussian (cell order)
    const bool valid = j < (unsigned)P.s;
    const size_t jj = valid ? j : (size_t)P.s - 1;
    const uint2 bb = *reinterpret_cast<const uint2 *>(V.bbox + 2 * jj);
    const float4 ra = V.rec[2 * jj], rb = V.rec[2 * jj + 1], fa = V.fin[2 * jj], fb = V.fin[2 * jj + 1];
    const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
    const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
    const bool dead = !valid || c0 > c1;
    const float x = ra.x, y = ra.y, cr = rb.y, cg = rb.z, cb = rb.w;
    const float cinv = fa.x, kappa = fa.y, rho = fa.z, isx = fa.w, isy = fb.x;
    const float nK1 = -HALF_LOG2E * cinv;
    const float dmax = (BOUNDED && (bb.x & 0x8000u)) ? P.dmax : INFINITY;
    const unsigned orig = __float_as_uint(fb.w);
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // The candidates: every point of the point-cells the window touches, SB_ROWS rows of cells at a time as ONE run
    // of indices (no per-row padding).  No window test per point: a point of a touched cell outside the window
    // carries a term below exp(-tau), and cells never straddle two samples of a batched canvas (make_pt_view).
    const int pcx0 = c0 >> S.shx, pcx1 = c1 >> S.shx, pcy1 = dead ? -1 : r1 >> S.shy;
    for (int row = dead ? 0 : r0 >> S.shy; row <= pcy1; row += SB_ROWS) {
        unsigned beg[SB_ROWS], cum[SB_ROWS];   // first point of row k; points in rows 0..k
#pragma unroll
        for (int k = 0; k < SB_ROWS; ++k) {
            const bool ok = row + k <= pcy1;
            const unsigned *p = S.start + (size_t)(ok ? row + k : row) * S.ncx;
            beg[k] = p[pcx0];
            cum[k] = ok ? p[pcx1 + 1] - beg[k] : 0u;
        }
#pragma unroll
        for (int k = 1; k < SB_ROWS; ++k) cum[k] += cum[k - 1];
        for (unsigned f = (unsigned)sl; f < cum[SB_ROWS - 1]; f += SB_LANES) {
            unsigned i = beg[0] + f;
#pragma unroll
            for (int k = 1; k < SB_ROWS; ++k) i = f >= cum[k - 1] ? beg[k] + (f - cum[k - 1]) : i;
            const float4 pt = S.sorted[i], gr = S.grads[i];
            const float dx = pt.x - x, dy = pt.y - y;
            const float u = dx * isx, vy = dy * isy, B = vy - rho * u;   // see bwd_trip
            float v = __builtin_amdgcn_exp2f((B * nK1) * B - HALF_LOG2E * u * u);
            if (BOUNDED) v = fmaxf(fabsf(dx), fabsf(dy)) <= dmax ? v : 0.f;
            const float q = fmaf(gr.z, cb, fmaf(gr.y, cg, gr.x * cr)) * v;
            const float A = u * kappa - rho * B, qA = q * A, qB = q * B;
            a[0] += qA; a[1] += qB; a[2] += qA * u; a[3] += qB * vy; a[4] += qA * B;
            a[5] += v * gr.x; a[6] += v * gr.y; a[7] += v * gr.z;
        }
    }
    if (!dead) bwd_scale(a, cinv, isx, isy);
    // sum over the Gaussian's lanes (within one DPP row of 16): its first lane gets the totals
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = a[k];
        if (SB_LANES > 8) v += dpp_row_shl<8>(v);
        if (SB_LANES > 4) v += dpp_row_shl<4>(v);
        v += dpp_row_shl<2>(v);
        v += dpp_row_shl<1>(v);
        a[k] = v;
    }
    if (sl != 0 || !valid) return;
    const bool store = (P.flags & GSASR_FLAG_OVERWRITE_GRADS) != 0u;
    if (dead && !store) return;
    float *pc = g_coords + (size_t)orig * stride2(P), *ps = g_sigmas + (size_t)orig * stride3(P),
          *pk = g_colors + (size_t)orig * stride3(P);
    if (store) {
        pc[0] = a[0]; pc[1] = a[1]; ps[0] = a[2]; ps[1] = a[3]; ps[2] = a[4]; pk[0] = a[5]; pk[1] = a[6]; pk[2] = a[7];
    } else {
        atomicAdd(pc, a[0]); atomicAdd(pc + 1, a[1]); atomicAdd(ps, a[2]); atomicAdd(ps + 1, a[3]); atomicAdd(ps + 2, a[4]);
        atomicAdd(pk, a[5]); atomicAdd(pk + 1, a[6]); atomicAdd(pk + 2, a[7]);
    }
}

}  // namespace

extern "C" {

// ---- sampled pixels --------------------------------------------------------------------------------
namespace {
struct PtLayout {
    size_t off_start, off_cursor, off_sorted, off_grads, total;
};
PtLayout make_pt_layout(long n_total)
{
    PtLayout L;
    L.off_start = 0;
    const size_t pts_bytes = align_up((size_t)(n_total > 0 ? n_total : 1) * 16, 256);
    L.off_cursor = align_up((size_t)(PT_CELLS + 2) * 4, 256);
    L.off_sorted = 2 * L.off_cursor;
    L.off_grads = L.off_sorted + pts_bytes;
    L.total = L.off_grads + pts_bytes;
    return L;
}
PtView make_pt_view(const gsasr_dims *d, void *ws, long n_total)
{
    const PtLayout L = make_pt_layout(n_total);
    PtView S;
    S.start = (unsigned *)((char *)ws + L.off_start);
    S.cursor = (unsigned *)((char *)ws + L.off_cursor);
    S.sorted = (float4 *)((char *)ws + L.off_sorted);
    S.grads = (float4 *)((char *)ws + L.off_grads);
    // 8x8-px point-cells while their number fits the sort's LDS table; else coarser ones.  On a batched canvas a
    // cell must not straddle two slots (multiples of 16 rows): it grows in height to 16 rows at most, then in width.
    S.shx = S.shy = PT_MIN_SHIFT;
    const int max_shy = d->batch > 1 ? CELL_SHIFT : PT_MAX_SHIFT;
    for (;;) {
        S.ncx = ((d->w - 1) >> S.shx) + 1;
        S.ncy = ((d->h - 1) >> S.shy) + 1;
        if ((long)S.ncx * S.ncy <= PT_CELLS || (S.shx >= PT_MAX_SHIFT && S.shy >= max_shy)) break;
        if ((S.ncy >= S.ncx || S.shx >= PT_MAX_SHIFT) && S.shy < max_shy) ++S.shy; else ++S.shx;
    }
    return S;
}
int sort_points(const Params &P, const PlanView &V, const PtView &S, const int *points, int n_total, int n_per, hipStream_t st)
{
    HIP_TRY(hipMemsetAsync(S.start, 0, (size_t)(S.ncx * S.ncy + 2) * 4, st));
    const dim3 grid((unsigned)((n_total + 255) / 256)), block(256);
    hipLaunchKernelGGL(k_pts_count, grid, block, 0, st, P, V, S, points, n_total, n_per);
    hipLaunchKernelGGL(k_pts_scan, dim3(1), dim3(1024), 0, st, S, n_total);
    hipLaunchKernelGGL(k_pts_place, grid, block, 0, st, P, V, S, points, n_total, n_per);
    return GSASR_OK;
}
int check_points(const gsasr_dims *dims, int n_points, const void *sample_ws, size_t sample_ws_bytes, long &n_total)
{
    if (dims->row0 != 0 || dims->row1 != dims->h) return fail(GSASR_ERR_ARG, "sampled pixels need the whole image (row0 = 0, row1 = h)");
    if (n_points < 0) return fail(GSASR_ERR_ARG, "n_points < 0");
    n_total = (long)n_points * batch_of(dims);
    if (n_total > 0x7fffffffL) return fail(GSASR_ERR_ARG, "too many points");
    {
        const PtView S = make_pt_view(dims, const_cast<void *>(sample_ws), 0);
        if ((long)S.ncx * S.ncy > PT_CELLS) return fail(GSASR_ERR_ARG, "batched canvas too large for the sampled-pixel path");
    }
    if (!sample_ws || ((uintptr_t)sample_ws & 255u) || sample_ws_bytes < make_pt_layout(n_total).total)
        return fail(GSASR_ERR_WORKSPACE, "sample workspace null, misaligned or smaller than gsasr_sample_workspace_bytes()");
    return GSASR_OK;
}
}  // namespace

size_t gsasr_sample_workspace_bytes(const gsasr_dims *dims, int n_points)
{
    if (!dims_ok(dims) || n_points < 0) {
        fail(GSASR_ERR_ARG, "bad dims");
        return 0;
    }
    return make_pt_layout((long)n_points * batch_of(dims)).total;
}

int gsasr_splat_sample_forward(const gsasr_dims *dims, const void *workspace, size_t workspace_bytes, const int *points,
                               int n_points, float *out, void *sample_ws, size_t sample_ws_bytes, void *stream)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L)) return rc;
    long n_total = 0;
    if (int rc = check_points(dims, n_points, sample_ws, sample_ws_bytes, n_total)) return rc;
    if (n_total == 0) return GSASR_OK;
    if (!points || !out) return fail(GSASR_ERR_ARG, "null pointer");
    const Params P = make_params(dims, L);
    const PlanView V = make_view(L, const_cast<void *>(workspace));
    const PtView S = make_pt_view(dims, sample_ws, n_total);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = sort_points(P, V, S, points, (int)n_total, n_points, st)) return rc;
    // one workgroup per 16x16-px block of point-cells (or per coarser cell) + one for the out-of-range bucket
    const int fsx = S.shx > CELL_SHIFT ? S.shx : CELL_SHIFT, fsy = S.shy > CELL_SHIFT ? S.shy : CELL_SHIFT;
    const dim3 grid((unsigned)((((dims->w - 1) >> fsx) + 1) * (((dims->h - 1) >> fsy) + 1) + 1)), block(64 * SAMPLE_WAVES);
    if (P.bounded) hipLaunchKernelGGL(k_sample_fwd<true>, grid, block, 0, st, P, V, S, n_points, out);
    else hipLaunchKernelGGL(k_sample_fwd<false>, grid, block, 0, st, P, V, S, n_points, out);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_splat_sample_backward(const float *sigmas, const float *coords, const float *colors, const float *grad_out,
                                float *g_sigmas, float *g_coords, float *g_colors, const gsasr_dims *dims,
                                const void *workspace, size_t workspace_bytes, const int *points, int n_points,
                                void *sample_ws, size_t sample_ws_bytes, void *stream)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L)) return rc;
    long n_total = 0;
    if (int rc = check_points(dims, n_points, sample_ws, sample_ws_bytes, n_total)) return rc;
    if (dims->flags & GSASR_FLAG_FORWARD_ONLY) return fail(GSASR_ERR_PLAN, "the plan was made with GSASR_FLAG_FORWARD_ONLY: it holds no backward records");
    if (dims->s == 0) return GSASR_OK;
    if (!g_sigmas || !g_coords || !g_colors) return fail(GSASR_ERR_ARG, "null pointer");
    (void)sigmas; (void)coords; (void)colors;   // (everything the kernel needs is in the plan)
    hipStream_t st = (hipStream_t)stream;
    if (n_total > 0 && !grad_out) return fail(GSASR_ERR_ARG, "null pointer");
    const Params P = make_params(dims, L);
    const PlanView V = make_view(L, const_cast<void *>(workspace));
    const PtView S = make_pt_view(dims, sample_ws, n_total);
    if (n_total == 0) {   // no points: empty point-cells, the kernel below writes (or adds) zeros
        HIP_TRY(hipMemsetAsync(S.start, 0, (size_t)(S.ncx * S.ncy + 2) * 4, st));
    } else {
        if (points)   // NULL: sample_ws still holds the sorted points of the forward call
            if (int rc = sort_points(P, V, S, points, (int)n_total, n_points, st)) return rc;
        hipLaunchKernelGGL(k_pts_grads, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, st, S, grad_out, (int)n_total, n_points);
    }
    const dim3 grid((unsigned)(((size_t)dims->s * SB_LANES + 255) / 256)), block(256);   // SB_LANES lanes per Gaussian
    if (P.bounded) hipLaunchKernelGGL(k_sample_bwd<true>, grid, block, 0, st, P, V, S, g_sigmas, g_coords, g_colors);
    else hipLaunchKernelGGL(k_sample_bwd<false>, grid, block, 0, st, P, V, S, g_sigmas, g_coords, g_colors);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_step_sample_forward(const float *gs_parameters, const float *step_size, const gsasr_dims *dims, void *workspace,
                              size_t workspace_bytes, const int *points, int n_points, float *out, void *sample_ws,
                              size_t sample_ws_bytes, void *stream)
{
    StepLayout S;
    StepSrc SS{};
    SS.step = step_size;
    if (!dims) return fail(GSASR_ERR_ARG, "bad dims");
    // (the sampled kernels walk the cells around their points: a plan made for them carries no tile lists.  All three sampled
    // step entry points lay the workspace out the same way; gsasr_step_workspace_bytes(dims) is never smaller)
    gsasr_dims dn = *dims;
    dn.list_cap = -1;
    if (int rc = step_prologue_plan(gs_parameters, SS, &dn, workspace, workspace_bytes, stream, S)) return rc;
    return gsasr_splat_sample_forward(&dn, workspace, S.plan_bytes, points, n_points, out, sample_ws, sample_ws_bytes, stream);
}

int gsasr_step_sample_forward_sm(const float *gs_parameters, const float *scale_modify, int sm_stride, float default_step_size,
                                 int *mismatch, const gsasr_dims *dims, void *workspace, size_t workspace_bytes, const int *points,
                                 int n_points, float *out, void *sample_ws, size_t sample_ws_bytes, void *stream)
{
    StepLayout S;
    StepSrc SS{};
    SS.sm = scale_modify; SS.stride = sm_stride; SS.def_step = default_step_size; SS.mismatch = mismatch;
    if (!scale_modify && dims && dims->s > 0) return fail(GSASR_ERR_ARG, "null pointer");
    if (!dims) return fail(GSASR_ERR_ARG, "bad dims");
    gsasr_dims dn = *dims;      // (no tile lists for the sampled kernels: gsasr_step_sample_forward)
    dn.list_cap = -1;
    if (int rc = step_prologue_plan(gs_parameters, SS, &dn, workspace, workspace_bytes, stream, S)) return rc;
    return gsasr_splat_sample_forward(&dn, workspace, S.plan_bytes, points, n_points, out, sample_ws, sample_ws_bytes, stream);
}

int gsasr_step_sample_backward(const float *gs_parameters, const float *step_size, const float *grad_out,
                               float *g_parameters, const gsasr_dims *dims, void *workspace, size_t workspace_bytes,
                               const int *points, int n_points, void *sample_ws, size_t sample_ws_bytes, void *stream)
{
    if (!dims_ok(dims)) return fail(GSASR_ERR_ARG, "bad dims");
    if (dims->flags & GSASR_FLAG_STRIDE8) return fail(GSASR_ERR_ARG, "GSASR_FLAG_STRIDE8 does not apply to the step entry points");
    gsasr_dims dn = *dims;      // (the layout the sampled forward planned with: no tile lists)
    dn.list_cap = -1;
    dims = &dn;
    const StepLayout S = make_step_layout(dims, workspace);
    if (!workspace || ((uintptr_t)workspace & 255u) || workspace_bytes < S.total)
        return fail(GSASR_ERR_WORKSPACE, "workspace null, misaligned or smaller than gsasr_step_workspace_bytes()");
    char *b = (char *)workspace;
    float *sig = (float *)(b + S.off_sig), *xy = (float *)(b + S.off_xy), *col = (float *)(b + S.off_col);
    float *gs = (float *)(b + S.off_gsig), *gc = (float *)(b + S.off_gxy), *gk = (float *)(b + S.off_gcol);
    if (!step_size) step_size = (const float *)(b + S.off_step);   // what the forward's prologue used
    gsasr_dims d = *dims;
    d.flags |= GSASR_FLAG_OVERWRITE_GRADS;
    if (int rc = gsasr_splat_sample_backward(sig, xy, col, grad_out, gs, gc, gk, &d, workspace, S.plan_bytes, points, n_points,
                                             sample_ws, sample_ws_bytes, stream))
        return rc;
    if (dims->s == 0) return GSASR_OK;
    if (!gs_parameters || !step_size || !g_parameters) return fail(GSASR_ERR_ARG, "null pointer");
    if (dims->batch > 1) {
        return prologue_backward_batched(gs_parameters, step_size, dims, workspace, gs, gc, gk, g_parameters, stream);
    }
    return gsasr_prologue_backward(gs_parameters, step_size, dims->s, dims->h, dims->w, gs, gc, gk, g_parameters, stream);
}

}  // extern "C"
