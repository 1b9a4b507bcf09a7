// splat_plan.hip -- the plan: classify -> [scan] -> bin (counting sort by cell, records, windows, tile lists)
// (one translation unit of libgsasr_splat.so; gsasr_splat.hip has the overview of the whole pipeline)
#include "splat_common.h"

using namespace gsasr_detail;

namespace {

// ---------------------------------------------------------------------------------------------------
// plan kernels
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_batch_geo(BatchSizes S, int batch, int slot, int w, int4 *__restrict__ geo)
{
    const int b = threadIdx.x;
    if (b < batch) geo[b] = make_int4((int)S.h[b], (int)S.w[b], b * slot, b * w);
}

template <bool PROLOGUE>
__global__ __launch_bounds__(256) void k_classify(Params P, const float *__restrict__ sigmas,
                                                  const float *__restrict__ coords, PlanView V,
                                                  const float *__restrict__ raw, StepSrc SS,
                                                  float *__restrict__ o_sig, float *__restrict__ o_xy, float *__restrict__ o_col)
{
    __shared__ unsigned s_rx[4], s_ry[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    // the counters of the NEXT plan on this workspace (the other parity) are zeroed on the side
    for (int k = i; k < P.count_words; k += (int)(gridDim.x * blockDim.x)) V.cell_count_next[k] = 0u;
    // ... and the cursors of THIS plan's tile lists (k_bin, the next kernel but one at most, counts them up)
    for (int k = i; k < P.tl_ntiles; k += (int)(gridDim.x * blockDim.x)) V.tl_cursor[(size_t)k * TL_STRIDE] = 0u;
    if (i == 0) V.hdr[2] = V.hdr[6] = V.hdr[7] = V.hdr[8] = V.hdr[9] = 0u;   // largest cell / block count: raised with atomicMax by k_scan_local / block_count_max
    // pixel-centre tables: the reference's double expression, rounded to float (gs_cuda/gs.cu:27-28)
    if (P.batch <= 1) {
        if (i < P.w) V.px[i] = (float)(2.0 * (double)i / (double)(P.w - 1) - 1.0);
        if (i < P.h) V.py[i] = (float)(2.0 * (double)i / (double)(P.h - 1) - 1.0);
    } else {  // one px table per sample, py over the canvas rows: each sample's own grid (padding continues it)
        if (i < P.w * P.batch) V.px[i] = (float)(2.0 * (double)(i % P.w) / (double)(sample_geo(P, V, i / P.w).w - 1) - 1.0);
        if (i < P.h) V.py[i] = (float)(2.0 * (double)(i % P.slot) / (double)(sample_geo(P, V, i / P.slot).h - 1) - 1.0);
    }
    unsigned rx = 0, ry = 0, key = 0xffffffffu;
    if (i < P.s) {
        const Geo g = sample_geo(P, V, P.batch > 1 ? i / P.nper : 0);
        float sx, sy, x, y;
        if (PROLOGUE) {
            float o[8];
            const int smp = P.batch > 1 ? i / P.nper : 0;
            float step;
            if (SS.sm) {
                const float s0 = SS.sm[(size_t)smp * SS.stride], s1 = SS.sm[(size_t)smp * SS.stride + 1];
                // (`default_step_size / scale_modify[0]` with a tensor on the right is torch's __rtruediv__: reciprocal, then
                // the product -- two roundings, reproduced here so that the step is the reference's float bit for bit)
                step = (1.0f / s0) * SS.def_step;
                if (i == smp * P.nper) {
                    SS.keep[smp] = step;
                    if (!(s0 == s1) && SS.mismatch) { SS.mismatch[0] = 1 + smp; SS.mismatch[1] = (int)__float_as_uint(s0); }
                }
            } else {
                step = SS.step[smp];
                if (i == smp * P.nper) SS.keep[smp] = step;
            }
            prologue_one(raw + (size_t)i * 9, step, g.h, g.w, o);
            o_sig[i * 3 + 0] = o[0]; o_sig[i * 3 + 1] = o[1]; o_sig[i * 3 + 2] = o[2];
            o_xy[i * 2 + 0] = o[3]; o_xy[i * 2 + 1] = o[4];
            o_col[i * 3 + 0] = o[5]; o_col[i * 3 + 1] = o[6]; o_col[i * 3 + 2] = o[7];
            sx = o[0]; sy = o[1]; x = o[3]; y = o[4];
        } else {
            const size_t i3 = (size_t)i * stride3(P), i2 = (size_t)i * stride2(P);
            sx = sigmas[i3 + 0]; sy = sigmas[i3 + 1];
            x = coords[i2 + 0]; y = coords[i2 + 1];
        }
        const Box b = gaussian_box(sx, sy, x, y, P, g, P.kcut);
        if (b.cls == 2) {
            // NDEAD counters instead of one: a row band of a large image sees most of the Gaussians here, and one
            // returning atomic per wave on a single word serialises (203 us for 1 M Gaussians, 7/8 dead)
            key = (unsigned)P.ncells + 1u + (unsigned)((i >> 6) & (NDEAD_NEAR - 1)) + (b.near ? (unsigned)NDEAD_NEAR : 0u);
        } else if (b.cls == 1) {
            key = (unsigned)P.ncells;
        } else {
            const float hx = 0.5f * (float)(g.w - 1), hy = 0.5f * (float)(g.h - 1);
            int cx = (int)fminf(fmaxf(floorf((x + 1.f) * hx), 0.f), (float)(g.w - 1)) >> CELL_SHIFT;
            int cy = ((int)fminf(fmaxf(floorf((y + 1.f) * hy), 0.f), (float)(g.h - 1)) + g.base) >> CELL_SHIFT;
            key = (unsigned)(cy * P.ncx + cx);
            rx = (unsigned)ceilf(b.ex) + 2u;
            ry = (unsigned)ceilf(b.ey) + 2u;
        }
    }
    // Rank of the Gaussian inside its cell, with ONE returning atomic per (wave, distinct key): decoder
    // output is in raster order, so the 64 Gaussians of a wave fall into a handful of cells (often one,
    // at 16 Gaussians per LR pixel) and per-lane atomics on the same word would serialise at ~10 ns each.
    unsigned rank = 0;
    {
        // match-any without atomics: every lane learns the lane-mask of its key's group ...
        unsigned long long mine = 0ull, todo = __ballot(key != 0xffffffffu);
        while (todo) {
            const unsigned k = (unsigned)__builtin_amdgcn_readlane((int)key, __builtin_ctzll(todo));
            const unsigned long long same = __ballot(key == k);
            if (key == k) mine = same;
            todo &= ~same;
        }
        // ... then ALL group leaders issue their returning atomic in one instruction (one round trip)
        if (mine) {
            const int leader = __builtin_ctzll(mine);
            unsigned base = 0;
            if (lane == leader) base = atomicAdd(&V.cell_count[count_index((int)key, P.ncells, P.dead_off)], (unsigned)__builtin_popcountll(mine));
            base = (unsigned)__shfl((int)base, leader);
            rank = base + (unsigned)__builtin_popcountll(mine & ((1ull << lane) - 1ull));
        }
    }
    if (i < P.s) {
        V.key[i] = key;
        V.rank[i] = rank;
    }
    // per-block max half-extent of the normal class -> one atomicMax pair per block on its group's line (32 blocks per line:
    // a single word for all blocks serialises at ~12 ns per atomic; the readers then reduce groups, not blocks)
    rx = wave_max_u32(rx);
    ry = wave_max_u32(ry);
    if (lane == 0) { s_rx[threadIdx.x >> 6] = rx; s_ry[threadIdx.x >> 6] = ry; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned mx = max(max(s_rx[0], s_rx[1]), max(s_rx[2], s_rx[3])), my = max(max(s_ry[0], s_ry[1]), max(s_ry[2], s_ry[3]));
        if (mx | my) {
            atomicMax(&V.blockmax[16 * (blockIdx.x >> 5) + 0], mx);
            atomicMax(&V.blockmax[16 * (blockIdx.x >> 5) + 1], my);
        }
    }
}

// adapt_kcut's second granularity: the largest number of Gaussians binned in one aligned block of 4 x 4 cells, straight
// from the histogram (final once k_classify is done) -- block `b` of the grid's ceil(ncx/4) x ceil(ncy/4), one per thread
// of whatever scan kernel runs anyway (no launch of its own); the caller reduces over its workgroup and issues ONE atomicMax
// (one per wave -- 1 024 of them on one word at config 4 -- serialised for 12 us).
__device__ __forceinline__ unsigned block_count(int ncx, int ncy, int b, const unsigned *__restrict__ count)
{
    const int nbx = (ncx + 3) >> 2, nby = (ncy + 3) >> 2;
    unsigned sum = 0u;
    if (b < nbx * nby) {
        const int bx = b % nbx, by = b / nbx;
        const int x0 = bx * 4, x1 = min(x0 + 4, ncx);
        if ((ncx & 3) == 0) {   // (whole rows of four counts, 16-byte aligned: one load per row)
            for (int r = by * 4; r < min(by * 4 + 4, ncy); ++r) {
                const uint4 c4 = *reinterpret_cast<const uint4 *>(count + (size_t)r * ncx + x0);
                sum += (c4.x + c4.y) + (c4.z + c4.w);
            }
        } else {
            for (int r = by * 4; r < min(by * 4 + 4, ncy); ++r)
                for (int xx = x0; xx < x1; ++xx) sum += count[r * ncx + xx];
        }
    }
    return sum;
}

__global__ __launch_bounds__(1024) void k_scan(Params P, int n, const unsigned *__restrict__ count,
                                               unsigned *__restrict__ start, int nblk,
                                               const unsigned *__restrict__ blockmax, unsigned *__restrict__ hdr)
{
    __shared__ unsigned part[16];
    __shared__ unsigned smax[4][16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    unsigned bm = 0u;
    if (P.adapt_cells4 > 0.f)
        for (int b0 = 0; b0 < ((P.ncx + 3) >> 2) * ((P.ncy + 3) >> 2); b0 += 1024) bm = max(bm, block_count(P.ncx, P.ncy, b0 + t, count));
    // (a) max half-extents over the classify groups -> plan header
    unsigned mx = 0, my = 0;
    for (int k = t; k < nblk; k += 1024) {
        mx = max(mx, blockmax[16 * k + 0]);
        my = max(my, blockmax[16 * k + 1]);
    }
    // (b) exclusive scan of the per-cell counts (+ the largest count of a cell, for adapt_kcut): up to eight consecutive
    // counts per thread, the 1024 partial sums scanned inside the waves with shuffles and across them through LDS
    const int per = (n + 1023) / 1024;
    const int b = t * per, e = min(n, b + per);
    unsigned sum = 0, mc = 0;
    for (int k = b; k < e; ++k) {
        const unsigned c = count[count_index(k, P.ncells, P.dead_off)];
        sum += c;
        if (k < P.ncells) mc = max(mc, c);
    }
    unsigned inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)inc, o);
        if (lane >= o) inc += v;
    }
    mx = wave_max_u32(mx);
    my = wave_max_u32(my);
    mc = wave_max_u32(mc);
    bm = wave_max_u32(bm);
    if (lane == 63) part[wv] = inc;
    if (lane == 0) { smax[0][wv] = mx; smax[1][wv] = my; smax[2][wv] = mc; smax[3][wv] = bm; }
    __syncthreads();
    if (t < 4) {
        unsigned m = 0;
        for (int k = 0; k < 16; ++k) m = max(m, smax[t][k]);
        hdr[t < 3 ? t : 6] = m;
    }
    unsigned run = inc - sum, total = 0;
    for (int k = 0; k < 16; ++k) {
        const unsigned p = part[k];
        run += k < wv ? p : 0u;
        total += p;
    }
    for (int k = b; k < e; ++k) {
        start[k] = run;
        run += count[count_index(k, P.ncells, P.dead_off)];
    }
    if (t == 0) start[n] = total;
}

// Large grids (> 8192 cells): two-pass scan.  Pass 1: every block scans 4096 counts (4 per thread,
// coalesced) and leaves its total in start_tot[b]; block 0 also reduces the max extents.  Pass 2: every
// block adds the totals of the blocks before it (<= a few hundred values) to its 4096 entries.
constexpr int SCAN_CHUNK = 4096;

__global__ __launch_bounds__(1024) void k_scan_local(int ncells, int n, const unsigned *__restrict__ count,
                                                     unsigned *__restrict__ start, unsigned *__restrict__ tot,
                                                     int nblk, const unsigned *__restrict__ blockmax,
                                                     unsigned *__restrict__ hdr, int ncx, int ncy, int want_blocks, int dead_off)
{
    __shared__ unsigned part[16];
    __shared__ unsigned smax[4][16];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    unsigned bm = 0u;
    if (want_blocks)
        for (int b0 = (int)blockIdx.x * 1024; b0 < ((ncx + 3) >> 2) * ((ncy + 3) >> 2); b0 += (int)gridDim.x * 1024)
            bm = max(bm, block_count(ncx, ncy, b0 + t, count));
    unsigned mx = 0, my = 0;
    if (blockIdx.x == 0) {
        for (int k = t; k < nblk; k += 1024) {
            mx = max(mx, blockmax[16 * k + 0]);
            my = max(my, blockmax[16 * k + 1]);
        }
    }
    const int base = blockIdx.x * SCAN_CHUNK + t * 4;
    unsigned c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = base + k < n ? count[count_index(base + k, ncells, dead_off)] : 0u;
    const unsigned sum = c[0] + c[1] + c[2] + c[3];
    unsigned mc = 0;   // largest count of a cell (adapt_kcut)
#pragma unroll
    for (int k = 0; k < 4; ++k) mc = base + k < ncells ? max(mc, c[k]) : mc;
    // scan of the 1024 partial sums: inside the waves with shuffles, across the sixteen waves through LDS (one barrier pair
    // instead of the twenty of a Hillis-Steele loop); the maxima ride along: ONE atomicMax per workgroup and word
    unsigned inc = sum;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)inc, o);
        if (lane >= o) inc += v;
    }
    mc = wave_max_u32(mc);
    bm = wave_max_u32(bm);
    mx = wave_max_u32(mx);
    my = wave_max_u32(my);
    if (lane == 63) part[wv] = inc;
    if (lane == 0) { smax[0][wv] = mx; smax[1][wv] = my; smax[2][wv] = mc; smax[3][wv] = bm; }
    __syncthreads();
    if (t < 4) {
        unsigned m = 0;
        for (int k = 0; k < 16; ++k) m = max(m, smax[t][k]);
        if (t < 2) { if (blockIdx.x == 0) hdr[t] = m; }
        else if (m) atomicMax(&hdr[t == 2 ? 2 : 6], m);
    }
    unsigned run = inc - sum, total = 0;
    for (int k = 0; k < 16; ++k) {
        const unsigned p = part[k];
        run += k < wv ? p : 0u;
        total += p;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) start[base + k] = run;
        run += c[k];
    }
    if (t == 0) tot[blockIdx.x] = total;

}

__global__ __launch_bounds__(1024) void k_scan_fix(Params P, int n, unsigned *__restrict__ start,
                                                   const unsigned *__restrict__ tot, int nchunks,
                                                   const unsigned *__restrict__ count, unsigned *__restrict__ hdr)
{
    __shared__ unsigned s_off;
    const int t = threadIdx.x;
    (void)P; (void)count; (void)hdr;
    if (t < 64) {  // one wave sums the totals of the preceding chunks
        unsigned v = 0;
        for (int k = t; k < (int)blockIdx.x; k += 64) v += tot[k];
        for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o);
        if (t == 0) s_off = v;
    }
    __syncthreads();
    const unsigned off = s_off;
    const int base = blockIdx.x * SCAN_CHUNK + t * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) start[base + k] += off;
    if ((int)blockIdx.x == nchunks - 1 && t == 0) start[n] = off + tot[blockIdx.x];
}

// Tile lists: the quadrants (8 x 8 px; bit 4 qy + qx) of list tile (tx, ty) that Gaussian's ellipse {exponent >= -tau'}
// reaches, from the window words k_bin has just built (bb = bbox[2j], sp = the spans of bands 4..7): exactly the
// window-rectangle + per-16-row-band column-span test the search kernels apply per sub-tile (fwd_block phase B), refined
// to quadrant rows by the window's own first and last row.
// The same from the per-8-row spans (`qs`: k_bin's qspan, present in plans whose backward is the tile-stationary kernel and for
// windows of at most eight such bands): the cull that kernel's own level 1 applies.
template <int HLOG>
__device__ __forceinline__ unsigned tl_mask8(int tx, int ty, const uint4 bb, const uint4 qs, int row0)
{
    const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16), r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
    const int q0 = (r0 - row0) >> 3, cu0 = c0 >> 3;
    constexpr int NQY = 1 << (HLOG - 3);     // quadrant rows per tile
    unsigned mask = 0u;
#pragma unroll
    for (int qy = 0; qy < NQY; ++qy) {
        const int G = ty * NQY + qy, y0 = row0 + (G << 3);
        if (!(r0 <= y0 + 7 && r1 >= y0)) continue;
        const unsigned t = (unsigned)(G - q0) & 7u, sh = (t & 3u) * 8u;
        const unsigned l = ((t < 4u ? qs.x : qs.z) >> sh) & 0xffu, h = ((t < 4u ? qs.y : qs.w) >> sh) & 0xffu;
        if (l > h) continue;
        const int lo = cu0 + (int)l, hi = h == 255u ? (c1 >> 3) : cu0 + (int)h;      // (255 = as far as the window goes)
        const int a = max(lo - 4 * tx, 0), b = min(hi - 4 * tx, 3);
        if (a <= b) mask |= ((2u << b) - (1u << a)) << (4 * qy);
    }
    return mask;
}

template <int HLOG>
__device__ __forceinline__ unsigned tl_mask(int tx, int ty, const uint4 bb, const uint2 sp, int row0)
{
    const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16), r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
    const bool spans = (bb.y & 0x8000u) != 0u;
    const int wb0 = (r0 - row0) >> SUBY_SHIFT, cu0 = c0 >> SUBX_SHIFT;
    constexpr int NB = 1 << (HLOG - 4);      // 16-row bands per tile
    unsigned mask = 0u;
#pragma unroll
    for (int bnd = 0; bnd < NB; ++bnd) {
        const int G = ty * NB + bnd;         // the band, counted from row0
        const int y0 = row0 + (G << SUBY_SHIFT);
        int lo = cu0, hi = c1 >> SUBX_SHIFT;
        bool any = r0 <= y0 + SUBY - 1 && r1 >= y0;
        if (spans) {
            const unsigned t = (unsigned)(G - wb0) & 7u, sh = (t & 3u) * 8u;
            const unsigned l = ((t < 4u ? bb.z : sp.x) >> sh) & 0xffu, h = ((t < 4u ? bb.w : sp.y) >> sh) & 0xffu;
            lo = cu0 + (int)l;
            hi = cu0 + (int)h;
            any = any && l <= h;
        }
        const int q0 = max(lo - 4 * tx, 0), q1 = min(hi - 4 * tx, 3);
        if (any && q0 <= q1) {
            const unsigned bits = (2u << q1) - (1u << q0);
            if (r0 <= y0 + 7) mask |= bits << (8 * bnd);
            if (r1 >= y0 + 8) mask |= bits << (8 * bnd + 4);
        }
    }
    return mask;
}

// Append {j | test << 31, mask} to the lists of the tiles Gaussian j's window touches, TLB tiles of every lane per round.
// The cursors are bumped with ONE returning atomic per (wave, round slot, distinct tile), all of a slot's issued in one
// instruction (cf. k_classify's ranks): raster-ordered decoder output puts the 64 Gaussians of a wave into a handful of tiles,
// and atomics on one word -- on one cache LINE -- serialise at ~12 ns each whichever wave they come from.
// (Measured and dropped, profiles/history/r05_lists_ab.txt run r05d: per-lane atomics without the match-any loops, six tiles per round --
// config 2's k_bin +10.5 us instead of +6, config 4's plan +165 us instead of +81: the atomics, not the loops, are what costs.)
constexpr int TLB = 4;

// Plans whose backward is the tile-stationary kernel on the same tiles (P.tl_hlog == P.bt_hlog, slots in use): the entry also carries
// the Gaussian's slot in part[] for that tile (bits 16..23; BT_WIDE = more tiles than slots: atomics), and a tile of the window's
// rectangle that the ellipse misses -- no entry -- has its slot zeroed here, because the gather adds every slot of the rectangle.
template <int HLOG>
__device__ __forceinline__ void tl_emit(const Params &P, const PlanView &V, bool emit, unsigned j, const uint4 bb, const uint2 sp,
                                        const uint4 qs, bool fine8)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    int nx = 0, ntile = 0, tX0 = 0, tY0 = 0;
    if (emit) {
        const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16), r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
        tX0 = c0 >> 5;
        nx = (c1 >> 5) - tX0 + 1;
        tY0 = (r0 - P.row0) >> HLOG;
        ntile = nx * (((r1 - P.row0) >> HLOG) - tY0 + 1);
    }
    const unsigned ex = j | ((bb.x & 0x8000u) << 16);
    const bool slots = P.part_k > 0 && P.tl_hlog == P.bt_hlog;
    const bool wide = !slots || ntile > P.part_k;      // (bt_tile_span counts the same rectangle of the same tiles)
    int ix = 0, iy = 0;
    for (int base = 0; __ballot(base < ntile) != 0ull; base += TLB) {
        unsigned m[TLB], ti[TLB], pos[TLB];
        unsigned long long mine[TLB];
#pragma unroll
        for (int k = 0; k < TLB; ++k) {
            const bool v = base + k < ntile;
            m[k] = !v ? 0u : fine8 ? tl_mask8<HLOG>(tX0 + ix, tY0 + iy, bb, qs, P.row0) : tl_mask<HLOG>(tX0 + ix, tY0 + iy, bb, sp, P.row0);
            ti[k] = (unsigned)((tY0 + iy) * P.tl_ntx + tX0 + ix);
            if (v) {
                const unsigned slot = wide ? 0xffu : (unsigned)(base + k);      // tile `base + k` of the window, row-major: bt_tile_span's order
                if (m[k]) m[k] |= slot << 16;
                else if (!wide) {      // the ellipse misses this tile of its window: the gather still adds the slot
                    float4 *o = reinterpret_cast<float4 *>(V.part + ((size_t)j * P.part_k + slot) * 8);
                    o[0] = make_float4(0.f, 0.f, 0.f, 0.f);
                    o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (v && ++ix == nx) { ix = 0; ++iy; }
            // lanes with the same tile in this slot: one group, one atomic
            mine[k] = 0ull;
            unsigned long long todo = __ballot(m[k] != 0u);
            while (todo) {
                const unsigned t0 = (unsigned)__builtin_amdgcn_readlane((int)ti[k], __builtin_ctzll(todo));
                const unsigned long long same = __ballot(m[k] != 0u && ti[k] == t0);
                if (m[k] != 0u && ti[k] == t0) mine[k] = same;
                todo &= ~same;
            }
        }
#pragma unroll
        for (int k = 0; k < TLB; ++k) {
            pos[k] = 0u;
            if (mine[k] && lane == __builtin_ctzll(mine[k]))
                pos[k] = atomicAdd(&V.tl_cursor[(size_t)ti[k] * TL_STRIDE], (unsigned)__builtin_popcountll(mine[k]));
        }
#pragma unroll
        for (int k = 0; k < TLB; ++k) {
            const int leader = mine[k] ? __builtin_ctzll(mine[k]) : 0;
            const unsigned at = (unsigned)__shfl((int)pos[k], leader) + (unsigned)__builtin_popcountll(mine[k] & below);
            if (mine[k] && at < (unsigned)P.tl_cap) V.tl_entries[(size_t)ti[k] * (size_t)P.tl_cap + at] = make_uint2(ex, m[k]);
        }
    }
}

// counting-sort placement (slot = cell start + rank, no atomics) fused with record packing
// FUSED_SCAN (grids of at most FUSED_CELLS cells+2, e.g. 1024^2): every block rebuilds the exclusive scan of
// the cell histogram in LDS itself (16 counts per thread) instead of waiting for a separate one-block scan
// kernel -- one launch less on a latency-bound plan; block 0 publishes cell_start[] and the header.
constexpr int FUSED_PER_THREAD = 17, FUSED_CELLS = 256 * FUSED_PER_THREAD;
static_assert(FUSED_CELLS == FUSED_CELLS_HOST, "make_params decides with FUSED_CELLS_HOST which plans run a scan kernel");

// TLH: the plan's tile lists -- 0 none, else log2 of the tile height (4 / 5)
template <bool FUSED_SCAN, int TLH>
__global__ __launch_bounds__(256) void k_bin(Params P, const float *__restrict__ sigmas,
                                             const float *__restrict__ coords,
                                             const float *__restrict__ colors, PlanView V, int nblk)
{
    __shared__ unsigned s_start[FUSED_SCAN ? FUSED_CELLS + 1 : 1];
    __shared__ unsigned s_part[FUSED_SCAN ? 256 : 1];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // Everything that does not depend on the Gaussian's slot j is done FIRST (its loads are issued together with
    // the counter loads of the scan below): this kernel runs one wave per SIMD, so its run time is the length
    // of its dependent chain, not its instruction count.
    const bool valid = i < P.s;
    unsigned c[FUSED_SCAN ? FUSED_PER_THREAD : 1];   // this thread's share of the per-cell counters (scan below)
    if (FUSED_SCAN) {
        // (measured in round 5 and dropped: requesting them coalesced -- counter k * 256 + t by thread t -- and transposing
        // through LDS costs a barrier more than the 60 cache lines per load it saves: k_bin 9.97 us against 9.4-9.6)
#pragma unroll
        for (int k = 0; k < FUSED_PER_THREAD; ++k) {
            const int q = (int)threadIdx.x * FUSED_PER_THREAD + k;
            c[k] = q < P.ncells + 1 + NDEAD ? V.cell_count[count_index(q, P.ncells, P.dead_off)] : 0u;
        }
    }
    const bool adapting = P.adapt_cells > 0.f || P.adapt_ring != 0;
    const unsigned nlarge = FUSED_SCAN && adapting ? V.cell_count[P.ncells] : 0u;
    unsigned key = 0u, rnk = 0u;
    float4 recA = make_float4(0.f, 0.f, 0.f, 0.f), recB = recA, finA = recA, finB = recA;
    uint4 bb = make_uint4(0u, 0u, 0u, 0u), bc = bb;
    uint4 qs = make_uint4(0u, 0xffffffffu, 0u, 0xffffffffu);   // quadrant-row spans: every column unless computed below
    bool fine8 = false;                                        // ... computed: the tile lists' masks are cut from them
    bool large = false;
    unsigned fb_rx = 0u, fb_ry = 0u;
    float sx = 0.f, sy = 0.f, rho = 0.f, x = 0.f, y = 0.f, col0 = 0.f, col1 = 0.f, col2 = 0.f;
    if (valid) {
        key = V.key[i];
        rnk = V.rank[i];
        const size_t i3 = (size_t)i * stride3(P), i2 = (size_t)i * stride2(P);
        sx = sigmas[i3 + 0]; sy = sigmas[i3 + 1]; rho = sigmas[i3 + 2];
        x = coords[i2 + 0]; y = coords[i2 + 1];
        col0 = colors[i3 + 0]; col1 = colors[i3 + 1]; col2 = colors[i3 + 2];
    }
    // The cutoff the windows are built with (adapt_kcut): from the largest cell count -- every block reduces the histogram it
    // holds anyway (FUSED_SCAN), or reads what the scan kernels left in the header.
    float kc = P.kcut, kc_tau = 0.f;
    unsigned kc_K = 0u, kc_mc = 0u, kc_nn = 0u;
    if constexpr (FUSED_SCAN) {
        if (adapting) {
            // block-wide: the largest cell count, the near-dead count (both from the histogram this block holds anyway) and the
            // class' largest extent (every block reduces k_classify's per-block maxima: block 0 alone publishes the header
            // further down, too late for the windows)
            unsigned mc = 0u, nn = 0u, ext = 0u, eyt = 0u;
#pragma unroll
            for (int k = 0; k < FUSED_PER_THREAD; ++k) {
                const int q = (int)threadIdx.x * FUSED_PER_THREAD + k;
                mc = q < P.ncells ? max(mc, c[k]) : mc;
                nn += q >= P.ncells + 1 + NDEAD_NEAR ? c[k] : 0u;      // (c[k] = 0 past the last class)
            }
            if (P.adapt_ring) {
                for (int k = (int)threadIdx.x; k < nblk; k += 256) {
                    ext = max(ext, V.blockmax[16 * k]);
                    eyt = max(eyt, V.blockmax[16 * k + 1]);
                }
            }
            mc = wave_max_u32(mc);
            ext = wave_max_u32(ext);
            eyt = wave_max_u32(eyt);
            nn = wave_add_u32(nn);
            if ((threadIdx.x & 63) == 0) {
                s_part[threadIdx.x >> 6] = mc;
                s_part[4 + (threadIdx.x >> 6)] = nn;
                s_part[8 + (threadIdx.x >> 6)] = ext;
                s_part[12 + (threadIdx.x >> 6)] = eyt;
            }
            __syncthreads();
            mc = max(max(s_part[0], s_part[1]), max(s_part[2], s_part[3]));
            nn = (s_part[4] + s_part[5]) + (s_part[6] + s_part[7]);
            ext = max(max(s_part[8], s_part[9]), max(s_part[10], s_part[11]));
            eyt = max(max(s_part[12], s_part[13]), max(s_part[14], s_part[15]));
            __syncthreads();   // (s_part is reused by the scan below)
            kc = adapt_kcut(P, mc, nlarge, nn, ext, eyt, kc_tau, kc_K);
            kc_mc = mc;
            kc_nn = nn;
        } else {
            kc_tau = 0.5f * P.kcut * P.kcut;
        }
    } else {   // (the scan kernels left the maxima in the header and the finished scan)
        const int ncls = P.ncells + 1 + NDEAD;
        kc = adapt_kcut(P, V.hdr[2], V.cell_start[P.ncells + 1] - V.cell_start[P.ncells],
                        V.cell_start[ncls] - V.cell_start[P.ncells + 1 + NDEAD_NEAR], V.hdr[0], V.hdr[1], kc_tau, kc_K, V.hdr[6]);
        if (i == 0) {
            V.hdr[3] = __float_as_uint(kc);
            V.hdr[4] = __float_as_uint(kc_tau);
            V.hdr[5] = kc_K;
            V.hdr[7] = V.cell_start[ncls] - V.cell_start[P.ncells + 1 + NDEAD_NEAR];
            atomicMax(&V.hdr[8], reach_of(V.hdr[0], kc, P.kcut, P.cap_px_x));
            atomicMax(&V.hdr[9], reach_of(V.hdr[1], kc, P.kcut, P.cap_px_y));
        }
    }
    if (valid) {
        const int smp = P.batch > 1 ? i / P.nper : 0;
        const Geo g = sample_geo(P, V, smp);
        Box b = gaussian_box(sx, sy, x, y, P, g, kc);
        // A Gaussian k_classify kept (with the conservative cutoff) whose window under the smaller cutoff holds no pixel keeps
        // its conservative window: every consumer finds a non-empty window behind a live key, and the classes' extents
        // (header words 0, 1: the conservative ones) cover it.
        float kw = kc;
        if (b.cls == 2 && key <= (unsigned)P.ncells && kc != P.kcut) {
            b = gaussian_box(sx, sy, x, y, P, g, P.kcut);
            kw = P.kcut;
            if (key < (unsigned)P.ncells) {   // (normal class: the tiles must search as far as this conservative window reaches)
                fb_rx = (unsigned)ceilf(b.ex) + 2u;
                fb_ry = (unsigned)ceilf(b.ey) + 2u;
            }
        }
        large = key == (unsigned)P.ncells;
        // exponent = w1*(dx^2/sx^2 - 2 rho dx dy/(sx sy) + dy^2/sy^2), w1 = -0.5/(1-rho^2)   (gs.cu:33-56);
        // everything per-Gaussian is evaluated ONCE here, in double, and rounded to float
        const double dr = rho, dsx = sx, dsy = sy;
        const double w1 = -0.5 / (1.0 - dr * dr);
        const double w2 = 1.0 / (dsx * dsx), w3 = 1.0 / (dsx * dsy), w4 = 1.0 / (dsy * dsy);
        // The forward evaluates the completed square (like the backward, bwd_trip): with u0 = dx/sx, v0 = dy/sy,
        //   dx^2/sx^2 - 2 rho dx dy/(sx sy) + dy^2/sy^2 = (1-rho^2) u0^2 + (v0 - rho u0)^2,
        // so log2(e) * exponent = -U^2 - Bq^2,  U = sqrt(h) dx/sx,  Bq = sqrt(h c) dy/sy - rho sqrt(c) U,  h = log2(e)/2,
        // c = 1/(1-rho^2).  Same seven instructions per record and lane as the monomial form A dx^2 + B dx dy + C dy^2, but
        // nothing cancels as |rho| -> 1: there the monomial form (the reference's own, gs.cu:33-56) subtracts terms of size
        // u0^2 c from each other in fp32 -- at rho = 0.999999 an image value was off by 0.3% of itself (tools/fuzz_step.py).
        const double cinv_d = -2.0 * w1, hl = 0.5 * LOG2E;
        const float IX = (float)(sqrt(hl) / dsx);
        const float IY = (float)(sqrt(hl * cinv_d) / dsy);
        const float NR = (float)(-dr * sqrt(cinv_d));
        (void)w2; (void)w3; (void)w4;
        // record layout {x, y, IX, NR | IY, r, g, b}: after the two 16-byte LDS reads of the forward every value it
        // broadcasts into a packed-fp32 operand (y, IY, r, g, b) is the low or high half of an aligned register pair
        recA = make_float4(x, y, IX, NR);
        recB = make_float4(IY, col0, col1, col2);
        // constants of the backward epilogue (gs.cu:112-117) + the Gaussian's original index
        // backward constants: c = 1/(1-rho^2) = -2 w1, kappa = 1-rho^2 (formed in double: no cancellation), rho, 1/sigma
        finA = make_float4((float)(-2.0 * w1), (float)(1.0 - dr * dr), rho, (float)(1.0 / dsx));
        // + where the sample's px table starts and which slot it is (0, 0 for a single image)
        finB = make_float4((float)(1.0 / dsy), __uint_as_float((unsigned)g.pxo), __uint_as_float((unsigned)smp),
                           __uint_as_float((unsigned)i));
        // does a pixel of a tile this Gaussian is binned to ever need the dmax test?  Not if its support box
        // lies inside its dmax box: pixels beyond the support box carry < exp(-tau) whether tested or not.
        const float hx = 0.5f * (float)(g.w - 1), hy = 0.5f * (float)(g.h - 1);
        const bool needs_test = P.bounded && !(kw > 0.f && kw * fabsf(sx) * hx + 1.f <= P.dmax * hx &&
                                               kw * fabsf(sy) * hy + 1.f <= P.dmax * hy);
        if (b.cls == 2) {
            bb = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);  // c0 = r0 = 32767 > c1 = r1 = 0: overlaps no tile
        } else {
            bb.x = (unsigned)b.c0 | (needs_test ? 0x8000u : 0u) | ((unsigned)b.c1 << 16);
            bb.y = (unsigned)b.r0 | ((unsigned)b.r1 << 16);
            bb.z = bb.w = 0u;
            // Row spans: for each 16-row band of forward tiles the window touches (at most 8 are encoded),
            // the range of 8-px tile columns that the ellipse {exponent >= -tau} actually reaches.  The
            // window's corners are empty for every Gaussian (and most of it for a correlated one), so this
            // removes ~30% of the forward's (tile, Gaussian) visits that the rectangular window admits.
            const int ty0 = (b.r0 - P.row0) >> SUBY_SHIFT, ty1 = (b.r1 - P.row0) >> SUBY_SHIFT;
            if (kw > 0.f && ty1 - ty0 < 8 && (b.c1 >> SUBX_SHIFT) - (b.c0 >> SUBX_SHIFT) <= 255) {
                // (fp32 relative to the centre: the plan runs one wave per SIMD, so the length of this dependent
                // chain is k_bin's run time; an ulp of a <= 128 px offset is far inside WINDOW_EPS.  Only the absolute
                // pixel coordinates stay in double.)
                const float spx = sx * hx, spy = sy * hy;                     // sigmas in pixels
                const double cxp = ((double)x + 1.0) * (double)hx, cyp = ((double)y + 1.0) * (double)hy + (double)g.base;
                const float tau = 0.5f * kw * kw;
                const float omr = (float)(1.0 - dr * dr);
                const float iq = 1.f / (omr * spx * spy);
                const float qa = 0.5f * iq * (spy / spx), qb = -rho * iq, qc = 0.5f * iq * (spx / spy);
                const float umax = fabsf(spx) * kw, vmax = fabsf(spy) * kw;
                const float vstar = -qb * umax / (2.f * qc);                  // v of the ellipse's rightmost point (= rho spy k)
                const float disc0 = 4.f * qa * tau, disc2 = 4.f * qa * qc - qb * qb, i2qa = 0.5f / qa;
                const float eps = (float)WINDOW_EPS;
                const int tx0 = b.c0 >> SUBX_SHIFT;
                // eight bands of (1 << shift) rows starting at band `first` (counted from row0)
                // (the loop runs as far as the tallest window of the WAVE reaches -- two or three bands at GSASR's x4, not
                // eight: the plan runs one wave per SIMD, every predicated iteration is on its critical path)
                auto spans = [&](int shift, int first, int nb, unsigned (&lo4)[2], unsigned (&hi4)[2]) {
                    lo4[0] = lo4[1] = 0x01010101u;   // every band empty (lo = 1 > hi = 0) until computed
                    hi4[0] = hi4[1] = 0u;
                    for (int t = 0; t < 8 && __ballot(t < nb) != 0ull; ++t) {
                        if (t >= nb) continue;
                        unsigned lo = 1u, hi = 0u;  // empty
                        // the band's pixel rows Ya..Ya+2^shift-1, relative to the centre
                        const float v0 = (float)((double)(P.row0 + ((first + t) << shift)) - cyp) - eps,
                                    v1 = v0 + (float)((1 << shift) - 1) + 2.f * eps;
                        if (v1 >= -vmax && v0 <= vmax) {
                            const float a0 = fmaxf(v0, -vmax), a1 = fminf(v1, vmax);
                            const float vr = fminf(fmaxf(vstar, a0), a1), vl = fminf(fmaxf(-vstar, a0), a1);
                            const float dr_ = disc0 - disc2 * vr * vr;
                            const float dl_ = disc0 - disc2 * vl * vl;
                            const float uhi = (-qb * vr + sqrtf(fmaxf(dr_, 0.f))) * i2qa;
                            const float ulo = (-qb * vl - sqrtf(fmaxf(dl_, 0.f))) * i2qa;
                            const int xl = max(b.c0, (int)fmax(ceil(cxp + (double)(ulo - eps)), -1.0));
                            const int xh = min(b.c1, (int)fmin(floor(cxp + (double)(uhi + eps)), 40000.0));
                            if (xl <= xh && !(umax != umax)) {
                                lo = (unsigned)min(255, (xl >> SUBX_SHIFT) - tx0);
                                hi = (unsigned)min(255, (xh >> SUBX_SHIFT) - tx0);
                            }
                        }
                        const unsigned sh = 8u * (unsigned)(t & 3), keep = ~(0xffu << sh);
                        if (t < 4) { lo4[0] = (lo4[0] & keep) | (lo << sh); hi4[0] = (hi4[0] & keep) | (hi << sh); }
                        else { lo4[1] = (lo4[1] & keep) | (lo << sh); hi4[1] = (hi4[1] & keep) | (hi << sh); }
                    }
                };
                unsigned lo4[2], hi4[2];
                spans(SUBY_SHIFT, ty0, ty1 - ty0 + 1, lo4, hi4);
                // the same per band of 8 rows, for the 8x8-px quadrants of the tile-stationary backward
                const int q0 = (b.r0 - P.row0) >> 3, q1 = (b.r1 - P.row0) >> 3;
                if (V.qspan && q1 - q0 < 8) {   // (plans with slots only: the others never run the tile-stationary backward)
                    unsigned l8[2], h8[2];
                    spans(3, q0, q1 - q0 + 1, l8, h8);
                    qs = make_uint4(l8[0], h8[0], l8[1], h8[1]);
                    fine8 = true;
                }
                bb.z = lo4[0]; bb.w = hi4[0];
                bc.x = lo4[1]; bc.y = hi4[1];
                bb.y |= 0x8000u;
            }
            {   // The window the Gaussian-stationary backward sweeps (round 6): that of min(tau', GSASR_SPLAT_GRAD_TAU) -- a
                // Gaussian's gradient sums over its own pixels only, so its window does not grow with the K of the forward's
                // bound (include/gsasr_splat.h).  A Gaussian whose smaller window holds no pixel keeps the forward's.
                Box w = b;
                bool test_b = needs_test;
                if (P.kb_max > 0.f && kw > P.kb_max) {
                    const Box t = gaussian_box(sx, sy, x, y, P, g, P.kb_max);
                    if (t.cls != 2) {
                        w = t;
                        test_b = P.bounded && !(P.kb_max * fabsf(sx) * hx + 1.f <= P.dmax * hx && P.kb_max * fabsf(sy) * hy + 1.f <= P.dmax * hy);
                    }
                }
                // Its rows rounded up to a whole number of trips
                // (8/4/2 rows for 16/32/64-lane columns) when the band has room -- the extra rows lie outside the window
                // (their terms are < exp(-tau), or fail the dmax test), and the ragged, masked last trip disappears.
                // Batched canvas: inside the sample's own rows (whatever gradient the caller left in the padding of the
                // slot must not be read).  Worked out here, once, instead of by every backward wave on its scalar unit.
                const int bwid = w.c1 - w.c0 + 1, nr = w.r1 - w.r0 + 1;
                const int rpt = bwid <= 16 ? 8 : (bwid <= BWD_LX21_MAX ? 6 : (bwid <= 32 ? 4 : 2));      // (bwd_sweep's rows per trip)
                const int pad = (rpt - nr % rpt) % rpt;
                const int lo = max(P.row0, g.base), hi = min(P.row1, g.base + g.h) - 1;
                int r0p = w.r0, r1p = w.r1;
                if (r1p + pad <= hi) r1p += pad;
                else if (r0p - pad >= lo) r0p -= pad;
                bc.z = (unsigned)r0p | ((unsigned)r1p << 16);
                bc.w = (unsigned)w.c0 | (test_b ? 0x8000u : 0u) | ((unsigned)w.c1 << 16);      // its columns | "a pixel may need the dmax test"
            }
        }
    }
    {   // Gaussians that kept their conservative window raise the reach (rare: one atomic pair per wave that holds any)
        if (__ballot(fb_rx != 0u) != 0ull) {
            const unsigned wx = wave_max_u32(fb_rx), wy = wave_max_u32(fb_ry);
            if ((threadIdx.x & 63) == 0) {
                atomicMax(&V.hdr[8], wx);
                atomicMax(&V.hdr[9], wy);
            }
        }
    }
    if constexpr (FUSED_SCAN) {
        const int t = threadIdx.x, ncls = P.ncells + 1 + NDEAD;
        const int b0 = t * FUSED_PER_THREAD;
        unsigned sum = 0;
#pragma unroll
        for (int k = 0; k < FUSED_PER_THREAD; ++k) sum += c[k];
        // block scan of the 256 partial sums: inside the waves with shuffles, across the four waves through LDS
        // (one barrier instead of the sixteen of a Hillis-Steele loop over s_part)
        unsigned inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = (unsigned)__shfl_up((int)inc, o);
            if ((t & 63) >= o) inc += v;
        }
        if ((t & 63) == 63) s_part[t >> 6] = inc;
        __syncthreads();
        unsigned run = inc - sum;
        for (int k = 0; k < (t >> 6); ++k) run += s_part[k];
#pragma unroll
        for (int k = 0; k < FUSED_PER_THREAD; ++k) {
            if (b0 + k <= ncls) s_start[b0 + k] = run;
            run += c[k];
        }
        __syncthreads();
        if (blockIdx.x == 0) {  // publish for the render kernels
            for (int k = t; k <= ncls; k += 256) V.cell_start[k] = s_start[k];
            unsigned mx = 0, my = 0;
            for (int k = t; k < nblk; k += 256) {
                mx = max(mx, V.blockmax[16 * k + 0]);
                my = max(my, V.blockmax[16 * k + 1]);
            }
            mx = wave_max_u32(mx);
            my = wave_max_u32(my);
            __syncthreads();
            if ((t & 63) == 0) { s_part[t >> 6] = mx; s_part[4 + (t >> 6)] = my; }
            __syncthreads();
            if (t == 0) {
                const unsigned ex0 = max(max(s_part[0], s_part[1]), max(s_part[2], s_part[3]));
                const unsigned ey0 = max(max(s_part[4], s_part[5]), max(s_part[6], s_part[7]));
                V.hdr[0] = ex0;
                V.hdr[1] = ey0;
                atomicMax(&V.hdr[8], reach_of(ex0, kc, P.kcut, P.cap_px_x));
                atomicMax(&V.hdr[9], reach_of(ey0, kc, P.kcut, P.cap_px_y));
                V.hdr[2] = kc_mc;
                V.hdr[3] = __float_as_uint(kc);
                V.hdr[4] = __float_as_uint(kc_tau);
                V.hdr[5] = kc_K;
                V.hdr[7] = kc_nn;
            }
        }
    }
    if (!valid && TLH == 0) return;
    const unsigned j = valid ? (FUSED_SCAN ? s_start[key] : V.cell_start[key]) + rnk : 0u;
    if (valid) {
    // A dead Gaussian (off the image, off this row band, non-finite) is never a candidate of any tile; all that is ever read
    // of it is its (empty) window and its original index, by the backward that writes its zero gradient.  A row band of
    // a sharded image plans every Gaussian of the image: most of them are dead there, and their records are not written.
    const bool live = key <= (unsigned)P.ncells;
    const bool backward_records = !(P.flags & GSASR_FLAG_FORWARD_ONLY);
    if (live) {
        V.rec[2 * j + 0] = recA;
        V.rec[2 * j + 1] = recB;
        V.bbox[2 * j + 1] = bc;
        if (V.qspan) V.qspan[j] = qs;
    }
    if (backward_records) {
        if (live) V.fin[2 * j + 0] = finA;
        V.fin[2 * j + 1] = finB;
    }
    V.bbox[2 * j] = bb;
    V.win[j] = make_uint2(bb.x, bb.y);
    // the atomic accumulators (row chunks of a large Gaussian; windows wider than their slots in the tile backward) start from zero
    // (needed by: the large class; a plan with slots, whose too-wide windows fall back to them; the atomic variant)
    if (backward_records && live && (large || V.qspan || (P.flags & GSASR_FLAG_BWD_ATOMIC))) {
        reinterpret_cast<float4 *>(V.sums)[2 * (size_t)j] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4 *>(V.sums)[2 * (size_t)j + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (large) V.done[j] = 0u;
    }
    }
    if constexpr (TLH != 0)   // tile lists: the normal class only (the large one stays a segment every tile scans)
        tl_emit<TLH == 0 ? 4 : TLH>(P, V, valid && key < (unsigned)P.ncells, j, bb, make_uint2(bc.x, bc.y), qs, fine8);
}

}  // namespace

namespace gsasr_detail {

// batched canvas: publish the per-sample geometry (host array in dims) to the workspace
int launch_batch_geo(const gsasr_dims *dims, const PlanView &V, hipStream_t st)
{
    if (dims->batch <= 1) return GSASR_OK;
    int uh, uw;
    if (batch_uniform(dims, uh, uw)) return GSASR_OK;   // (one size for all samples: the kernels get it as an argument, nothing reads PlanView::geo)
    BatchSizes S;
    for (int b = 0; b < GSASR_MAX_BATCH; ++b) {
        S.h[b] = (unsigned short)(b < dims->batch ? dims->sample_hw[2 * b] : 0);
        S.w[b] = (unsigned short)(b < dims->batch ? dims->sample_hw[2 * b + 1] : 0);
    }
    hipLaunchKernelGGL(k_batch_geo, dim3(1), dim3(64), 0, st, S, dims->batch, dims->slot, dims->w, V.geo);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

// The plan: [memset of this parity's counters unless the caller vouches for them] -> classify (with the host prologue
// fused in when `raw` is given: sigmas/coords/colors are then OUTPUTS) -> [scan] -> bin.
int plan_impl(const float *sigmas, const float *coords, const float *colors, const gsasr_dims *dims, void *workspace,
              size_t workspace_bytes, void *stream, const float *raw, const StepSrc &SS)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L, true)) return rc;
    note_plan(workspace, dims, L.part_k, L.tl_hlog, L.tl_cap);
    if (dims->s > 0 && (!sigmas || !coords || !colors)) return fail(GSASR_ERR_ARG, "null input pointer");
    hipStream_t st = (hipStream_t)stream;
    const Params P = make_params(dims, L);
    const PlanView V = make_view(L, workspace, dims->flags);
    if (!(dims->flags & GSASR_FLAG_COUNTERS_CLEAN)) HIP_TRY(hipMemsetAsync(V.cell_count, 0, L.count_bytes, st));
    if (!raw)
        if (int rc = launch_batch_geo(dims, V, st)) return rc;   // (a step call has published the geometry already)
    const int nblk = classify_blocks(dims);
    if (raw)
        hipLaunchKernelGGL(k_classify<true>, dim3(nblk), dim3(256), 0, st, P, sigmas, coords, V, raw, SS,
                           const_cast<float *>(sigmas), const_cast<float *>(coords), const_cast<float *>(colors));
    else
        hipLaunchKernelGGL(k_classify<false>, dim3(nblk), dim3(256), 0, st, P, sigmas, coords, V, (const float *)nullptr,
                           SS, (float *)nullptr, (float *)nullptr, (float *)nullptr);
    const int ncls = L.ncells + 1 + NDEAD;
    const unsigned nbin = (unsigned)((dims->s + 255) / 256);
    static const int fused_max_blocks = dev_switch("GSASR_SPLAT_FUSED_MAX") ? atoi(dev_switch("GSASR_SPLAT_FUSED_MAX")) : FUSED_MAX_BLOCKS;
    if (ncls <= FUSED_CELLS && dims->s > 0 && (int)nbin <= fused_max_blocks) {
        // small grid, not too many blocks: k_bin rebuilds the scan per block (no separate scan launch)
#define GSASR_BIN(F) do { \
        const int tlh = P.tl_hlog; \
        if (tlh == 5) hipLaunchKernelGGL((k_bin<F, 5>), dim3(nbin), dim3(256), 0, st, P, sigmas, coords, colors, V, L.ext_groups); \
        else if (tlh == 4) hipLaunchKernelGGL((k_bin<F, 4>), dim3(nbin), dim3(256), 0, st, P, sigmas, coords, colors, V, L.ext_groups); \
        else hipLaunchKernelGGL((k_bin<F, 0>), dim3(nbin), dim3(256), 0, st, P, sigmas, coords, colors, V, L.ext_groups); } while (0)
        GSASR_BIN(true);
    } else {
        if (ncls <= 2 * SCAN_CHUNK) {
            hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, P, ncls, V.cell_count, V.cell_start, L.ext_groups, V.blockmax,
                               V.hdr);
        } else {
            const int nchunks = (ncls + SCAN_CHUNK - 1) / SCAN_CHUNK;
            hipLaunchKernelGGL(k_scan_local, dim3(nchunks), dim3(1024), 0, st, L.ncells, ncls, V.cell_count, V.cell_start,
                               V.scan_tot, L.ext_groups, V.blockmax, V.hdr, L.ncx, L.ncy, (int)(P.adapt_cells4 > 0.f), P.dead_off);
            hipLaunchKernelGGL(k_scan_fix, dim3(nchunks), dim3(1024), 0, st, P, ncls, V.cell_start, V.scan_tot, nchunks,
                               V.cell_count, V.hdr);
        }
        if (dims->s > 0) GSASR_BIN(false);
#undef GSASR_BIN
    }
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

}  // namespace gsasr_detail

extern "C" {

int gsasr_splat_plan(const float *sigmas, const float *coords, const float *colors, const gsasr_dims *dims,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    return plan_impl(sigmas, coords, colors, dims, workspace, workspace_bytes, stream, nullptr, StepSrc{});
}

}  // extern "C"
