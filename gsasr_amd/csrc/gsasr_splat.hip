// gsasr_splat.hip -- MI355X (gfx950 / CDNA4) 2D Gaussian-splatting rasterizer behind the C ABI of
// include/gsasr_splat.h.  Written for wave64 / 256 CUs / 8 XCDs; not a translation of the
// reference's CUDA (utils/gs_cuda*/gs.cu), whose semantics it reproduces:
//
//   forward : img[p,:] += sum_s [|dx|<=dmax & |dy|<=dmax] exp(w1_s * q_s(p)) * colors[s,:]
//   backward: analytic gradient of sum(grad_img * img) w.r.t. sigmas[s,3], coords[s,2], colors[s,3]
//
// Pipeline (all on the caller's stream, no host sync; DESIGN.md has the measurements):
//   plan     k_classify  per Gaussian: pixel window of (dmax box  ∩  sigma*sqrt(2 tau) support box) under the
//                        CONSERVATIVE cutoff tau = ln(s / eps); class {normal -> 16x16-px cell of its centre | large |
//                        dead (far / near: adapt_kcut)}; rank in its cell (one returning atomic per wave and distinct
//                        cell, one round trip); max extent of the normal class per group of 32 blocks; the px/py
//                        pixel-coordinate tables (double expression rounded to float, as gs.cu:27-28 does per pixel).
//            k_scan      exclusive scan of the cell histogram + max-extent reduction (one workgroup;
//                        k_scan_local + k_scan_fix for grids above 8192 cells).
//            k_bin       the cutoff the WINDOWS are built with -- tau' = ln(K / budget) <= tau, K counted from the cell
//                        histogram (adapt_kcut: same eps * max|colour| bound per pixel, 20-30% fewer pairs) -- then
//                        counting-sort placement (cell start + rank) fused with packing: 32-byte records
//                        {x, y, IX, NR, IY, r, g, b} (the coefficients of the completed-square exponent
//                        -(IX dx)^2 - (IY dy + NR IX dx)^2 in log2 units, computed in double), backward-epilogue constants, 16-byte windows with the
//                        per-tile-band column spans of the ellipse {exponent >= -tau}, and the first 8 bytes of the
//                        windows once more as a dense array for the coarse tests.
//   forward  k_render_fwd2 PIXEL-stationary: one wave64 = one 8x16 pixel sub-tile (2 px per lane, packed
//                        fp32), RGB accumulators in registers; four sub-tiles side by side per workgroup.  Two-level
//                        walk: the workgroup tests the candidates of its 32x16 tile (cell rows within the class'
//                        max extent, 64 per wave and chunk) ONCE, cooperatively, into a shared LDS list; each wave
//                        then runs the full window + span test on the survivors only, compacts the hits' records
//                        into its LDS stage and evaluates them from broadcast LDS reads; no atomics on the
//                        image, one coalesced store / read-modify-write of the tile.
//                        Images with 4096..8191 sub-tiles: eight waves per workgroup, two per sub-tile.
//            k_render_fwd_split   same per-wave code for small images: one sub-tile per workgroup, its
//                        chunks dealt to 2..16 waves, partial sums combined through LDS.
//   backward k_render_bwd  GAUSSIAN-stationary: one wave64 = one Gaussian (its records fetched by the scalar
//                        unit in one batch), lanes laid 16/32/64 wide over its window, two rows per trip
//                        (packed fp32), grad_img through L1/L2, three residual moments + three colour
//                        sums per lane, one LDS + DPP wave reduction, and the gradient written by the same
//                        wave (no finalize pass).  Gaussians of the "large" class are split into row chunks
//                        spread over all waves, combined with fp32 atomics; the wave finishing the last
//                        chunk writes the gradient.
//   prologue k_prologue_fwd/bwd  the reference's host prologue (activations + kernel frame) and its
//                        chain rule as one kernel each.
//   shard    k_band_select/merge  multi-GPU row bands: the Gaussians whose window crosses a band edge go to the
//                        neighbouring rank and their partial gradients come back.
//   batch    a batched canvas (gsasr_dims.batch > 1, row f2) runs B samples of different sizes through the same
//                        kernels: per-sample geometry (Geo) instead of the image's.
//   sampled  k_pts_count/scan/place, k_sample_fwd, k_pts_grads, k_sample_bwd   values and gradients at a list of
//                        pixels only (the reference's `sample_coords`, row f4): point-stationary two-level walk forward,
//                        Gaussian-stationary backward with eight Gaussians per wave.
//
// No MFMA: this is gather/scatter-accumulate with one transcendental per pair, not a contraction.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "gsasr_splat.h"

namespace {

constexpr int CELL = 16;        // binning cell side in pixels
constexpr int CELL_SHIFT = 4;
constexpr int NDEAD = 128;      // "dead" (nothing to draw) sub-classes: spreads the classify atomics of off-band Gaussians.
                                // Sub-classes 0..63: Gaussians that add EXACTLY nothing to this plan's pixels (non-finite; the
                                // bounded op's box misses the rows); 64..127: "near dead" -- the support does not reach, yet every
                                // term it would have added (< exp(-tau) each) is a skipped term the cutoff's bound must count
constexpr int NDEAD_NEAR = 64;
constexpr int SUBX = 8;         // forward sub-tile: 8 px wide x 16 px tall per wave64 (2 px per lane)
constexpr int SUBY = 16;
constexpr int SUBX_SHIFT = 3, SUBY_SHIFT = 4;
constexpr int RCAP_PX = 128;    // half-extent (px) above which a Gaussian is binned as "large"
#ifndef BWD_WAVES_N
#define BWD_WAVES_N 2
#endif
#ifndef BWD_OCC
#define BWD_OCC 7
#endif
#ifndef BWD_UNROLL_OCC
#define BWD_UNROLL_OCC 6
#endif
#ifndef FWD_WIDE_MIN
#define FWD_WIDE_MIN 25.0    // HR pixels per Gaussian from which the wide forward (16 x 16 sub-tiles, k_render_fwd16) is used:
                             // x5 -3..-7%, x8 -4..-10%, x12 -13..-15%, x16 -20%, x32 -25%; x4: +8% (profiles/r04_fwd_wide.txt)
#endif
#ifndef BWD_UNROLL_MIN
#define BWD_UNROLL_MIN 32.0
#endif
constexpr int BWD_WAVES = BWD_WAVES_N;  // waves (= consecutive cell-ordered Gaussians) per backward workgroup
constexpr int NCH = 64;         // row chunks a large Gaussian is split into in backward
#ifndef FUSED_MAX_BLOCKS
#define FUSED_MAX_BLOCKS 1536         // k_bin blocks up to which every block rebuilds the scan itself; beyond (393 216 Gaussians) the
                                      // 4 161 counter loads per block outweigh the scan kernel's launch: 16 Gaussians per LR pixel at
                                      // 1024^2 54.0 -> 44.7 us, the config-5 canvas 35.4 -> 32.4; at 262 144 Gaussians fused 22.8 vs 24.0
                                      // (development: GSASR_SPLAT_FUSED_MAX)
#endif
constexpr int FUSED_CELLS_HOST = 256 * 17;   // = FUSED_CELLS (k_bin<true, ..>): grids up to this many classes never run a scan kernel
constexpr int TL_W = 32;         // tile lists: tile width in pixels (height 16 or 32: Params::tl_hlog)
constexpr int TL_STRIDE = 16;    // ... words between two tiles' cursors: one per 64-byte line (atomics serialise per LINE)
constexpr int HDR_WORDS = 64;   // plan header (uint32): [0]=max x half-extent of normals, [1]=max y, [2]=largest cell count,
                                //   [3]=bits of sqrt(2 tau') the windows were built with, [4]=bits of tau', [5]=K (see adapt_kcut),
                                //   [6]=largest count of a 4 x 4 block of cells (block_count_max), [7]=near-dead Gaussians (adapt_kcut (3)),
                                //   [8],[9]=REACH in x, y: the half-extents the render kernels search with -- words 0, 1 shrunk to the
                                //   windows' cutoff tau' where no window is capped by the dmax box (reach_of), raised again (atomicMax)
                                //   by every Gaussian that kept its conservative window
constexpr double LOG2E = 1.4426950408889634074;

struct Params {
    int s, h, w, row0, row1;
    int bounded;     // 1: gs_cuda_dmax box test, 0: gs_cuda (no test)
    float dmax;      // box half-size (normalised units); +inf when !bounded
    float kcut;      // sqrt(2 tau) or 0 when the support cutoff is disabled (the CONSERVATIVE tau: classes, dead set)
    float adapt_cells;  // > 0: the windows are built with the data-derived cutoff tau' = ln(K / eps) <= tau, K = the most Gaussians
                     // whose dmax box can cover one pixel <= (largest cell count) * adapt_cells + (large class); 0: kcut everywhere
    int count_words; // words of one parity's counter array (cell counters + extent groups): what k_classify zeroes for the next plan
    int ext_groups;  // groups of 32 k_classify blocks (PlanView::blockmax)
    int dead_off;    // word offset of the dead sub-classes' counters inside a parity's counter array (count_at)
    int adapt_ring;  // 1: K also bounded from the SUPPORT (adapt_kcut: cells within the class' largest extent + a geometric tail)
    float cap_px_x, cap_px_y;  // the dmax box in pixels (smallest over the samples of a batch): a class extent below it means
                     // no Gaussian's window is capped by the box, so all of them shrink with the cutoff (reach_of)
    float adapt_cells4; // > 0: the same bound counted in blocks of 4 x 4 cells (block_count_max; sparse cells on large grids): K is
                     // the smaller of the two
    int ncx, ncy, ncells;
    unsigned flags;  // GSASR_FLAG_*
    int batch;       // 1: one image.  B > 1: B samples stacked in a canvas of B slots (h = B*slot rows, w columns)
    int slot;        // rows per slot (multiple of 16)
    int nper;        // Gaussians per sample (sample-major order)
    int part_k;      // tile-stationary backward: partial-gradient slots per Gaussian (PlanView::part)
    int geo_h, geo_w; // batched canvas whose samples all have ONE size (training crops): that size -- sample_geo is then arithmetic
                      // and the plan launches no k_batch_geo; 0: per-sample sizes in PlanView::geo
    int bt_hlog;     // ... and log2 of its tile height: 4 (32 x 16 px) or 5 (32 x 32, from 32 HR pixels per Gaussian); the tile
                     // kernel and the gather number a window's tiles with it (bt_tile_span)
    int grad_rows;   // rows per plane of a planar (GSASR_FLAG_CHW_GRAD) upstream gradient of a batched canvas
    int tl_hlog;     // tile lists (PlanView::tl_entries): log2 of the tile height, 4 (32 x 16 px: the 8 x 16 forward) or 5 (32 x 32: the
                     // wide forward); 0: this plan carries none
    int tl_cap;      // ... entries a tile's list can hold (a tile whose cursor ends above it is rendered by the search instead)
    int tl_ntx, tl_ntiles;   // ... tiles per row of tiles, tiles in all (over the rows [row0, row1))
};

// One sample of a batched canvas: its own pixel-grid size, its first canvas row and its px-table offset.
// A single image is the sample {h, w, 0, 0}.
struct Geo {
    int h, w, base, pxo;
};

// element strides of the caller's Gaussian arrays: [s,3]/[s,2]/[s,3], or columns of packed [s,8] records
__device__ __forceinline__ int stride3(const Params &P) { return (P.flags & GSASR_FLAG_STRIDE8) ? 8 : 3; }
__device__ __forceinline__ int stride2(const Params &P) { return (P.flags & GSASR_FLAG_STRIDE8) ? 8 : 2; }

struct PlanView {
    int4 *geo;              // [GSASR_MAX_BATCH] {h_b, w_b, first canvas row, px-table offset} (batched canvas only)
    unsigned *hdr;          // [HDR_WORDS]
    unsigned *cell_count;   // [ncells+1+NDEAD] (ncells = "large" class, ncells+1.. = "dead" sub-classes); this plan's parity
    unsigned *cell_count_next;  // the other parity's array: zeroed by k_classify for the next plan on this workspace
    unsigned *cell_start;   // [ncells+3]   exclusive scan of cell_count, last = s
    float *px, *py;         // [w], [h]
    unsigned *key;          // [s] class/cell of Gaussian i
    unsigned *rank;         // [s] position of Gaussian i inside its cell
    unsigned *blockmax;     // [16 * groups] max half-extents {x, y, -...} of the normal class per GROUP of 32 k_classify blocks, one
                            //       64-byte line each (atomicMax by the blocks: 64 atomics per line); they live behind the cell
                            //       counters of this plan's parity, so whoever zeroes those zeroes these
    unsigned *scan_tot;     // [ceil((ncells+1+NDEAD)/4096)] per-chunk totals of the two-pass scan
    float4 *rec;            // [2*s] {x,y,A,B},{C,r,g,b}   (cell order)
    float4 *fin;            // [2*s] backward constants {1/(1-rho^2), 1-rho^2, rho, 1/sx}, {1/sy, -, -, original index}
    float *sums;            // [8*s] raw backward sums {qA, qB, quA, qvB, qAB, Cr, Cg, Cb}: atomic accumulators of the large class
    unsigned *done;         // [s] row chunks of a large Gaussian finished so far (backward)
    uint4 *bbox;            // [2*s] {c0 | test<<15 | c1<<16, r0 | spans<<15 | r1<<16, span_lo[0..3], span_hi[0..3]},
                            //       {span_lo[4..7], span_hi[4..7], -, -}
    uint2 *win;             // [s] the first two words of bbox again, densely: what the coarse tests stream through
    uint4 *qspan;           // [s] (plans with slots only) per band of 8 rows of the window (8 bands at most): the range of 8-px
                            //       columns, counted from the window's first, that the ellipse {exponent >= -tau} reaches:
                            //       {lo[0..3], hi[0..3], lo[4..7], hi[4..7]} bytes; lo > hi = none
    unsigned *tl_cursor;    // [tl_ntiles * TL_STRIDE] tile lists: entries appended to tile t's list so far (zeroed by k_classify,
                            //       counted up by k_bin: wave-aggregated returning atomics); > tl_cap = overflowed
    uint2 *tl_entries;      // [tl_ntiles * tl_cap] tile t's list: {slot in cell order | needs the dmax test << 31, mask of the
                            //       tile's 8 x 8-px quadrants (bit 4 qy + qx) that the ellipse {exponent >= -tau'} reaches}
    float *part;            // [s * part_k * 8] tile-stationary backward: the raw sums {qA, qB, quA, qvB, qAB, Cr, Cg, Cb} of
                            //       Gaussian j (cell order) over the t-th 32x16-px tile of its window, written with plain stores
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

__device__ __forceinline__ Geo sample_geo(const Params &P, const PlanView &V, int b)
{
    if (P.batch <= 1) return Geo{P.h, P.w, 0, 0};
    if (P.geo_h) return Geo{P.geo_h, P.geo_w, b * P.slot, b * P.w};
    const int4 g = V.geo[b];
    return Geo{g.x, g.y, g.z, g.w};
}

struct Layout {
    size_t off_geo, off_hdr, off_count, off_start, off_px, off_py, off_key, off_rank, off_bmax, off_stot, off_rec, off_fin, off_sums, off_done, off_bbox, off_win, off_part, off_qspan;
    size_t off_tlc, off_tle;     // tile lists (at the END of the workspace: every other offset is the same with and without them)
    int tl_hlog, tl_cap, tl_ntx, tl_ntiles;
    bool tl_ok;                  // the lists may be READ: the plan's note says it wrote them (plan_layout)
    int part_k;
    size_t count_bytes;  // one array of per-cell counters + extent groups (there are two, used alternately: GSASR_FLAG_PARITY)
    size_t ext_off_words; // where the extent groups start inside such an array
    size_t dead_off_words; // ... and the counters of the dead sub-classes, ONE PER 64-BYTE LINE (count_at)
    int ext_groups;
    size_t total;
    int ncx, ncy, ncells;
};

bool dims_ok(const gsasr_dims *d)
{
    if (!(d && d->s >= 0 && d->h >= 2 && d->w >= 2 && d->h <= 32767 && d->w <= 32767 && d->c == 3 &&
          d->row0 >= 0 && d->row0 <= d->row1 && d->row1 <= d->h && !(d->dmax != d->dmax)))
        return false;
    if (d->batch <= 1) return true;
    // batched canvas: B slots of `slot` rows, whole canvas, uniform Gaussian count, per-sample sizes inside the slot
    if (d->batch > GSASR_MAX_BATCH || d->slot < 16 || (d->slot & 15) || d->h != d->batch * d->slot || d->row0 != 0 ||
        d->row1 != d->h || !d->sample_hw || d->s % d->batch != 0)
        return false;
    for (int b = 0; b < d->batch; ++b)
        if (d->sample_hw[2 * b] < 2 || d->sample_hw[2 * b] > d->slot || d->sample_hw[2 * b + 1] < 2 ||
            d->sample_hw[2 * b + 1] > d->w)
            return false;
    return true;
}

int batch_of(const gsasr_dims *d) { return d->batch > 1 ? d->batch : 1; }

int classify_blocks(const gsasr_dims *d)
{
    const int pxn = d->w * batch_of(d);  // one px table per sample
    const int n = d->s > pxn ? (d->s > d->h ? d->s : d->h) : (pxn > d->h ? pxn : d->h);
    return (n + 255) / 256;
}

// DEVELOPMENT switches (A/B runs of one build on one box; tools/collect_profiles.sh): environment variables that override a
// kernel choice the library makes by shape.  They are read ONLY when GSASR_SPLAT_DEV=1 is set as well -- a production process
// that happens to carry one of the names in its environment is not affected -- and each is read once.
//   GSASR_SPLAT_FWD_WIDE=0|1   GSASR_SPLAT_BWD=gaussian|tile|atomic   GSASR_SPLAT_BT_TALL=0|1   GSASR_SPLAT_ADAPT=0
//   GSASR_SPLAT_FUSED_MAX=<k_bin blocks>   GSASR_SPLAT_LISTS=0|1
// (GSASR_SPLAT_CUTOFF is not one of them: it is the documented process default of the support cutoff, INTEGRATION.md.)
const char *dev_switch(const char *name)
{
    static const bool on = [] { const char *e = getenv("GSASR_SPLAT_DEV"); return e && atoi(e) != 0; }();
    return on ? getenv(name) : nullptr;
}

// which backward kernel: explicit flag > development switch GSASR_SPLAT_BWD (gaussian | tile | atomic) > default
// development switch: GSASR_SPLAT_FWD_WIDE=0 / 1 forces the wide forward (16 x 16 sub-tiles) off / on
int fwd_wide_env()
{
    static std::atomic<int> cached{-2};
    int v = cached.load(std::memory_order_relaxed);
    if (v == -2) {
        const char *e = dev_switch("GSASR_SPLAT_FWD_WIDE");
        v = !e ? -1 : atoi(e) != 0;
        cached.store(v, std::memory_order_relaxed);
    }
    return v;
}

// Which forward kernel.  Scale factors from x5 up (FWD_WIDE_MIN HR pixels per Gaussian: windows of ~25 px and more), single
// images of at least 2 Mpx (16 384 sub-tiles of 8 x 16): 16 x 16 sub-tiles, four pixels per lane (k_render_fwd16).
// GSASR_FLAG_FWD_WIDE / _NARROW (or the environment) override the rule for A/B runs and tests -- the wide kernel renders any
// single image.
bool fwd_wants_wide(const gsasr_dims *d)
{
    const int want = (d->flags & GSASR_FLAG_FWD_WIDE) ? 1 : (d->flags & GSASR_FLAG_FWD_NARROW) ? 0 : fwd_wide_env();
    if (d->batch > 1 || want == 0) return false;
    if (want == 1) return true;
    const int rows = d->row1 - d->row0;
    const long nsub = (long)((d->w + SUBX - 1) / SUBX) * ((rows + SUBY - 1) / SUBY);
    // (pixels of the WHOLE grid per Gaussian = the scale factor squared: a row band that is handed every Gaussian of the image
    // has the image's window sizes, not those of rows * w / s)
    return nsub >= 2 * 8192 && (double)d->h * (double)d->w >= FWD_WIDE_MIN * (double)d->s;
}

int bwd_env()
{
    static std::atomic<int> cached{-1};
    int v = cached.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = dev_switch("GSASR_SPLAT_BWD");
        v = !e ? 0 : !strcmp(e, "gaussian") ? 1 : !strcmp(e, "tile") ? 2 : !strcmp(e, "atomic") ? 3 : 0;
        cached.store(v, std::memory_order_relaxed);
    }
    return v;
}

// Tile-stationary backward: every (32x16-px tile, Gaussian) pair leaves its partial sums in slot t of the Gaussian's
// own row of `part_k` slots, t = the tile's ordinal inside the Gaussian's window (row-major), so that nothing is
// accumulated atomically and the result does not depend on scheduling.  GSASR's Gaussians are about one LR pixel in
// size, i.e. a window of ~5.6 LR pixels = 23 px at x4 (2 x 3 tiles at most), 45 px at x8 (3 x 4): 8 slots cover x4 and
// below, 16 the larger scales; a Gaussian whose window spans more tiles than it has slots adds into `sums` with
// fp32 atomics instead (any input stays correct).  The window sizes live on the device, so the host picks by HR pixels
// per Gaussian, as for the Gaussian-stationary kernel's unrolling.
// Does this plan carry slots, i.e. will its backward be the tile-stationary kernel?  Explicitly (GSASR_FLAG_BWD_TILE),
// or by default where it is the faster one on this chip: measured the two backward kernels are level at
// GSASR's x4 (one Gaussian per 16 HR pixels; the Gaussian-stationary one 10% ahead), and the tile-stationary one wins
// from ~32 pixels per Gaussian up (x8: -7%), where a window holds enough quadrants to amortise the per-tile search.
bool bwd_wants_tile(const gsasr_dims *d)
{
    if (d->flags & (GSASR_FLAG_FORWARD_ONLY | GSASR_FLAG_BWD_GAUSSIAN | GSASR_FLAG_BWD_ATOMIC)) return false;
    if ((d->flags & GSASR_FLAG_BWD_TILE) || bwd_env() == 2) return true;
    // (by default only for whole images: a row band of a sharded image may hold all the Gaussians or just its own, so
    // its pixels per Gaussian say nothing about the window size -- the shard's caller knows the scale and sets the flag)
    if (bwd_env() != 0 || d->batch > 1 || d->row0 != 0 || d->row1 != d->h) return false;
    const double px_per_gaussian = (double)d->h * (double)d->w / (double)(d->s > 0 ? d->s : 1);
    // (round 4, with the windows of the data-derived cutoff: at 2048^2 x8 the Gaussian-stationary kernel is 7% ahead, at
    // 3072^2 x6 the tile-stationary one 3%, from 5120^2 up 4..10%: the line is drawn at 8 Mpx)
    return px_per_gaussian >= 32.0 && (double)d->h * (double)d->w >= 8388608.0;
}

// Tile height of the tile-stationary backward: 32 rows from 64 whole-grid HR pixels per Gaussian (x8 and up: windows of
// 45 px and more) on single images -- the per-tile search is shared by twice the pixels and a window meets 40% fewer tiles
// (slots written, and read by the gather): x8 -2% (the gather 94 -> 68 us, the tile kernel level; HBM traffic 1.72 -> 1.56 GB),
// x12 -2%, x16 -9%, x24 -17%; nothing at x6 (profiles/r04_bwd_experiments.txt (6)).  16 rows below that and on the batched
// canvas (slots are multiples of 16 rows).
// development switch: GSASR_SPLAT_BT_TALL=0 / 1
bool bt_tall(const gsasr_dims *d)
{
    static std::atomic<int> cached{-2};
    int v = cached.load(std::memory_order_relaxed);
    if (v == -2) {
        const char *e = dev_switch("GSASR_SPLAT_BT_TALL");
        v = !e ? -1 : atoi(e) != 0;
        cached.store(v, std::memory_order_relaxed);
    }
    if (d->batch > 1 || v == 0) return false;
    if (v == 1) return true;
    return (double)d->h * (double)d->w >= 64.0 * (double)(d->s > 0 ? d->s : 1);
}

int bwd_part_k(const gsasr_dims *d)
{
    // only plans made for the tile-stationary backward carry slots (32 * part_k bytes per Gaussian)
    if (!bwd_wants_tile(d)) return 0;
    const double px_per_gaussian = (double)d->h * (double)d->w / (double)(d->s > 0 ? d->s : 1);
    return px_per_gaussian >= 32.0 ? 16 : 8;
}

// Tile lists (round 5).  The render kernels used to FIND their Gaussians: every 32 x 16-px tile walked the cells within the
// class' largest extent of it and tested 4-5 candidates per hit (half of the forward at x4).  A plan with lists does that
// work once per Gaussian instead: k_bin, which holds the Gaussian's window and per-band ellipse spans in registers anyway,
// appends {slot, quadrant mask} to the list of every tile the ellipse reaches; the forward reads its tile's list and tests
// nothing but a mask bit.  Fixed capacity per tile (the host cannot know the window sizes, they live on the device): a tile
// whose list overflows is rendered by the search -- same results, graceful.  The "large" class (half-extent > 128 px: one
// Gaussian would enter thousands of lists) stays a segment every tile scans.
// development switch: GSASR_SPLAT_LISTS=0 / 1
int lists_env()
{
    static std::atomic<int> cached{-2};
    int v = cached.load(std::memory_order_relaxed);
    if (v == -2) {
        const char *e = dev_switch("GSASR_SPLAT_LISTS");
        v = !e ? -1 : atoi(e) != 0;
        cached.store(v, std::memory_order_relaxed);
    }
    return v;
}

// dense plan: at least one Gaussian per four pixels of the rows rendered (GSASR's 16 per LR pixel at x4 and below).  The 64
// Gaussians of a k_bin wave then share a handful of tiles, so their cursor atomics aggregate (tl_emit) and the lists cost less
// than the search they replace: 16 Gaussians per LR pixel at 1024^2 -5.8% per step, the config-5 canvas -5.2%.  At one
// Gaussian per LR pixel the atomics outweigh the search: config 2 +5%, config 3 +20% (profiles/r05_lists_ab.txt).
bool tl_dense(const gsasr_dims *d)
{
    const double rows = (double)(d->row1 - d->row0 > 0 ? d->row1 - d->row0 : 1);
    return 4.0 * (double)d->s >= (double)d->w * rows;
}

// log2 of the list tiles' height for a plan of these dims: 5 where the forward will be the wide kernel (32 x 32-px tiles),
// 4 for the two-level 8 x 16 kernels (32 x 16), 0 = no lists (small images: the split kernel; list_cap < 0; no Gaussians)
int tl_hlog_for(const gsasr_dims *d)
{
    if (d->list_cap < 0 || d->s <= 0 || lists_env() == 0) return 0;
    // by default for dense plans only (tl_dense); an explicit capacity (or the development switch) asks for them anywhere
    if (d->list_cap == 0 && lists_env() != 1 && !tl_dense(d)) return 0;
    const int rows = d->row1 - d->row0;
    if (fwd_wants_wide(d)) return 5;
    const long nsub = (long)((d->w + SUBX - 1) / SUBX) * ((rows + SUBY - 1) / SUBY);
    return (nsub >= 4096 || d->list_cap > 0) ? 4 : 0;      // (an explicit capacity asks for lists on any image: tests)
}

// Entries per tile.  gsasr_dims.list_cap when given; else four times what a tile of GSASR-shaped Gaussians (about one LR
// pixel in size: half-extent ~ 2.5 / sqrt(Gaussians per pixel), twice that allowed for) collects, + 64.
int tl_cap_for(const gsasr_dims *d, int hlog)
{
    if (!hlog) return 0;
    long cap = d->list_cap;
    if (cap <= 0) {
        const double rows = (double)(d->row1 - d->row0 > 0 ? d->row1 - d->row0 : 1);
        // (density over the rows rendered; a row band that is handed every Gaussian of the image sees most of them dead: its
        // tiles then have room to spare, never too little)
        const double rho = (double)d->s / ((double)d->w * rows), e = 5.0 / std::sqrt(rho > 1e-9 ? rho : 1e-9);
        const double want = 4.0 * rho * ((double)TL_W + e) * ((double)(1 << hlog) + e) + 64.0;
        cap = (long)std::fmin(want, 65536.0);
    }
    cap = (cap + 63) / 64 * 64;
    return (int)(cap > 65536 ? 65536 : cap);
}

Layout make_layout(const gsasr_dims *d, int part_k = -1, int tl_hlog = -1)
{
    Layout L{};
    L.ncx = (d->w + CELL - 1) / CELL;
    L.ncy = (d->h + CELL - 1) / CELL;
    L.ncells = L.ncx * L.ncy;
    const size_t ncls = (size_t)L.ncells + 1 + NDEAD, s = (size_t)d->s;
    size_t o = 0;
    L.off_hdr = o;    o += HDR_WORDS * 4;
    L.ext_groups = (classify_blocks(d) + 31) / 32;
    L.ext_off_words = align_up(ncls, 16);
    L.dead_off_words = L.ext_off_words + 16 * (size_t)L.ext_groups;
    L.count_bytes = align_up((L.dead_off_words + 16 * (size_t)NDEAD) * 4, 256);
    L.off_count = o;  o += 2 * L.count_bytes;
    L.off_geo = o;    o += GSASR_MAX_BATCH * 16;   // (outside the zeroed region: written once by k_batch_geo)
    L.off_start = o;  o += align_up((ncls + 1) * 4, 256);
    L.off_px = o;     o += align_up((size_t)d->w * 4 * (size_t)batch_of(d), 256);
    L.off_py = o;     o += align_up((size_t)d->h * 4, 256);
    L.off_key = o;    o += align_up(s * 4, 256);
    L.off_rank = o;   o += align_up(s * 4, 256);
    L.off_bmax = o;   // (unused since round 4: the per-block maxima became per-group maxima inside the counter arrays)
    L.off_stot = o;   o += align_up((ncls / 4096 + 2) * 4, 256);
    L.off_rec = o;    o += align_up(s * 32, 256);
    // (a forward-only plan -- inference -- carries none of the backward's records)
    const size_t bw = (d->flags & GSASR_FLAG_FORWARD_ONLY) ? 0 : s;
    L.off_fin = o;    o += align_up(bw * 32, 256);
    L.off_sums = o;   o += align_up(bw * 32, 256);
    L.off_done = o;   o += align_up(bw * 4, 256);
    L.off_bbox = o;   o += align_up(s * 32, 256);
    L.off_win = o;    o += align_up(s * 8, 256);
    L.part_k = part_k >= 0 ? part_k : bwd_part_k(d);
    L.off_part = o;   o += align_up(s * 32 * (size_t)L.part_k, 256);
    L.off_qspan = o;  o += L.part_k ? align_up(s * 16, 256) : 0;
    // tile lists LAST: a caller whose flags differ from the plan's (GSASR_FLAG_FWD_WIDE at forward time) lays out everything
    // else identically; whether the workspace carries lists, and of which tile height, is the plan's note (plan_layout)
    L.tl_hlog = tl_hlog >= 0 ? tl_hlog : tl_hlog_for(d);
    L.tl_cap = tl_cap_for(d, L.tl_hlog);
    L.tl_ntx = (d->w + TL_W - 1) / TL_W;
    L.tl_ntiles = L.tl_hlog ? L.tl_ntx * ((d->row1 - d->row0 + (1 << L.tl_hlog) - 1) >> L.tl_hlog) : 0;
    L.off_tlc = o;    o += align_up((size_t)L.tl_ntiles * TL_STRIDE * 4, 256);
    L.off_tle = o;    o += align_up((size_t)L.tl_ntiles * (size_t)L.tl_cap * 8, 256);
    L.tl_ok = L.tl_hlog != 0;
    L.total = o;
    return L;
}

PlanView make_view(const Layout &L, void *ws, unsigned flags = 0u)
{
    char *b = (char *)ws;
    PlanView V;
    V.geo = (int4 *)(b + L.off_geo);
    V.hdr = (unsigned *)(b + L.off_hdr);
    V.cell_count = (unsigned *)(b + L.off_count + ((flags & GSASR_FLAG_PARITY) ? L.count_bytes : 0));
    V.cell_count_next = (unsigned *)(b + L.off_count + ((flags & GSASR_FLAG_PARITY) ? 0 : L.count_bytes));
    V.cell_start = (unsigned *)(b + L.off_start);
    V.px = (float *)(b + L.off_px);
    V.py = (float *)(b + L.off_py);
    V.key = (unsigned *)(b + L.off_key);
    V.rank = (unsigned *)(b + L.off_rank);
    V.blockmax = V.cell_count + L.ext_off_words;
    V.scan_tot = (unsigned *)(b + L.off_stot);
    V.rec = (float4 *)(b + L.off_rec);
    V.fin = (float4 *)(b + L.off_fin);
    V.sums = (float *)(b + L.off_sums);
    V.done = (unsigned *)(b + L.off_done);
    V.bbox = (uint4 *)(b + L.off_bbox);
    V.win = (uint2 *)(b + L.off_win);
    V.part = (float *)(b + L.off_part);
    V.qspan = L.part_k ? (uint4 *)(b + L.off_qspan) : nullptr;
    V.tl_cursor = (unsigned *)(b + L.off_tlc);
    V.tl_entries = (uint2 *)(b + L.off_tle);
    return V;
}

// process default of the support cutoff: 0 = adaptive; first read from the environment (GSASR_SPLAT_CUTOFF), then
// whatever gsasr_set_default_cutoff stored.  One atomic word: setting and planning from different threads is a benign
// race on WHICH value a plan sees, never a torn one.
std::atomic<float> g_default_cutoff{-12345.f};

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

// development switch: GSASR_SPLAT_ADAPT=0 keeps the conservative tau = ln(s / eps) in the windows (A/B of adapt_kcut)
bool adapt_env()
{
    static std::atomic<int> cached{-1};
    int v = cached.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = dev_switch("GSASR_SPLAT_ADAPT");
        v = !e ? 1 : atoi(e) != 0;
        cached.store(v, std::memory_order_relaxed);
    }
    return v != 0;
}

// tau used for `s` Gaussians: explicit, process-fixed, or adaptive ln(s/eps) in [16, 104] (see the header)
float resolve_cutoff(float cutoff, int s)
{
    if (cutoff == 0.f) cutoff = default_cutoff();
    if (cutoff != 0.f) return cutoff;
    const double tau = std::log((double)(s > 1 ? s : 1) / (double)GSASR_SPLAT_DEFAULT_EPS);
    return (float)(tau < 16.0 ? 16.0 : tau > (double)GSASR_SPLAT_EXACT_CUTOFF ? (double)GSASR_SPLAT_EXACT_CUTOFF : tau);
}

// batched canvas whose samples all have one size (the training crops): that size; false otherwise
bool batch_uniform(const gsasr_dims *d, int &h, int &w)
{
    h = w = 0;
    if (d->batch <= 1) return false;
    for (int b = 1; b < d->batch; ++b)
        if (d->sample_hw[2 * b] != d->sample_hw[0] || d->sample_hw[2 * b + 1] != d->sample_hw[1]) return false;
    h = d->sample_hw[0];
    w = d->sample_hw[1];
    return true;
}

Params make_params(const gsasr_dims *d, const Layout &L)
{
    Params P;
    P.s = d->s; P.h = d->h; P.w = d->w; P.row0 = d->row0; P.row1 = d->row1;
    P.bounded = d->dmax >= 0.f;
    P.dmax = P.bounded ? d->dmax : INFINITY;
    const float tau = resolve_cutoff(d->cutoff, d->s);
    P.kcut = tau > 0.f ? (float)(std::sqrt(2.0 * (double)tau) * (1.0 + 1e-6)) : 0.f;
    // data-derived cutoff (adapt_kcut), under the adaptive default only -- an explicit tau (per call, per process,
    // environment) is used as given unless GSASR_FLAG_CUTOFF_CAP says it is an upper bound.  Both ops count K from the
    // SUPPORT (adapt_ring); the bounded op also from its dmax box (adapt_cells), the smaller count wins
    P.adapt_cells = P.adapt_cells4 = 0.f;
    P.adapt_ring = 0;
    P.count_words = (int)(L.count_bytes / 4);
    P.ext_groups = L.ext_groups;
    P.dead_off = (int)L.dead_off_words;
    {
        double wmin = d->w, hmin = d->h;
        if (d->batch > 1) {
            for (int b = 0; b < d->batch; ++b) {
                hmin = std::fmin(hmin, (double)d->sample_hw[2 * b]);
                wmin = std::fmin(wmin, (double)d->sample_hw[2 * b + 1]);
            }
        }
        P.cap_px_x = P.bounded ? (float)((double)d->dmax * 0.5 * (wmin - 1.0)) : INFINITY;
        P.cap_px_y = P.bounded ? (float)((double)d->dmax * 0.5 * (hmin - 1.0)) : INFINITY;
    }
    const bool adapt = ((d->cutoff == 0.f && default_cutoff() == 0.f) || (d->flags & GSASR_FLAG_CUTOFF_CAP)) && P.kcut > 0.f && adapt_env();
    P.adapt_ring = adapt && tau >= 16.f;       // (the tail constant of adapt_kcut is derived for tau >= 16)
    if (P.bounded && adapt) {
        const int B = batch_of(d);
        const double dpx = (double)d->dmax * 0.5 * (double)(d->w - 1), dpy = (double)d->dmax * 0.5 * (double)((B > 1 ? d->slot : d->h) - 1);
        const double cx = std::ceil(2.0 * std::floor(dpx + 1.02) / (double)CELL) + 1.0, cy = std::ceil(2.0 * std::floor(dpy + 1.02) / (double)CELL) + 1.0;
        const double cells = std::fmin(cx, (double)L.ncx) * std::fmin(cy, (double)L.ncy);
        P.adapt_cells = (float)std::fmin(cells, 1.0e9) * (1.f + 1e-6f);
        // Sparse cells (fewer than 8 Gaussians per cell on average: x8 and up) on a grid with a scan pass of its own: the
        // largest single cell is several times the mean there, the largest 64 x 64-px block is not -- count in blocks too
        if ((double)d->s < 8.0 * (double)L.ncells && L.ncells + 1 + NDEAD > FUSED_CELLS_HOST) {
            const double bx = std::ceil(2.0 * std::floor(dpx + 1.02) / (4.0 * CELL)) + 1.0, by = std::ceil(2.0 * std::floor(dpy + 1.02) / (4.0 * CELL)) + 1.0;
            P.adapt_cells4 = (float)std::fmin(std::fmin(bx, std::ceil(L.ncx / 4.0)) * std::fmin(by, std::ceil(L.ncy / 4.0)), 1.0e9) * (1.f + 1e-6f);
        }
    }
    P.ncx = L.ncx; P.ncy = L.ncy; P.ncells = L.ncells;
    P.flags = d->flags;
    if (!(P.flags & (GSASR_FLAG_BWD_GAUSSIAN | GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_ATOMIC)))   // (development A/B switch)
        P.flags |= bwd_env() == 2 ? GSASR_FLAG_BWD_TILE : bwd_env() == 3 ? GSASR_FLAG_BWD_ATOMIC : 0u;
    P.batch = batch_of(d);
    P.slot = d->batch > 1 ? d->slot : d->h;
    P.nper = d->batch > 1 ? d->s / d->batch : d->s;
    P.part_k = L.part_k;
    P.bt_hlog = bt_tall(d) ? 5 : 4;
    batch_uniform(d, P.geo_h, P.geo_w);
    P.grad_rows = d->grad_rows > 0 ? d->grad_rows : P.slot;
    P.tl_hlog = L.tl_hlog; P.tl_cap = L.tl_cap; P.tl_ntx = L.tl_ntx; P.tl_ntiles = L.tl_ntiles;
    return P;
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
std::atomic<unsigned long long> g_notes[NOTES];

unsigned note_slot(const void *ws) { return (unsigned)(((uintptr_t)ws >> 8) * 2654435761u >> 22) & (NOTES - 1); }

unsigned long long note_shape(const gsasr_dims *d)
{
    unsigned long long x = (unsigned long long)(unsigned)d->s * 0x9E3779B97F4A7C15ull;
    x ^= ((unsigned long long)(unsigned)d->h << 32 | (unsigned)d->w) * 0xC2B2AE3D27D4EB4Full;
    x ^= (unsigned long long)(unsigned)batch_of(d) * 0x27D4EB2F165667C5ull;
    return (x >> 40) & 0xffffull;
}

// (payload byte: slots per Gaussian in bits 0..4, the tile lists' tile height in bits 5..6: 0 none, 1 = 16 rows, 2 = 32)
void note_plan(const void *ws, const gsasr_dims *d, int part_k, int tl_hlog)
{
    const unsigned long long w = ((unsigned long long)(uintptr_t)ws & 0x0000ffffffffff00ull) << 16 | note_shape(d) << 8 |
                                 (unsigned long long)((part_k & 0x1f) | (tl_hlog ? (tl_hlog - 3) << 5 : 0));
    g_notes[note_slot(ws)].store(w, std::memory_order_relaxed);
}

// layout of the plan in `ws`: from the note its plan left, else from these dims -- except the tile lists, which a call uses
// only on the note's word (a lost note means the search, never a list nobody wrote)
Layout plan_layout(const gsasr_dims *d, const void *ws)
{
    int part_k = -1, tl_hlog = -1;
    if (ws) {
        const unsigned long long w = g_notes[note_slot(ws)].load(std::memory_order_relaxed);
        const unsigned long long key = ((unsigned long long)(uintptr_t)ws & 0x0000ffffffffff00ull) << 16 | note_shape(d) << 8;
        if ((w & ~0xffull) == key) {
            part_k = (int)(w & 0x1full);
            tl_hlog = (int)((w >> 5) & 3ull) ? (int)((w >> 5) & 3ull) + 3 : 0;
        }
    }
    // (without a note the list region is still SIZED from the dims -- the step entry points place their scratch behind the
    // plan -- but nothing reads it)
    Layout L = make_layout(d, part_k, tl_hlog);
    if (tl_hlog < 0) L.tl_ok = false;
    return L;
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
    bool near;           // dead, but the op would add its (tiny) tails to pixels of these rows: adapt_kcut's kind (3)
};

constexpr double WINDOW_EPS = 0.02;  // px; covers every rounding between these windows and the kernels' float tests

__device__ __forceinline__ Box gaussian_box(float sx, float sy, float x, float y, const Params &P, const Geo &g, float kcut)
{
    Box b;
    float ext_x = P.dmax, ext_y = P.dmax;
    if (kcut > 0.f) {  // marginal bound of the ellipse {exponent >= -tau}: |dx| <= sx*sqrt(2 tau), any rho
        // (|sigma|: the reference's formulas only see sigma^2 and 1/(sx sy), gs.cu:33-56, so a negative sigma -- the
        // raw op accepts any float, check.py feeds randn -- is a Gaussian like any other, with the sign of rho flipped)
        ext_x = fminf(ext_x, kcut * fabsf(sx));
        ext_y = fminf(ext_y, kcut * fabsf(sy));
    }
    // Pixel X sits at px = 2X/(w-1)-1, so |px - x| <= ext  <=>  |X - cxp| <= ext*hx with cxp = (x+1)*hx.
    // Evaluated in double (once per Gaussian); the float pixel table differs from the exact grid by
    // < 1e-2 px even at w = 32767, which WINDOW_EPS covers, so the window is tight to the pixel.
    // (g = the sample's own grid; its rows start at canvas row g.base.)
    const double hx = 0.5 * (double)(g.w - 1), hy = 0.5 * (double)(g.h - 1);
    const double cxp = ((double)x + 1.0) * hx, cyp = ((double)y + 1.0) * hy + (double)g.base;
    const double ex = (double)ext_x * hx, ey = (double)ext_y * hy;
    b.ex = (float)ex;
    b.ey = (float)ey;
    const double lox = ceil(cxp - ex - WINDOW_EPS), hix = floor(cxp + ex + WINDOW_EPS);
    const double loy = ceil(cyp - ey - WINDOW_EPS), hiy = floor(cyp + ey + WINDOW_EPS);
    const bool finite = (sx - sx == 0.f) && (sy - sy == 0.f) && (x - x == 0.f) && (y - y == 0.f);
    b.c0 = (int)fmax(lox, 0.0);
    b.c1 = (int)fmin(hix, (double)(g.w - 1));
    b.r0 = (int)fmax(loy, (double)max(P.row0, g.base));
    b.r1 = (int)fmin(hiy, (double)(min(P.row1, g.base + g.h) - 1));
    b.near = false;
    if (!finite || b.c0 > b.c1 || b.r0 > b.r1 || !(hix >= 0.0) || !(hiy >= (double)g.base)) {
        b.cls = 2;
        if (finite) {   // unbounded op: every pixel gets a term of it; bounded op: the pixels inside its dmax box do
            const double rmin = (double)max(P.row0, g.base), rmax = (double)(min(P.row1, g.base + g.h) - 1);
            const double dpx = (double)P.dmax * hx + 1.0, dpy = (double)P.dmax * hy + 1.0;    // (+1 px: on the counting side)
            b.near = !P.bounded || (cxp - dpx <= (double)(g.w - 1) && cxp + dpx >= 0.0 && cyp - dpy <= rmax && cyp + dpy >= rmin);
        }
    } else if (!(b.ex <= (float)RCAP_PX && b.ey <= (float)RCAP_PX))
        b.cls = 1;
    else
        b.cls = 0;
    return b;
}

// wave64 reductions of an unsigned, result uniform: four DPP row rotations leave every lane of a row of 16 with its row's
// result (VALU only; a __shfl_xor butterfly is six ds_bpermute round trips through the LDS pipe, on kernels whose run time is
// their dependent chain), the four rows are combined on the scalar unit.
template <int N>
__device__ __forceinline__ unsigned dpp_row_ror(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xf, 0xf, false);
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
    v = max(v, dpp_row_ror<1>(v));
    v = max(v, dpp_row_ror<2>(v));
    v = max(v, dpp_row_ror<4>(v));
    v = max(v, dpp_row_ror<8>(v));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ unsigned wave_add_u32(unsigned v)
{
    v += dpp_row_ror<1>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<8>(v);
    return ((unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16)) +
           ((unsigned)__builtin_amdgcn_readlane((int)v, 32) + (unsigned)__builtin_amdgcn_readlane((int)v, 48));
}

// wave64 sum; result valid in every lane (butterfly)
__device__ __forceinline__ float wave_sum(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The data-derived support cutoff (adaptive default; GSASR_FLAG_CUTOFF_CAP).  The windows are built with tau' <= tau (the
// conservative cutoff k_classify used) such that, on any pixel p of the rows rendered and for ANY input,
//     sum of the skipped terms  <  eps * max|colour|.
// A term is skipped when p lies outside its Gaussian's window.  Three kinds of Gaussians can lose terms at p:
//   (1) LIVE ones binned near p.  Each skipped term is < exp(-tau') |colour|.  How many there can be is bounded from the
//       plan's own histogram, two ways, the smaller count wins (K_live = min(box, ring) * (largest cell count) + large class):
//       box  (bounded op, gs_cuda_dmax/gs.cu:41-50: only a Gaussian whose dmax box covers p adds anything): a normal-class
//            Gaussian is binned by the cell of its (clamped, floored) centre and covers p only from the cells that the
//            2m + 1 pixels around p touch, m = floor(dmax_px + 1.02): Cx * Cy cells, C = ceil(2m / 16) + 1
//            [adapt_cells; on large sparse grids the same count in aligned 4 x 4-cell blocks, adapt_cells4];
//       ring (either op): with E = the class' largest half-extent under tau in x / in y (header words 0, 1: <= 130 px),
//            the CORE = the Cx * Cy cells that the 2 m + 1 pixels around p touch, m = E + 1, C = ceil(2m / 16) + 1: every
//            Gaussian within E pixels of p is binned there.  A Gaussian binned r >= 1 cells beyond the core along an axis is
//            at least E + 16 (r - 1) pixels away along it, where it is worth at most exp(-tau (d / E)^2) <= exp(-tau) q^(r-1),
//            q = exp(-32 tau / E) <= 0.02 (marginal of the bivariate normal; a window capped by the dmax box adds exactly
//            nothing beyond the cap).  Ring r holds 2 (Cx + Cy) + 8 r - 4 cells; summed over r >= 1 they add at most
//            2.1 (Cx + Cy) + 5 cells' worth of terms below exp(-tau) -- the CONSERVATIVE tau: they are paid from the budget
//            like (3), not counted at exp(-tau').  At x8 the core is 49 cells against the 2 809 of the box; at x4 16 against 64.
//   (2) the LARGE class (extent > 128 px): counted in full.
//   (3) NEAR-DEAD ones: classified dead because their support (under tau) does not reach the rows, though the op would
//       add their tails (bounded op: the dmax box does reach; unbounded op: every dead Gaussian).  Each term is
//       < exp(-tau); k_classify counts them in their own sub-classes (n_near) and the budget left for (1) + (2) is
//       eps - (n_near + ring tail) exp(-tau)  [n_near exp(-tau) = eps n_near / s under the adaptive tau = ln(s / eps)].
//   Gaussians whose box misses the rows (bounded op) and non-finite ones add exactly nothing, skipped or not.
// tau' = ln(K_live / budget) + 1e-3 (the log is the hardware's: 1 ulp), clamped to [16, tau].  Gaussians stacked on one spot
// make the largest count ~s and tau' = tau: nothing is lost on adversarial input (tests/test_adaptive_cutoff.py).
__device__ __forceinline__ float adapt_kcut(const Params &P, unsigned maxcount, unsigned nlarge, unsigned nnear, unsigned ext_x,
                                            unsigned ext_y, float &tau, unsigned &K, unsigned maxblock = 0u)
{
    const float tau_cap = 0.5f * P.kcut * P.kcut;
    tau = tau_cap;
    K = 0u;
    if (!(P.adapt_cells > 0.f) && !P.adapt_ring) return P.kcut;
    float Kn = INFINITY;
    if (P.adapt_cells > 0.f) {
        Kn = (float)maxcount * P.adapt_cells;
        if (P.adapt_cells4 > 0.f) Kn = fminf(Kn, (float)maxblock * P.adapt_cells4);   // the same pixels' boxes, in 4 x 4-cell blocks
    }
    float far_terms = (float)nnear;    // terms worth < exp(-tau) each, paid from the budget: near-dead Gaussians, ring tails
    if (P.adapt_ring) {
        // cells the 2 m + 1 pixels around a pixel touch, m = E + 1 (the centre is binned by its floor), per axis
        const float cx = fminf(ceilf((float)(2u * (min(ext_x, 130u) + 1u)) * (1.f / CELL)) + 1.f, (float)P.ncx);
        const float cy = fminf(ceilf((float)(2u * (min(ext_y, 130u) + 1u)) * (1.f / CELL)) + 1.f, (float)P.ncy);
        if ((float)maxcount * cx * cy < Kn) {
            Kn = (float)maxcount * cx * cy * (1.f + 1e-6f);
            far_terms += (float)maxcount * (2.1f * (cx + cy) + 5.f);
        }
    }
    const float Kf = fmaxf(Kn + (float)nlarge, 1.f);
    K = (unsigned)fminf(Kf, 4.0e9f);
    // what the far terms leave of eps; with less than a quarter left the conservative cutoff stays
    const float budget = GSASR_SPLAT_DEFAULT_EPS - far_terms * __builtin_amdgcn_exp2f(-tau_cap * 1.44269504f) * (1.f + 1e-5f);
    if (!(budget >= 0.25f * GSASR_SPLAT_DEFAULT_EPS)) return P.kcut;
    const float t = fmaxf(__log2f(Kf / budget) * 0.69314718f + 1e-3f, 16.f);
    if (!(t < tau_cap)) return P.kcut;
    tau = t;
    return fminf(sqrtf(2.f * t) * (1.f + 1e-6f), P.kcut);
}

// Half-extent the render kernels search with, from the class' conservative maximum `ext` (= ceil(largest extent) + 2 under
// the cutoff k_classify used): with windows built for a smaller cutoff (kc < P.kcut) every extent that is not capped by the
// dmax box shrinks by kc / P.kcut -- and none is capped when the largest one lies below the box (cap_px).
__device__ __forceinline__ unsigned reach_of(unsigned ext, float kc, float kcut, float cap_px)
{
    if (ext < 2u || !(kc < kcut) || !((float)(ext - 2u) < cap_px - 1.f)) return ext;
    return min(ext, (unsigned)ceilf((float)(ext - 2u) * (kc / kcut) * (1.f + 1e-6f)) + 2u);
}

// Histogram entry of class k.  Cells and the large class are dense; the dead sub-classes' counters sit one per 64-byte line
// behind them: a row band of a sharded image sees 7/8 of a million Gaussians there, one wave-aggregated atomic each, and
// atomics serialise per cache LINE (~12 ns): 64 dense counters are four lines (86 us of queueing), 128 padded ones 128 lines.
__device__ __forceinline__ unsigned count_index(int k, int ncells, int dead_off)
{
    return k <= ncells ? (unsigned)k : (unsigned)(dead_off + (k - ncells - 1) * 16);
}

// ---------------------------------------------------------------------------------------------------
// plan kernels
// ---------------------------------------------------------------------------------------------------
struct BatchSizes {   // kernel argument: the host's per-sample sizes
    unsigned short h[GSASR_MAX_BATCH], w[GSASR_MAX_BATCH];
};

__global__ __launch_bounds__(64) void k_batch_geo(BatchSizes S, int batch, int slot, int w, int4 *__restrict__ geo)
{
    const int b = threadIdx.x;
    if (b < batch) geo[b] = make_int4((int)S.h[b], (int)S.w[b], b * slot, b * w);
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// The reference's host prologue for one Gaussian (utils/gaussian_splatting.py:174-180 activations, :121-123 kernel
// frame): q = raw decoder output [sigma_x, sigma_y, rho, alpha, r, g, b, mu_x, mu_y] -> o = {sx, sy, rho | x, y | r, g, b}
__device__ __forceinline__ void prologue_one(const float *__restrict__ q, float step, int h, int w, float (&o)[8])
{
#pragma clang fp contract(off)   // torch rounds after every operation: no fused multiply-adds in here
    const float sx = 0.99999f * sigmoidf_(q[0]) + 1e-6f;  // the network's sigma_x is the ROW std
    const float sy = 0.99999f * sigmoidf_(q[1]) + 1e-6f;
    const float alpha = sigmoidf_(q[3]);
    const float W = (float)w, H = (float)h;
    // Rounded exactly as torch rounds the reference's expressions on the GPU: a tensor divided by a Python number is
    // multiplied by the number's fp32 reciprocal (BinaryDivTrueKernel), a tensor divided by the 0-dim step tensor is a true
    // division.  An ulp of a centre is 2e-4 px on a 3000-px image, which a sub-pixel Gaussian (sigma ~ 0.07 px: randn x 1.5
    // parameters at x8) turns into 1e-3 of its value -- the fused and the unfused host paths must not differ by that.
    const float iw1 = 1.f / (W - 1.f), ih1 = 1.f / (H - 1.f);
    o[0] = sy / step * 2.f * iw1;     // kernel's first sigma pairs with WIDTH (:121)
    o[1] = sx / step * 2.f * ih1;
    o[2] = 0.999999f * tanhf(q[2]);
    const float c0 = q[7] * 2.f - 1.f, c1 = q[8] * 2.f - 1.f;
    o[3] = (c0 + 1.f - (float)(1.0 / (double)w)) * W * iw1 - 1.f;   // align_corners=False -> True (:122-123)
    o[4] = (c1 + 1.f - (float)(1.0 / (double)h)) * H * ih1 - 1.f;
    o[5] = sigmoidf_(q[4]) * alpha;
    o[6] = sigmoidf_(q[5]) * alpha;
    o[7] = sigmoidf_(q[6]) * alpha;
}

// PROLOGUE: the step entry points hand over the RAW decoder parameters; the kernel-frame tensors are formed here (and
// stored for k_bin and the backward) instead of by a separate k_prologue_fwd launch in front of the plan.
// Where the step size of the prologue comes from: a device array step[b] (the reference's 0-dim tensor
// default_step_size / scale), or -- sm != nullptr -- the caller's scale_modify pairs themselves: the reference's
// `assert scale_modify[0] == scale_modify[1]; step = default_step_size / scale_modify[0]`
// (utils/gaussian_splatting.py:168-171) evaluated HERE, so that the host issues no division, comparison, copy or event
// per call.  The step is left in `keep[b]` for the backward; a differing pair sets the caller's sticky word.
struct StepSrc {
    const float *step;    // [batch] or nullptr
    const float *sm;      // scale_modify: sample b's pair at sm[b * stride + {0, 1}]
    int stride;
    float def_step;       // default_step_size
    int *mismatch;        // device int[2] or nullptr: {1 + sample index, bits of scale_modify[0]} of a differing pair
    float *keep;          // [GSASR_MAX_BATCH] in the step workspace: the step sizes used
};

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
// (Measured and dropped, profiles/r05_lists_ab.txt run r05d: per-lane atomics without the match-any loops, six tiles per round --
// config 2's k_bin +10.5 us instead of +6, config 4's plan +165 us instead of +81: the atomics, not the loops, are what costs.)
constexpr int TLB = 4;

template <int HLOG>
__device__ __forceinline__ void tl_emit(const Params &P, const PlanView &V, bool emit, unsigned j, const uint4 bb, const uint2 sp)
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
    int ix = 0, iy = 0;
    for (int base = 0; __ballot(base < ntile) != 0ull; base += TLB) {
        unsigned m[TLB], ti[TLB], pos[TLB];
        unsigned long long mine[TLB];
#pragma unroll
        for (int k = 0; k < TLB; ++k) {
            const bool v = base + k < ntile;
            m[k] = v ? tl_mask<HLOG>(tX0 + ix, tY0 + iy, bb, sp, P.row0) : 0u;
            ti[k] = (unsigned)((tY0 + iy) * P.tl_ntx + tX0 + ix);
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
                }
                bb.z = lo4[0]; bb.w = hi4[0];
                bc.x = lo4[1]; bc.y = hi4[1];
                bb.y |= 0x8000u;
            }
            {   // Rows the Gaussian-stationary backward sweeps: the window's rows rounded up to a whole number of trips
                // (8/4/2 rows for 16/32/64-lane columns) when the band has room -- the extra rows lie outside the window
                // (their terms are < exp(-tau), or fail the dmax test), and the ragged, masked last trip disappears.
                // Batched canvas: inside the sample's own rows (whatever gradient the caller left in the padding of the
                // slot must not be read).  Worked out here, once, instead of by every backward wave on its scalar unit.
                const int bwid = b.c1 - b.c0 + 1, nr = b.r1 - b.r0 + 1;
                const int rpt = bwid <= 16 ? 8 : (bwid <= 32 ? 4 : 2);
                const int pad = (rpt - (nr & (rpt - 1))) & (rpt - 1);
                const int lo = max(P.row0, g.base), hi = min(P.row1, g.base + g.h) - 1;
                int r0p = b.r0, r1p = b.r1;
                if (r1p + pad <= hi) r1p += pad;
                else if (r0p - pad >= lo) r0p -= pad;
                bc.z = (unsigned)r0p | ((unsigned)r1p << 16);
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
        tl_emit<TLH == 0 ? 4 : TLH>(P, V, valid && key < (unsigned)P.ncells, j, bb, make_uint2(bc.x, bc.y));
}

// ---------------------------------------------------------------------------------------------------
// forward: one wave64 per 8-wide x 16-tall pixel sub-tile (lane = column X, rows Y and Y+8, so the
// per-pair arithmetic is 2-wide packed fp32); four sub-tiles side by side per workgroup (32x16 px).
// Candidates are window-tested 64 at a time (one per lane); the records of the hits are compacted into a
// 2 KB per-wave LDS stage and then evaluated by all lanes from broadcast LDS reads (12 VALU instructions
// + 2 v_exp_f32 per record for 128 pixels).  Measured alternatives: fetching hit records with scalar
// loads (s_load_dwordx8, one or four in flight) was 4% slower at config 2 and 23% slower on small images.
// ---------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// Evaluate `n` records staged in LDS (32 B each, broadcast reads).  Unlike scalar-memory loads, LDS reads
// return in order, so the compiler can keep several records in flight behind counted lgkmcnt waits.
template <bool TEST>
__device__ __forceinline__ void fwd_eval_one(const float4 a, const float4 b, float px, v2f py, float dmax, v2f &ar,
                                             v2f &ag, v2f &ab)
{
    // a = {x, y, IX, NR}, b = {IY, r, g, b}:  exponent (log2) = -U^2 - Bq^2,  U = IX dx,  Bq = IY dy + NR U   (k_bin)
    const float dx = px - a.x;
    const v2f dy = py - a.y;
    const float u = a.z * dx;
    const float k0 = -u * u, ru = a.w * u;
    const v2f bq = b.x * dy + ru;
    const v2f pw = k0 - bq * bq;
    v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
    if (TEST) {
        const bool inx = fabsf(dx) <= dmax;
        v.x = (inx && fabsf(dy.x) <= dmax) ? v.x : 0.f;
        v.y = (inx && fabsf(dy.y) <= dmax) ? v.y : 0.f;
    }
    ar += v * b.y;
    ag += v * b.z;
    // ab += v * b.w with b.w read as the HIGH half of the (g, b) register pair.  The compiler folds four of the
    // five broadcasts {y, C, r, g, b} into op_sel but copies the fifth with a v_mov whatever the record order (one
    // VALU slot in 13.5 per record; -3% at config 2, -8% at config 3).  Inline asm is outside the compiler's
    // hazard recogniser, and this instruction may be scheduled right behind the v_exp_f32 that produces `v`
    // (trans-use hazard on gfx950: one wait state) -- hence the s_nop, without which results are garbage.
    {
        const v2f gb = {b.z, b.w};
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(ab) : "v"(gb), "v"(v));
    }
}

// The same evaluation split into its column part and its row part: the pixels of a lane of the WIDE forward
// (k_render_fwd16: four per lane, one column) share dx, U, -U^2 and NR U of a record -- four of the ten instructions.
struct FwdCol {
    float k0, ru;   // -U^2, NR U
    bool inx;       // (TEST) |dx| <= dmax
};

template <bool TEST>
__device__ __forceinline__ FwdCol fwd_eval_col(const float4 a, float px, float dmax)
{
    const float dx = px - a.x;
    const float u = a.z * dx;
    FwdCol c;
    c.k0 = -u * u;
    c.ru = a.w * u;
    c.inx = !TEST || fabsf(dx) <= dmax;
    return c;
}

template <bool TEST>
__device__ __forceinline__ void fwd_eval_row(const FwdCol c, const float4 a, const float4 b, v2f py, float dmax, v2f &ar,
                                             v2f &ag, v2f &ab)
{
    const v2f dy = py - a.y;
    const v2f bq = b.x * dy + c.ru;
    const v2f pw = c.k0 - bq * bq;
    v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
    if (TEST) {
        v.x = (c.inx && fabsf(dy.x) <= dmax) ? v.x : 0.f;
        v.y = (c.inx && fabsf(dy.y) <= dmax) ? v.y : 0.f;
    }
    ar += v * b.y;
    ag += v * b.z;
    {   // (see fwd_eval_one: b.w as the high half of the (g, b) pair; s_nop for the trans-use hazard)
        const v2f gb = {b.z, b.w};
        asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(ab) : "v"(gb), "v"(v));
    }
}

// records [beg, end) of the stage on the lane's pixels; acc = {rA, gA, bA, rB, gB, bB}.  HALVES: which of the lane's two row
// pairs the records of this list reach -- 1 = pair A (the sub-tile's rows 0..7), 2 = pair B (rows 8..15), 3 = both
// (k_render_fwd16 sorts a chunk's hits by that: a window that ends in the upper half costs one row part, not two).
template <bool TEST, int HALVES>
__device__ __forceinline__ void fwd_eval_lds16(const float4 *__restrict__ st, int beg, int end, float px, v2f pyA, v2f pyB,
                                               float dmax, v2f (&acc)[6])
{
    int i = beg;
    for (; i + 1 < end; i += 2) {   // two records per iteration so their dependent chains interleave
        const float4 a0 = st[2 * i], b0 = st[2 * i + 1], a1 = st[2 * i + 2], b1 = st[2 * i + 3];
        const FwdCol c0 = fwd_eval_col<TEST>(a0, px, dmax), c1 = fwd_eval_col<TEST>(a1, px, dmax);
        if (HALVES & 1) {
            fwd_eval_row<TEST>(c0, a0, b0, pyA, dmax, acc[0], acc[1], acc[2]);
            fwd_eval_row<TEST>(c1, a1, b1, pyA, dmax, acc[0], acc[1], acc[2]);
        }
        if (HALVES & 2) {
            fwd_eval_row<TEST>(c0, a0, b0, pyB, dmax, acc[3], acc[4], acc[5]);
            fwd_eval_row<TEST>(c1, a1, b1, pyB, dmax, acc[3], acc[4], acc[5]);
        }
    }
    if (i < end) {
        const float4 a = st[2 * i], b = st[2 * i + 1];
        const FwdCol c = fwd_eval_col<TEST>(a, px, dmax);
        if (HALVES & 1) fwd_eval_row<TEST>(c, a, b, pyA, dmax, acc[0], acc[1], acc[2]);
        if (HALVES & 2) fwd_eval_row<TEST>(c, a, b, pyB, dmax, acc[3], acc[4], acc[5]);
    }
}

template <bool TEST>
__device__ __forceinline__ void fwd_eval_lds(const float4 *__restrict__ st, int beg, int end, float px, v2f py,
                                             float dmax, v2f &ar, v2f &ag, v2f &ab)
{
    int i = beg;
    for (; i + 1 < end; i += 2) {   // two records per iteration so their dependent chains interleave
        const float4 a0 = st[2 * i], b0 = st[2 * i + 1], a1 = st[2 * i + 2], b1 = st[2 * i + 3];
        fwd_eval_one<TEST>(a0, b0, px, py, dmax, ar, ag, ab);
        fwd_eval_one<TEST>(a1, b1, px, py, dmax, ar, ag, ab);
    }
    if (i < end) fwd_eval_one<TEST>(st[2 * i], st[2 * i + 1], px, py, dmax, ar, ag, ab);
}

// RECORD-PAIR evaluation (round 5).  fwd_eval_one packs the two PIXELS of a lane: of its instructions per record the four
// that depend on the column alone (dx, U, -U^2, NR U) have nothing to pack with.  Packed over two RECORDS instead -- the stage
// holds pairs interleaved {x0,x1, y0,y1, IX0,IX1, NR0,NR1 | IY0,IY1, r0,r1, g0,g1, b0,b1} -- every instruction is packed:
// 4 (column) + 2 rows x 6 = 16 packed + 4 v_exp_f32 per record PAIR and 128 pixels = 96 cycles against 2 x 64.  The sums of
// the even and the odd records of a list are kept apart (two accumulators per channel and row) and added at the end; a list
// of odd length ends in a zero record (colour 0).
#ifndef FWD_PAIR
#define FWD_PAIR 1
#endif
constexpr int STAGE_F4 = 136;    // float4 per wave's stage: 64 records + a zero record behind each of the two lists (pairs)

__device__ __forceinline__ void stage_put_pair(float4 *stage, int slot, const float4 a, const float4 b)
{
    float *p = reinterpret_cast<float *>(stage) + (slot >> 1) * 16 + (slot & 1);
    p[0] = a.x; p[2] = a.y; p[4] = a.z; p[6] = a.w; p[8] = b.x; p[10] = b.y; p[12] = b.z; p[14] = b.w;
}

template <bool TEST>
__device__ __forceinline__ void fwd_eval_pair(const float4 q0, const float4 q1, const float4 q2, const float4 q3, float px, v2f py,
                                              float dmax, v2f (&acc)[6])
{
    // q0 = {x0,x1,y0,y1}, q1 = {IX0,IX1,NR0,NR1}, q2 = {IY0,IY1,r0,r1}, q3 = {g0,g1,b0,b1}; acc = {rA, gA, bA, rB, gB, bB} (row A / B)
    const v2f x = {q0.x, q0.y}, y = {q0.z, q0.w}, ix = {q1.x, q1.y}, nr = {q1.z, q1.w}, iy = {q2.x, q2.y};
    const v2f cr = {q2.z, q2.w}, cg = {q3.x, q3.y}, cb = {q3.z, q3.w};
    const v2f dx = px - x;
    const v2f u = ix * dx;
    const v2f k0 = -u * u, ru = nr * u;
    const v2f dyA = py.x - y, dyB = py.y - y;
    const v2f bqA = iy * dyA + ru, bqB = iy * dyB + ru;
    const v2f pwA = k0 - bqA * bqA, pwB = k0 - bqB * bqB;
    v2f vA = {__builtin_amdgcn_exp2f(pwA.x), __builtin_amdgcn_exp2f(pwA.y)};
    v2f vB = {__builtin_amdgcn_exp2f(pwB.x), __builtin_amdgcn_exp2f(pwB.y)};
    if (TEST) {
        const bool in0 = fabsf(dx.x) <= dmax, in1 = fabsf(dx.y) <= dmax;
        vA.x = (in0 && fabsf(dyA.x) <= dmax) ? vA.x : 0.f;
        vA.y = (in1 && fabsf(dyA.y) <= dmax) ? vA.y : 0.f;
        vB.x = (in0 && fabsf(dyB.x) <= dmax) ? vB.x : 0.f;
        vB.y = (in1 && fabsf(dyB.y) <= dmax) ? vB.y : 0.f;
    }
    acc[0] += vA * cr; acc[1] += vA * cg; acc[2] += vA * cb;
    acc[3] += vB * cr; acc[4] += vB * cg; acc[5] += vB * cb;
}

// pairs [beg, end) of the stage
template <bool TEST>
__device__ __forceinline__ void fwd_eval_lds_pairs(const float4 *__restrict__ st, int beg, int end, float px, v2f py, float dmax,
                                                   v2f (&acc)[6])
{
    int i = beg;
    for (; i + 1 < end; i += 2) {   // two pairs per iteration so their dependent chains interleave
        const float4 a0 = st[4 * i], a1 = st[4 * i + 1], a2 = st[4 * i + 2], a3 = st[4 * i + 3];
        const float4 b0 = st[4 * i + 4], b1 = st[4 * i + 5], b2 = st[4 * i + 6], b3 = st[4 * i + 7];
        fwd_eval_pair<TEST>(a0, a1, a2, a3, px, py, dmax, acc);
        fwd_eval_pair<TEST>(b0, b1, b2, b3, px, py, dmax, acc);
    }
    if (i < end) fwd_eval_pair<TEST>(st[4 * i], st[4 * i + 1], st[4 * i + 2], st[4 * i + 3], px, py, dmax, acc);
}

// One chunk of a wave's walk: compact the hits' records (ra, rb of the hit lanes; those that need the dmax test behind the
// others) into the wave's LDS stage and evaluate them on the lane's two pixels from broadcast LDS reads.
template <bool BOUNDED>
__device__ __forceinline__ void fwd_stage_eval(float4 *stage, bool hit, bool needs, const float4 ra, const float4 rb, int lane, float px,
                                               v2f py, float dmax, v2f &ar, v2f &ag, v2f &ab)
{
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned long long m0 = __ballot(hit && !needs), m1 = BOUNDED ? __ballot(hit && needs) : 0ull;
    const int n0 = __builtin_popcountll(m0), n1 = __builtin_popcountll(m1);
    if (n0 + n1 == 0) return;
#if FWD_PAIR
    const int b1 = (n0 + 1) & ~1;     // first slot of the tested list (the lists are padded to whole pairs)
    __builtin_amdgcn_wave_barrier();
    if (hit) {
        const int r = needs ? __builtin_popcountll(m1 & below) : __builtin_popcountll(m0 & below);
        const int slot = needs ? b1 + r : r;
        stage_put_pair(stage, slot, ra, rb);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (((needs ? n1 : n0) & 1) && r == (needs ? n1 : n0) - 1) stage_put_pair(stage, slot + 1, z, z);   // the odd list's zero record
    }
    __builtin_amdgcn_wave_barrier();
    v2f acc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = (v2f){0.f, 0.f};
    fwd_eval_lds_pairs<false>(stage, 0, (n0 + 1) >> 1, px, py, dmax, acc);
    if (BOUNDED) fwd_eval_lds_pairs<true>(stage, b1 >> 1, (b1 + n1 + 1) >> 1, px, py, dmax, acc);
    ar += (v2f){acc[0].x + acc[0].y, acc[3].x + acc[3].y};
    ag += (v2f){acc[1].x + acc[1].y, acc[4].x + acc[4].y};
    ab += (v2f){acc[2].x + acc[2].y, acc[5].x + acc[5].y};
#else
    __builtin_amdgcn_wave_barrier();
    if (hit) {
        const int slot = needs ? n0 + __builtin_popcountll(m1 & below) : __builtin_popcountll(m0 & below);
        stage[2 * slot] = ra;
        stage[2 * slot + 1] = rb;
    }
    __builtin_amdgcn_wave_barrier();
    fwd_eval_lds<false>(stage, 0, n0, px, py, dmax, ar, ag, ab);
    if (BOUNDED) fwd_eval_lds<true>(stage, n0, n0 + n1, px, py, dmax, ar, ag, ab);
#endif
}

// Candidate index of this lane in flat chunk `c` of the concatenated segments.  The segment table lives in
// lanes (lane r: start `sbeg`, exclusive/inclusive prefix of the lengths `pex`/`pin`); `r` is the first
// segment that reaches into the chunk (wave-uniform, advanced monotonically).  Returns 0xffffffff for
// lanes past the end.  A chunk overlaps one or two segments at 16 Gaussians per cell and 3-5 when cells
// are sparse (x12 inference), so every chunk is full instead of one mostly-empty chunk per segment.
__device__ __forceinline__ unsigned fwd_candidate(unsigned c, int lane, int nseg, int &r, unsigned sbeg, unsigned pex,
                                                  unsigned pin)
{
    const unsigned q0 = c * 64u, q = q0 + (unsigned)lane;
    while (r < nseg && (unsigned)__builtin_amdgcn_readlane((int)pin, r) <= q0) ++r;
    unsigned j = 0xffffffffu;
    for (int rr = r; rr < nseg; ++rr) {
        const unsigned p0 = (unsigned)__builtin_amdgcn_readlane((int)pex, rr);
        if (p0 >= q0 + 64u) break;
        const unsigned p1 = (unsigned)__builtin_amdgcn_readlane((int)pin, rr);
        const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)sbeg, rr);
        if (q >= p0 && q < p1) j = b + (q - p0);
    }
    return j;
}

// One wave, one 8x16 sub-tile at (sx0, sy0): accumulate every Gaussian binned near it into ar/ag/ab
// (lane = column sx0 + lane%8, rows sy0 + lane/8 and +8).  With nparts > 1 the 64-candidate chunks are
// dealt round-robin to `nparts` waves and the caller adds their partial sums.
// LARGE_ONLY: the walk over the "large" class alone (what a list kernel still has to scan: tile lists hold the normal class).
template <bool BOUNDED, bool LARGE_ONLY = false>
__device__ __forceinline__ void fwd_tile(const Params &P, const PlanView &V, int sx0, int sy0, int lane,
                                         unsigned part, unsigned nparts, float4 *stage, v2f &ar, v2f &ag, v2f &ab)
{
    const int sx1 = min(sx0 + SUBX - 1, P.w - 1), sy1 = min(sy0 + SUBY - 1, P.row1 - 1);
    const int X = sx0 + (lane & 7), Y0 = sy0 + (lane >> 3), Y1 = Y0 + 8;
    // (batched canvas: slots are whole tile rows, so a sub-tile belongs to one sample; its px table is the sy0/slot-th)
    const float px = V.px[(P.batch > 1 ? (sy0 / P.slot) * P.w : 0) + min(X, P.w - 1)];
    const v2f py = {V.py[min(Y0, P.h - 1)], V.py[min(Y1, P.h - 1)]};

    const float4 *__restrict__ rec = V.rec;
    const uint4 *__restrict__ bbox = V.bbox;
    const unsigned *__restrict__ cs = V.cell_start;
    const int wty = (sy0 - P.row0) >> SUBY_SHIFT, wtx = sx0 >> SUBX_SHIFT;

    // Segment table: lane r holds [beg,end) of cell row cy0+r restricted to the columns a normal-class
    // Gaussian can reach this sub-tile from (max half-extent from the plan header); one more lane holds
    // the large class.  One vector round trip instead of a dependent scalar load per row.
    const int rx = LARGE_ONLY ? 0 : (int)V.hdr[8], ry = LARGE_ONLY ? 0 : (int)V.hdr[9];
    int nseg = 0;
    unsigned sbeg = 0, send = 0;
    if (rx > 0) {
        const int cx0 = max(sx0 - rx, 0) >> CELL_SHIFT, cx1 = min((sx1 + rx) >> CELL_SHIFT, P.ncx - 1);
        const int cy0 = max(sy0 - ry, 0) >> CELL_SHIFT, cy1 = min((sy1 + ry) >> CELL_SHIFT, P.ncy - 1);
        nseg = cy1 - cy0 + 1;
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

    // Flat walk over full 64-candidate chunks of the concatenated segments (this wave takes chunks
    // part, part+nparts, ...), software-pipelined: the window record of the NEXT chunk is in flight while
    // the hits of the current one are evaluated.
    const unsigned len = send - sbeg;
    unsigned pin = len;  // inclusive prefix sum of the segment lengths over the lanes
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)pin, o);
        if (lane >= o) pin += v;
    }
    const unsigned pex = pin - len;
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)pin, nseg - 1);
    const unsigned nchunks = (total + 63u) >> 6;
    const uint4 dead = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);  // c0 = r0 = 32767 > c1 = r1 = 0: overlaps no tile
    int rseg = 0;
    unsigned c = part;
    unsigned j = c < nchunks ? fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
    uint4 bb = dead;
    uint2 bs = make_uint2(0u, 0u);
    if (j != 0xffffffffu) {
        bb = bbox[2 * (size_t)j];
        bs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)j + 1);
    }
    while (c < nchunks) {
        const unsigned nc = c + nparts;
        const unsigned nj = nc < nchunks ? fwd_candidate(nc, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
        uint4 nbb = dead;
        uint2 nbs = make_uint2(0u, 0u);
        if (nj != 0xffffffffu) {
            nbb = bbox[2 * (size_t)nj];
            nbs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)nj + 1);
        }

        const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
        const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
        bool hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
        if (bb.y & 0x8000u) {  // per-tile-row column spans (k_bin)
            const unsigned t = (unsigned)(wty - ((r0 - P.row0) >> SUBY_SHIFT)) & 7u, sh = (t & 3u) * 8u;
            const unsigned lo = t < 4u ? bb.z : bs.x, hi = t < 4u ? bb.w : bs.y;
            const int txr = wtx - (c0 >> SUBX_SHIFT);
            hit &= (txr >= (int)((lo >> sh) & 0xffu)) & (txr <= (int)((hi >> sh) & 0xffu));
        }
        const bool needs = BOUNDED && (bb.x & 0x8000u) != 0u;
        // Compact the hits' records into this wave's LDS stage (untested ones first), then every lane
        // evaluates all of them from broadcast LDS reads.
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
        if (hit) {
            const float4 *src = rec + 2 * (size_t)j;
            ra = src[0];
            rb = src[1];
        }
        fwd_stage_eval<BOUNDED>(stage, hit, needs, ra, rb, lane, px, py, P.dmax, ar, ag, ab);
        c = nc; j = nj; bb = nbb; bs = nbs;
    }
}

// ---- two-level walk (large images) ---------------------------------------------------------------------
// Measured: finding a sub-tile's hits -- fetching the 24-byte windows of every Gaussian binned within reach and
// testing them, 64 per wave and round trip -- is HALF of the forward's time (the candidates come from cells within
// the class' MAXIMUM extent, 4-5x more than hit).  The four waves of a workgroup render four sub-tiles side by side,
// and their candidate sets are almost the same, so the workgroup walks the candidates of its 32x16 tile ONCE,
// cooperatively: each wave tests a quarter of the chunks against the whole tile (window only, 8 bytes per
// candidate) and appends the survivors to a shared list in LDS; after a barrier every wave runs the full test
// (window + ellipse span) over that list only.  Rounds of 1024 candidates bound the list.
constexpr int COARSE_CHUNKS = 4;                          // coarse chunks per wave and round
constexpr int COARSE_LIST = 4 * COARSE_CHUNKS * 64;       // candidates per round and part = capacity of the shared list

// PARTS = 2: eight waves per workgroup, two per sub-tile taking alternate chunks of the survivor list (images with
// fewer sub-tiles than the chip has wave slots); the caller adds the two partial sums.
template <bool BOUNDED, int PARTS>
__device__ __forceinline__ void fwd_block(const Params &P, const PlanView &V, int bx0, int by0, int wv, int lane,
                                          float4 *stage, unsigned *s_list, unsigned *s_cnt, v2f &ar, v2f &ag, v2f &ab)
{
    const int bx1 = min(bx0 + 4 * SUBX - 1, P.w - 1), by1 = min(by0 + SUBY - 1, P.row1 - 1);
    const int sx0 = bx0 + (wv & 3) * SUBX;
    const unsigned part = (unsigned)(wv >> 2);
    const bool live = sx0 < P.w;                              // wave-uniform (image width not a multiple of 32)
    const int sx1 = min(sx0 + SUBX - 1, P.w - 1);
    const int X = sx0 + (lane & 7);
    const float px = V.px[(P.batch > 1 ? (by0 / P.slot) * P.w : 0) + min(X, P.w - 1)];
    const int Y0 = by0 + (lane >> 3);
    const v2f py = {V.py[min(Y0, P.h - 1)], V.py[min(Y0 + 8, P.h - 1)]};
    const float4 *__restrict__ rec = V.rec;
    const uint4 *__restrict__ bbox = V.bbox;
    const unsigned *__restrict__ cs = V.cell_start;
    const int wtx = sx0 >> SUBX_SHIFT;

    // segment table of the 32x16 tile (every wave builds the same one: a single vector round trip)
    const int rx = (int)V.hdr[8], ry = (int)V.hdr[9];
    int nseg = 0;
    unsigned sbeg = 0, send = 0;
    if (rx > 0) {
        const int cx0 = max(bx0 - rx, 0) >> CELL_SHIFT, cx1 = min((bx1 + rx) >> CELL_SHIFT, P.ncx - 1);
        const int cy0 = max(by0 - ry, 0) >> CELL_SHIFT, cy1 = min((by1 + ry) >> CELL_SHIFT, P.ncy - 1);
        nseg = cy1 - cy0 + 1;
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
    const unsigned len = send - sbeg;
    unsigned pin = len;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)pin, o);
        if (lane >= o) pin += v;
    }
    const unsigned pex = pin - len;
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)pin, nseg - 1);
    const unsigned nchunks = (total + 63u) >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int rseg = 0;

    for (unsigned base = 0, round = 0; base < nchunks; base += 4u * PARTS * COARSE_CHUNKS, ++round) {
        unsigned *cnt = s_cnt + (round & 1u);
        // ---- phase A: this wave's share of the round's chunks against the whole tile -------------------
        unsigned cj[COARSE_CHUNKS];
        uint2 cw[COARSE_CHUNKS];
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {     // all loads of the round in flight together
            const unsigned c = base + (unsigned)wv + 4u * PARTS * (unsigned)k;
            cj[k] = c < nchunks ? fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
            cw[k] = make_uint2(0x7fffu, 0x7fffu);
            if (cj[k] != 0xffffffffu) cw[k] = V.win[cj[k]];
        }
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {
            const int c0 = (int)(cw[k].x & 0x7fffu), c1 = (int)(cw[k].x >> 16);
            const int r0 = (int)(cw[k].y & 0x7fffu), r1 = (int)(cw[k].y >> 16);
            const bool hit = (c0 <= bx1) & (c1 >= bx0) & (r0 <= by1) & (r1 >= by0);
            const unsigned long long m = __ballot(hit);
            if (m) {
                unsigned at = 0;
                if (lane == 0) at = atomicAdd(cnt, (unsigned)__builtin_popcountll(m));
                at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
                if (hit) s_list[at + (unsigned)__builtin_popcountll(m & below)] = cj[k];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cnt[(round + 1u) & 1u] = 0u;   // nobody touches the other counter before the next barrier
        const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)*cnt);
        // ---- phase B: the full test of the tile's survivors against this wave's sub-tile(s) -------------
        const int sy0 = by0, sy1 = by1;
        const int wty = (sy0 - P.row0) >> SUBY_SHIFT;
        if (live) {
            const unsigned q0 = part * 64u;
            unsigned j = q0 + lane < n ? s_list[q0 + lane] : 0xffffffffu;
            const uint4 dead = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);
            uint4 bb = dead;
            uint2 bs = make_uint2(0u, 0u);
            if (j != 0xffffffffu) {
                bb = bbox[2 * (size_t)j];
                bs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)j + 1);
            }
            for (unsigned q = q0; q < n; q += 64u * PARTS) {
                const unsigned nq = q + 64u * PARTS + (unsigned)lane;
                const unsigned nj = nq < n ? s_list[nq] : 0xffffffffu;
                uint4 nbb = dead;
                uint2 nbs = make_uint2(0u, 0u);
                if (nj != 0xffffffffu) {
                    nbb = bbox[2 * (size_t)nj];
                    nbs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)nj + 1);
                }
                const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
                const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
                bool hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
                if (bb.y & 0x8000u) {  // per-tile-row column spans (k_bin)
                    const unsigned t = (unsigned)(wty - ((r0 - P.row0) >> SUBY_SHIFT)) & 7u, sh = (t & 3u) * 8u;
                    const unsigned lo = t < 4u ? bb.z : bs.x, hi = t < 4u ? bb.w : bs.y;
                    const int txr = wtx - (c0 >> SUBX_SHIFT);
                    hit &= (txr >= (int)((lo >> sh) & 0xffu)) & (txr <= (int)((hi >> sh) & 0xffu));
                }
                const bool needs = BOUNDED && (bb.x & 0x8000u) != 0u;
                float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
                if (hit) {
                    const float4 *src = rec + 2 * (size_t)j;
                    ra = src[0];
                    rb = src[1];
                }
                fwd_stage_eval<BOUNDED>(stage, hit, needs, ra, rb, lane, px, py, P.dmax, ar, ag, ab);
                j = nj; bb = nbb; bs = nbs;
            }
        }
        __syncthreads();   // the list is rewritten in the next round
    }
}

__device__ __forceinline__ void fwd_store(const Params &P, const PlanView &V, float *__restrict__ img, int sx0, int sy0,
                                          int lane, v2f ar, v2f ag, v2f ab)
{
    const int X = sx0 + (lane & 7), Y0 = sy0 + (lane >> 3), Y1 = Y0 + 8;
    if (X >= P.w) return;
    const bool store = P.flags & GSASR_FLAG_OVERWRITE_IMAGE;
    bool ok0 = Y0 < P.row1, ok1 = Y1 < P.row1;
    // CHW: planar [3, rows, w]; batched canvas: [B, 3, slot, w] (HWC is simply the canvas [B*slot, w, 3])
    size_t plane = (size_t)(P.row1 - P.row0) * P.w, chw0 = (size_t)(Y0 - P.row0) * P.w + X, chw1 = chw0 + 8 * (size_t)P.w;
    if (P.batch > 1) {
        // pixels of the slot outside the sample's own h_b x w_b grid are padding: stored as zero, never added to
        const int smp = sy0 / P.slot;
        const Geo g = sample_geo(P, V, smp);
        const bool inx = X < g.w, in0 = inx && Y0 - g.base < g.h, in1 = inx && Y1 - g.base < g.h;
        if (!in0) { ar.x = ag.x = ab.x = 0.f; ok0 = ok0 && store; }
        if (!in1) { ar.y = ag.y = ab.y = 0.f; ok1 = ok1 && store; }
        plane = (size_t)P.slot * P.w;
        chw0 = ((size_t)smp * 3 * P.slot + (size_t)(Y0 - g.base)) * P.w + X;
        chw1 = chw0 + 8 * (size_t)P.w;
    }
    if (P.flags & GSASR_FLAG_CHW_IMAGE) {
        if (ok0) {
            float *o = img + chw0;
            if (store) { o[0] = ar.x; o[plane] = ag.x; o[2 * plane] = ab.x; }
            else { o[0] += ar.x; o[plane] += ag.x; o[2 * plane] += ab.x; }
        }
        if (ok1) {
            float *o = img + chw1;
            if (store) { o[0] = ar.y; o[plane] = ag.y; o[2 * plane] = ab.y; }
            else { o[0] += ar.y; o[plane] += ag.y; o[2 * plane] += ab.y; }
        }
        return;
    }
    if (ok0) {
        float *o = img + ((size_t)(Y0 - P.row0) * P.w + X) * 3;
        if (store) { o[0] = ar.x; o[1] = ag.x; o[2] = ab.x; }
        else { o[0] += ar.x; o[1] += ag.x; o[2] += ab.x; }
    }
    if (ok1) {
        float *o = img + ((size_t)(Y1 - P.row0) * P.w + X) * 3;
        if (store) { o[0] = ar.y; o[1] = ag.y; o[2] = ab.y; }
        else { o[0] += ar.y; o[1] += ag.y; o[2] += ab.y; }
    }
}

// XCD-aware tile order: blocks are dealt round-robin to the 8 XCDs (block b -> XCD b%8), so give each XCD
// a contiguous band of tile rows: neighbouring tiles then share records in ONE L2.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned nb)
{
    const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u;
    return xcd * q + min(xcd, r) + (b >> 3);
}

// Two-level walk (fwd_block).  PARTS = 1: large images, the workgroup shape of k_render_fwd.  PARTS = 2: images
// with fewer sub-tiles than wave slots -- eight waves, two per sub-tile, partial sums combined through LDS.
template <bool BOUNDED, int PARTS>
__global__ __launch_bounds__(256 * PARTS) void k_render_fwd2(Params P, PlanView V, float *__restrict__ img, int tiles_x)
{
    const unsigned t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int bx = (int)(t % (unsigned)tiles_x), by = (int)(t / (unsigned)tiles_x);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float4 s_stage[4 * PARTS][STAGE_F4];
    __shared__ unsigned s_list[COARSE_LIST * PARTS];
    __shared__ unsigned s_cnt[2];
    __shared__ float s_part[PARTS > 1 ? 4 : 1][6][64];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    v2f ar = {0.f, 0.f}, ag = {0.f, 0.f}, ab = {0.f, 0.f};
    const int bx0 = bx * 4 * SUBX, by0 = P.row0 + by * SUBY;
    fwd_block<BOUNDED, PARTS>(P, V, bx0, by0, wv, lane, s_stage[wv], s_list, s_cnt, ar, ag, ab);
    const int sub = wv & 3;
    if (PARTS > 1) {   // (fwd_block ends on a barrier)
        if (wv >= 4) {
            float (*o)[64] = s_part[sub];
            o[0][lane] = ar.x; o[1][lane] = ar.y; o[2][lane] = ag.x; o[3][lane] = ag.y; o[4][lane] = ab.x; o[5][lane] = ab.y;
        }
        __syncthreads();
        if (wv >= 4) return;
        float (*o)[64] = s_part[sub];
        ar.x += o[0][lane]; ar.y += o[1][lane]; ag.x += o[2][lane]; ag.y += o[3][lane]; ab.x += o[4][lane]; ab.y += o[5][lane];
    }
    if (bx0 + sub * SUBX < P.w) fwd_store(P, V, img, bx0 + sub * SUBX, by0, lane, ar, ag, ab);
}

// WIDE forward (large windows: x8 and up, single images).  A wave owns a 16 x 16 sub-tile, lane = column sx0 + lane%16 and
// the four rows sy0 + lane/16 + {0, 4 | 8, 12}: two packed row pairs in ONE column, so a record's column arithmetic is done
// once for four pixels (ten instructions + four v_exp_f32 per record and 256 pixels, against two times six + two for the
// 8 x 16 sub-tile), and half as many waves search.  The workgroup's tile is 32 x 32 (2 x 2 sub-tiles), walked like
// fwd_block: cooperative window test of the candidates against the tile, then every wave tests the survivors against its
// own sub-tile.  Against this stands the coarser cull ((w + 16)(h + 16) instead of (w + 8)(h + 16) evaluated pixels per
// window): it pays from ~40-px windows up.
constexpr int WIDE = 16;   // sub-tile side

__device__ __forceinline__ void fwd_store_px(const Params &P, float *__restrict__ img, int X, int Y, float r, float g, float b)
{
    if (Y >= P.row1) return;
    const bool store = P.flags & GSASR_FLAG_OVERWRITE_IMAGE;
    if (P.flags & GSASR_FLAG_CHW_IMAGE) {
        const size_t plane = (size_t)(P.row1 - P.row0) * P.w;
        float *o = img + (size_t)(Y - P.row0) * P.w + X;
        if (store) { o[0] = r; o[plane] = g; o[2 * plane] = b; }
        else { o[0] += r; o[plane] += g; o[2 * plane] += b; }
        return;
    }
    float *o = img + ((size_t)(Y - P.row0) * P.w + X) * 3;
    if (store) { o[0] = r; o[1] = g; o[2] = b; }
    else { o[0] += r; o[1] += g; o[2] += b; }
}

template <bool BOUNDED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_render_fwd16(Params P, PlanView V, float *__restrict__ img, int tiles_x)
{
    const unsigned tt = xcd_swizzle(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float4 s_stage[4][128];
    __shared__ unsigned s_list[COARSE_LIST];
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    const int bx0 = (int)(tt % (unsigned)tiles_x) * 2 * WIDE, by0 = P.row0 + (int)(tt / (unsigned)tiles_x) * 2 * WIDE;
    const int bx1 = min(bx0 + 2 * WIDE - 1, P.w - 1), by1 = min(by0 + 2 * WIDE - 1, P.row1 - 1);
    const int sx0 = bx0 + (wv & 1) * WIDE, sy0 = by0 + (wv >> 1) * WIDE;
    const bool live = sx0 < P.w && sy0 < P.row1;                  // wave-uniform
    const int sx1 = min(sx0 + WIDE - 1, P.w - 1), sy1 = min(sy0 + WIDE - 1, P.row1 - 1);
    const int X = sx0 + (lane & 15), Y = sy0 + (lane >> 4);
    const float px = V.px[min(X, P.w - 1)];
    const v2f pyA = {V.py[min(Y, P.h - 1)], V.py[min(Y + 4, P.h - 1)]};
    const v2f pyB = {V.py[min(Y + 8, P.h - 1)], V.py[min(Y + 12, P.h - 1)]};
    float4 *stage = s_stage[wv];
    const float4 *__restrict__ rec = V.rec;
    const uint4 *__restrict__ bbox = V.bbox;
    const unsigned *__restrict__ cs = V.cell_start;
    const int wtx = sx0 >> SUBX_SHIFT, wty = (sy0 - P.row0) >> SUBY_SHIFT;   // in the units of k_bin's spans (8 columns, 16 rows)

    // segment table of the 32x32 tile (every wave builds the same one; cf. fwd_block)
    const int rx = (int)V.hdr[8], ry = (int)V.hdr[9];
    int nseg = 0;
    unsigned sbeg = 0, send = 0;
    if (rx > 0) {
        const int cx0 = max(bx0 - rx, 0) >> CELL_SHIFT, cx1 = min((bx1 + rx) >> CELL_SHIFT, P.ncx - 1);
        const int cy0 = max(by0 - ry, 0) >> CELL_SHIFT, cy1 = min((by1 + ry) >> CELL_SHIFT, P.ncy - 1);
        nseg = cy1 - cy0 + 1;
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
    const unsigned len = send - sbeg;
    unsigned pin = len;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)pin, o);
        if (lane >= o) pin += v;
    }
    const unsigned pex = pin - len;
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)pin, nseg - 1);
    const unsigned nchunks = (total + 63u) >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int rseg = 0;
    v2f acc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = (v2f){0.f, 0.f};

    for (unsigned base = 0, round = 0; base < nchunks; base += 4u * COARSE_CHUNKS, ++round) {
        unsigned *cnt = s_cnt + (round & 1u);
        // ---- phase A: this wave's share of the round's chunks against the whole tile -------------------
        unsigned cj[COARSE_CHUNKS];
        uint2 cw[COARSE_CHUNKS];
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {
            const unsigned c = base + (unsigned)wv + 4u * (unsigned)k;
            cj[k] = c < nchunks ? fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
            cw[k] = make_uint2(0x7fffu, 0x7fffu);
            if (cj[k] != 0xffffffffu) cw[k] = V.win[cj[k]];
        }
#pragma unroll
        for (int k = 0; k < COARSE_CHUNKS; ++k) {
            const int c0 = (int)(cw[k].x & 0x7fffu), c1 = (int)(cw[k].x >> 16);
            const int r0 = (int)(cw[k].y & 0x7fffu), r1 = (int)(cw[k].y >> 16);
            const bool hit = (c0 <= bx1) & (c1 >= bx0) & (r0 <= by1) & (r1 >= by0);
            const unsigned long long m = __ballot(hit);
            if (m) {
                unsigned at = 0;
                if (lane == 0) at = atomicAdd(cnt, (unsigned)__builtin_popcountll(m));
                at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
                if (hit) s_list[at + (unsigned)__builtin_popcountll(m & below)] = cj[k];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_cnt[(round + 1u) & 1u] = 0u;
        const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)*cnt);
        // ---- phase B: the full test of the tile's survivors against this wave's sub-tile ------------------
        if (live) {
            unsigned j = (unsigned)lane < n ? s_list[lane] : 0xffffffffu;
            const uint4 dead = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);
            uint4 bb = dead;
            uint2 bs = make_uint2(0u, 0u);
            if (j != 0xffffffffu) {
                bb = bbox[2 * (size_t)j];
                bs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)j + 1);
            }
            for (unsigned q = 0; q < n; q += 64u) {
                const unsigned nq = q + 64u + (unsigned)lane;
                const unsigned nj = nq < n ? s_list[nq] : 0xffffffffu;
                uint4 nbb = dead;
                uint2 nbs = make_uint2(0u, 0u);
                if (nj != 0xffffffffu) {
                    nbb = bbox[2 * (size_t)nj];
                    nbs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)nj + 1);
                }
                const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
                const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
                bool hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
                if (bb.y & 0x8000u) {  // per-tile-row column spans (k_bin), in 8-column units: the sub-tile covers two
                    const unsigned t = (unsigned)(wty - ((r0 - P.row0) >> SUBY_SHIFT)) & 7u, sh = (t & 3u) * 8u;
                    const unsigned lo = t < 4u ? bb.z : bs.x, hi = t < 4u ? bb.w : bs.y;
                    const int txr = wtx - (c0 >> SUBX_SHIFT);
                    hit &= (txr + 1 >= (int)((lo >> sh) & 0xffu)) & (txr <= (int)((hi >> sh) & 0xffu));
                }
                const bool needs = BOUNDED && (bb.x & 0x8000u) != 0u;
                // Sort the hits by the row pairs their window reaches: both, only the upper eight rows of the sub-tile (pair A),
                // only the lower eight (pair B); windows cut by the dmax box (exact in-kernel test) stay one list on both.
                const int cls = !hit ? 4 : needs ? 3 : r1 < sy0 + 8 ? 1 : r0 >= sy0 + 8 ? 2 : 0;
                const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
                const unsigned long long m3 = BOUNDED ? __ballot(cls == 3) : 0ull;
                const int n0 = __builtin_popcountll(m0), n1 = __builtin_popcountll(m1), n2 = __builtin_popcountll(m2);
                const int n3 = __builtin_popcountll(m3);
                if (n0 + n1 + n2 + n3) {
                    __builtin_amdgcn_wave_barrier();
                    if (hit) {
                        const unsigned long long mine = cls == 0 ? m0 : cls == 1 ? m1 : cls == 2 ? m2 : m3;
                        const int base = cls == 0 ? 0 : cls == 1 ? n0 : cls == 2 ? n0 + n1 : n0 + n1 + n2;
                        const int slot = base + __builtin_popcountll(mine & below);
                        const float4 *src = rec + 2 * (size_t)j;
                        stage[2 * slot] = src[0];
                        stage[2 * slot + 1] = src[1];
                    }
                    __builtin_amdgcn_wave_barrier();
                    fwd_eval_lds16<false, 3>(stage, 0, n0, px, pyA, pyB, P.dmax, acc);
                    fwd_eval_lds16<false, 1>(stage, n0, n0 + n1, px, pyA, pyB, P.dmax, acc);
                    fwd_eval_lds16<false, 2>(stage, n0 + n1, n0 + n1 + n2, px, pyA, pyB, P.dmax, acc);
                    if (BOUNDED) fwd_eval_lds16<true, 3>(stage, n0 + n1 + n2, n0 + n1 + n2 + n3, px, pyA, pyB, P.dmax, acc);
                }
                j = nj; bb = nbb; bs = nbs;
            }
        }
        __syncthreads();   // the list is rewritten in the next round
    }
    if (live && X < P.w) {
        fwd_store_px(P, img, X, Y, acc[0].x, acc[1].x, acc[2].x);
        fwd_store_px(P, img, X, Y + 4, acc[0].y, acc[1].y, acc[2].y);
        fwd_store_px(P, img, X, Y + 8, acc[3].x, acc[4].x, acc[5].x);
        fwd_store_px(P, img, X, Y + 12, acc[3].y, acc[4].y, acc[5].y);
    }
}

// ---- forward from the plan's tile lists (round 5) -------------------------------------------------------------
// The tile's hit list was written by k_bin (tl_emit): nothing is searched and nothing is tested but a mask bit.  A wave streams
// the tile's entries 64 at a time, keeps those whose quadrant mask meets its own sub-tile, gathers their 32-byte records into
// its LDS stage and evaluates them as the search kernels do (same fwd_eval_lds: same sums in another order).  The records of
// chunk k+1 are in flight while chunk k is evaluated.  No barriers, no shared lists: the four waves of a workgroup only share
// the tile.  A tile whose list overflowed its capacity is rendered by the one-level search (fwd_tile); the "large" class is
// scanned by every tile as before.
template <bool BOUNDED, int PARTS>
__global__ __launch_bounds__(256 * PARTS) void k_render_fwd_list(Params P, PlanView V, float *__restrict__ img, int tiles_x)
{
    const unsigned t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int bx = (int)(t % (unsigned)tiles_x), by = (int)(t / (unsigned)tiles_x);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = wv & 3;
    const unsigned part = (unsigned)(wv >> 2);
    __shared__ float4 s_stage[4 * PARTS][STAGE_F4];
    __shared__ float s_part[PARTS > 1 ? 4 : 1][6][64];
    float4 *stage = s_stage[wv];
    v2f ar = {0.f, 0.f}, ag = {0.f, 0.f}, ab = {0.f, 0.f};
    const int bx0 = bx * 4 * SUBX, by0 = P.row0 + by * SUBY;
    const int sx0 = bx0 + sub * SUBX;
    if (sx0 < P.w) {   // wave-uniform (image width not a multiple of 32)
        // (the first 64 entries are requested together with the cursor that says how many of them are real: one dependent
        // round trip less in a wave whose whole life is three or four of them)
        const uint2 *__restrict__ ent = V.tl_entries + (size_t)t * (size_t)P.tl_cap;
        const unsigned q_first = part * 64u;
        uint2 e_first = q_first + (unsigned)lane < (unsigned)P.tl_cap ? ent[q_first + lane] : make_uint2(0u, 0u);
        const unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)V.tl_cursor[(size_t)t * TL_STRIDE]);
        if (cnt > (unsigned)P.tl_cap) {
            fwd_tile<BOUNDED, false>(P, V, sx0, by0, lane, part, (unsigned)PARTS, stage, ar, ag, ab);
        } else {
            const int X = sx0 + (lane & 7);
            const float px = V.px[(P.batch > 1 ? (by0 / P.slot) * P.w : 0) + min(X, P.w - 1)];
            const int Y0 = by0 + (lane >> 3);
            const v2f py = {V.py[min(Y0, P.h - 1)], V.py[min(Y0 + 8, P.h - 1)]};
            const float4 *__restrict__ rec = V.rec;
            const unsigned mybits = 0x11u << sub;
            const uint2 none = make_uint2(0u, 0u);
            // chunk k: entries -> hits, slots -> records (registers) -> stage -> evaluation; k+1's records fly under k's evaluation
            unsigned q = q_first;
            uint2 e = q + (unsigned)lane < cnt ? e_first : none;
            bool hit = (e.y & mybits) != 0u, needs = BOUNDED && (e.x >> 31) != 0u;
            float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
            if (hit) {
                const float4 *src = rec + 2 * (size_t)(e.x & 0x7fffffffu);
                ra = src[0];
                rb = src[1];
            }
            unsigned nq = q + 64u * PARTS;
            uint2 ne = nq + (unsigned)lane < cnt ? ent[nq + lane] : none;
            while (q < cnt) {
                // next chunk: hits and their record loads, then the entries of the one after -- requested before this
                // chunk is evaluated
                const bool chit = hit, cneeds = needs;
                const float4 ca = ra, cb = rb;
                q = nq;
                e = ne;
                hit = (e.y & mybits) != 0u;
                needs = BOUNDED && (e.x >> 31) != 0u;
                if (hit) {
                    const float4 *src = rec + 2 * (size_t)(e.x & 0x7fffffffu);
                    ra = src[0];
                    rb = src[1];
                }
                nq = q + 64u * PARTS;
                ne = nq + (unsigned)lane < cnt ? ent[nq + lane] : none;
                fwd_stage_eval<BOUNDED>(stage, chit, cneeds, ca, cb, lane, px, py, P.dmax, ar, ag, ab);
            }
            // the large class (half-extent > 128 px) is in nobody's list
            const unsigned nlarge = V.cell_start[P.ncells + 1] - V.cell_start[P.ncells];
            if (__builtin_amdgcn_readfirstlane((int)nlarge) != 0)
                fwd_tile<BOUNDED, true>(P, V, sx0, by0, lane, part, (unsigned)PARTS, stage, ar, ag, ab);
        }
    }
    if (PARTS > 1) {
        if (wv >= 4) {
            float (*o)[64] = s_part[sub];
            o[0][lane] = ar.x; o[1][lane] = ar.y; o[2][lane] = ag.x; o[3][lane] = ag.y; o[4][lane] = ab.x; o[5][lane] = ab.y;
        }
        __syncthreads();
        if (wv >= 4) return;
        float (*o)[64] = s_part[sub];
        ar.x += o[0][lane]; ar.y += o[1][lane]; ag.x += o[2][lane]; ag.y += o[3][lane]; ab.x += o[4][lane]; ab.y += o[5][lane];
    }
    if (sx0 < P.w) fwd_store(P, V, img, sx0, by0, lane, ar, ag, ab);
}

// The wide forward from tile lists: 32 x 32-px list tiles = the 2 x 2 sub-tiles of 16 x 16 px of k_render_fwd16's workgroup.
// A wave's hits are the entries whose quadrant mask meets the four quadrants of its sub-tile; the row-pair classes of
// k_render_fwd16 (window reaches both halves of the sub-tile / the upper eight rows only / the lower eight only) come from the
// same mask.  An overflowed tile falls back to a one-level search over the tile's cells with the wide evaluation.
template <bool BOUNDED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_render_fwd16_list(Params P, PlanView V, float *__restrict__ img, int tiles_x)
{
    const unsigned tt = xcd_swizzle(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ float4 s_stage[4][128];
    float4 *stage = s_stage[wv];
    const int bx0 = (int)(tt % (unsigned)tiles_x) * 2 * WIDE, by0 = P.row0 + (int)(tt / (unsigned)tiles_x) * 2 * WIDE;
    const int sx0 = bx0 + (wv & 1) * WIDE, sy0 = by0 + (wv >> 1) * WIDE;
    if (!(sx0 < P.w && sy0 < P.row1)) return;                     // wave-uniform; no barriers below
    const int sx1 = min(sx0 + WIDE - 1, P.w - 1), sy1 = min(sy0 + WIDE - 1, P.row1 - 1);
    const int X = sx0 + (lane & 15), Y = sy0 + (lane >> 4);
    const float px = V.px[min(X, P.w - 1)];
    const v2f pyA = {V.py[min(Y, P.h - 1)], V.py[min(Y + 4, P.h - 1)]};
    const v2f pyB = {V.py[min(Y + 8, P.h - 1)], V.py[min(Y + 12, P.h - 1)]};
    const float4 *__restrict__ rec = V.rec;
    const unsigned long long below = (1ull << lane) - 1ull;
    v2f acc[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) acc[k] = (v2f){0.f, 0.f};

    // stage the hits of one chunk sorted by row-pair class (cls 0 both, 1 upper rows only, 2 lower rows only, 3 dmax-tested,
    // 4 none) and evaluate them
    auto stage_eval = [&](int cls, const float4 ra, const float4 rb) {
        const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
        const unsigned long long m3 = BOUNDED ? __ballot(cls == 3) : 0ull;
        const int n0 = __builtin_popcountll(m0), n1 = __builtin_popcountll(m1), n2 = __builtin_popcountll(m2);
        const int n3 = __builtin_popcountll(m3);
        if (n0 + n1 + n2 + n3 == 0) return;
        __builtin_amdgcn_wave_barrier();
        if (cls < 4) {
            const unsigned long long mine = cls == 0 ? m0 : cls == 1 ? m1 : cls == 2 ? m2 : m3;
            const int base = cls == 0 ? 0 : cls == 1 ? n0 : cls == 2 ? n0 + n1 : n0 + n1 + n2;
            const int slot = base + __builtin_popcountll(mine & below);
            stage[2 * slot] = ra;
            stage[2 * slot + 1] = rb;
        }
        __builtin_amdgcn_wave_barrier();
        fwd_eval_lds16<false, 3>(stage, 0, n0, px, pyA, pyB, P.dmax, acc);
        fwd_eval_lds16<false, 1>(stage, n0, n0 + n1, px, pyA, pyB, P.dmax, acc);
        fwd_eval_lds16<false, 2>(stage, n0 + n1, n0 + n1 + n2, px, pyA, pyB, P.dmax, acc);
        if (BOUNDED) fwd_eval_lds16<true, 3>(stage, n0 + n1 + n2, n0 + n1 + n2 + n3, px, pyA, pyB, P.dmax, acc);
    };
    // one-level search of this sub-tile over candidate segments (the large class; every class when the tile's list overflowed)
    auto search = [&](bool large_only) {
        const uint4 *__restrict__ bbox = V.bbox;
        const unsigned *__restrict__ cs = V.cell_start;
        const int wtx = sx0 >> SUBX_SHIFT, wty = (sy0 - P.row0) >> SUBY_SHIFT;
        const int rx = large_only ? 0 : (int)V.hdr[8], ry = large_only ? 0 : (int)V.hdr[9];
        int nseg = 0;
        unsigned sbeg = 0, send = 0;
        if (rx > 0) {
            const int cx0 = max(sx0 - rx, 0) >> CELL_SHIFT, cx1 = min((sx1 + rx) >> CELL_SHIFT, P.ncx - 1);
            const int cy0 = max(sy0 - ry, 0) >> CELL_SHIFT, cy1 = min((sy1 + ry) >> CELL_SHIFT, P.ncy - 1);
            nseg = cy1 - cy0 + 1;
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
        const unsigned len = send - sbeg;
        unsigned pin = len;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = (unsigned)__shfl_up((int)pin, o);
            if (lane >= o) pin += v;
        }
        const unsigned pex = pin - len;
        const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)pin, nseg - 1);
        const unsigned nchunks = (total + 63u) >> 6;
        int rseg = 0;
        for (unsigned c = 0; c < nchunks; ++c) {
            const unsigned j = fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin);
            uint4 bb = make_uint4(0x7fffu, 0x7fffu, 0u, 0u);
            uint2 bs = make_uint2(0u, 0u);
            if (j != 0xffffffffu) {
                bb = bbox[2 * (size_t)j];
                bs = *reinterpret_cast<const uint2 *>(bbox + 2 * (size_t)j + 1);
            }
            const int c0 = (int)(bb.x & 0x7fffu), c1 = (int)(bb.x >> 16);
            const int r0 = (int)(bb.y & 0x7fffu), r1 = (int)(bb.y >> 16);
            bool hit = (c0 <= sx1) & (c1 >= sx0) & (r0 <= sy1) & (r1 >= sy0);
            if (bb.y & 0x8000u) {
                const unsigned t = (unsigned)(wty - ((r0 - P.row0) >> SUBY_SHIFT)) & 7u, sh = (t & 3u) * 8u;
                const unsigned lo = t < 4u ? bb.z : bs.x, hi = t < 4u ? bb.w : bs.y;
                const int txr = wtx - (c0 >> SUBX_SHIFT);
                hit &= (txr + 1 >= (int)((lo >> sh) & 0xffu)) & (txr <= (int)((hi >> sh) & 0xffu));
            }
            const bool needs = BOUNDED && (bb.x & 0x8000u) != 0u;
            const int cls = !hit ? 4 : needs ? 3 : r1 < sy0 + 8 ? 1 : r0 >= sy0 + 8 ? 2 : 0;
            float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
            if (hit) {
                ra = rec[2 * (size_t)j];
                rb = rec[2 * (size_t)j + 1];
            }
            stage_eval(cls, ra, rb);
        }
    };

    const unsigned cnt = (unsigned)__builtin_amdgcn_readfirstlane((int)V.tl_cursor[(size_t)tt * TL_STRIDE]);
    if (cnt > (unsigned)P.tl_cap) {
        search(false);
    } else {
        const uint2 *__restrict__ ent = V.tl_entries + (size_t)tt * (size_t)P.tl_cap;
        // the sub-tile's quadrants: columns 2 sx, 2 sx + 1 of quadrant rows 2 sy (upper eight pixel rows) and 2 sy + 1 (lower)
        const unsigned up = 0x3u << (2 * (wv & 1) + 8 * (wv >> 1)), lo = up << 4;
        const uint2 none = make_uint2(0u, 0u);
        auto classify = [&](const uint2 e) {
            const bool u = (e.y & up) != 0u, l = (e.y & lo) != 0u;
            return !(u || l) ? 4 : (BOUNDED && (e.x >> 31)) ? 3 : !l ? 1 : !u ? 2 : 0;
        };
        unsigned q = 0;
        uint2 e = (unsigned)lane < cnt ? ent[lane] : none;
        int cls = classify(e);
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
        if (cls < 4) {
            const float4 *src = rec + 2 * (size_t)(e.x & 0x7fffffffu);
            ra = src[0];
            rb = src[1];
        }
        uint2 ne = 64u + (unsigned)lane < cnt ? ent[64 + lane] : none;
        while (q < cnt) {
            // (the next chunk's records are requested before this chunk is evaluated)
            const int ncls = classify(ne);
            float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nb = na;
            if (ncls < 4) {
                const float4 *src = rec + 2 * (size_t)(ne.x & 0x7fffffffu);
                na = src[0];
                nb = src[1];
            }
            q += 64u;
            const uint2 nne = q + 64u + (unsigned)lane < cnt ? ent[q + 64u + lane] : none;
            stage_eval(cls, ra, rb);
            cls = ncls; ra = na; rb = nb; ne = nne;
        }
        const unsigned nlarge = V.cell_start[P.ncells + 1] - V.cell_start[P.ncells];
        if (__builtin_amdgcn_readfirstlane((int)nlarge) != 0) search(true);
    }
    if (X < P.w) {
        fwd_store_px(P, img, X, Y, acc[0].x, acc[1].x, acc[2].x);
        fwd_store_px(P, img, X, Y + 4, acc[0].y, acc[1].y, acc[2].y);
        fwd_store_px(P, img, X, Y + 8, acc[3].x, acc[4].x, acc[5].x);
        fwd_store_px(P, img, X, Y + 12, acc[3].y, acc[4].y, acc[5].y);
    }
}

// Small images (fewer sub-tiles than the chip has wave slots, e.g. the 192x192 training crops of
// BASELINE config 5): a workgroup = ONE sub-tile, its candidate chunks dealt to all `blockDim/64` waves,
// partial sums combined through LDS.  Parallelism comes from the Gaussian list instead of from pixels.
template <bool BOUNDED>
__global__ __launch_bounds__(1024) void k_render_fwd_split(Params P, PlanView V, float *__restrict__ img, int subs_x)
{
    __shared__ float s_part[16][6][64];
    __shared__ float4 s_stage[16][STAGE_F4];
    const unsigned t = xcd_swizzle(blockIdx.x, gridDim.x);
    const int sx0 = (int)(t % (unsigned)subs_x) * SUBX, sy0 = P.row0 + (int)(t / (unsigned)subs_x) * SUBY;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    v2f ar = {0.f, 0.f}, ag = {0.f, 0.f}, ab = {0.f, 0.f};
    fwd_tile<BOUNDED>(P, V, sx0, sy0, lane, (unsigned)wv, (unsigned)nw, s_stage[wv], ar, ag, ab);
    if (wv > 0) {
        s_part[wv][0][lane] = ar.x; s_part[wv][1][lane] = ar.y;
        s_part[wv][2][lane] = ag.x; s_part[wv][3][lane] = ag.y;
        s_part[wv][4][lane] = ab.x; s_part[wv][5][lane] = ab.y;
    }
    __syncthreads();
    if (wv == 0) {
        for (int k = 1; k < nw; ++k) {
            ar.x += s_part[k][0][lane]; ar.y += s_part[k][1][lane];
            ag.x += s_part[k][2][lane]; ag.y += s_part[k][3][lane];
            ab.x += s_part[k][4][lane]; ab.y += s_part[k][5][lane];
        }
        fwd_store(P, V, img, sx0, sy0, lane, ar, ag, ab);
    }
}

// ---------------------------------------------------------------------------------------------------
// backward: one wave64 per Gaussian (cell order, so neighbouring waves read neighbouring pixels)
// ---------------------------------------------------------------------------------------------------
// Sweep the pixel window [c0,c0+bw) x [r0,r1] of one Gaussian with a wave.  Lanes are laid LX = 16/32/64
// wide (the narrowest that covers bw, a template parameter so all the lane geometry is constant) and
// 64/LX rows deep; a lane keeps ONE column (u = dx/sx is a lane constant) and handles TWO rows per trip, so
// the per-pixel arithmetic is 2-wide packed fp32.  Because u is constant per lane only three sums over
// rows are accumulated per pixel column,
//     M0 = sum q,  N1 = sum q*B,  N2 = sum q*B^2,      q = v * <grad, colour>,  B = dy/sy - rho u,
// and expanded at the end of the column (see below).
// The dy/sy values of a 64-row block are staged in LDS (256 B per wave); full trips carry no masks or
// address clamps, the ragged last trip is peeled.
// acc[] = {qA, qB, quA, qvB, qAB, Cr, Cg, Cb} (per lane, summed over the wave by the caller).
struct BwdRow {
    v2f m1, m2, k01;         // moments N1, N2 (row pair); colour sums r, g of the first row
    float ka2, kb0, kb1, kb2;  // colour sums: b of the first row, r g b of the second
};

struct Grad6 {  // the three gradient channels of the two pixels (rows Y, Y+RPI) a lane owns in one trip
    float a0, a1, a2, b0, b1, b2;
};

typedef unsigned u3v __attribute__((ext_vector_type(3)));

// Two 12-byte pixels through a raw buffer resource: address = base(SGPR x4) + per-lane byte offset (one VGPR per
// row of the pair, constant over the sweep) + ONE running row offset (SGPR), so a trip spends no VALU instruction and
// a single scalar add on addressing, and reads past the end of the slab return 0 instead of faulting.  (An instruction
// added to a trip of ANY kind, scalar or vector, costs 0.35 us at config 2: the trips are the wave's dependent chain, and
// that chain at seven waves per SIMD is the run time -- DESIGN.md 3c (c), (d).)
__device__ __forceinline__ Grad6 bwd_load(__amdgpu_buffer_rsrc_t rsrc, int voff, int voff_b, int soff_a)
{
    const int soff_b = soff_a;
    const u3v a = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff, soff_a, 0);
    const u3v b = __builtin_amdgcn_raw_buffer_load_b96(rsrc, voff_b, soff_b, 0);
    Grad6 g;
    g.a0 = __uint_as_float(a.x); g.a1 = __uint_as_float(a.y); g.a2 = __uint_as_float(a.z);
    g.b0 = __uint_as_float(b.x); g.b1 = __uint_as_float(b.y); g.b2 = __uint_as_float(b.z);
    return g;
}

template <bool TEST, bool TAIL>
__device__ __forceinline__ void bwd_trip(BwdRow &R, const Grad6 g, v2f dyn, v2f dyraw, bool ok1, bool ok2, float K0,
                                         float nK1, float rho_u, float cr, float cg, float cb, float dmax)
{
    // With u = dx/sx, v = dy/sy and B = v - rho u (the residual of v about its conditional mean given u) the
    // quadratic form completes to  u^2 - 2 rho u v + v^2 = (1-rho^2) u^2 + B^2,  so the exponent is
    //   log2(e) w1 (...) = K0 - K1 B^2,   K0 = -log2(e)/2 u^2 (lane constant),  K1 = log2(e)/2 / (1-rho^2),
    // and the same B feeds the gradient moments: nothing here cancels as |rho| -> 1.
    const v2f Bv = dyn - rho_u;
    const v2f pw = (Bv * nK1) * Bv + K0;
    v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
    if (TEST || TAIL) {
        v.x = ((!TAIL || ok1) && (!TEST || fabsf(dyraw.x) <= dmax)) ? v.x : 0.f;
        v.y = ((!TAIL || ok2) && (!TEST || fabsf(dyraw.y) <= dmax)) ? v.y : 0.f;
    }
    // six consecutive-in-memory floats per pixel pair are used as they land: no register shuffling
    const v2f gp = {fmaf(g.a2, cb, fmaf(g.a1, cg, g.a0 * cr)), fmaf(g.b2, cb, fmaf(g.b1, cg, g.b0 * cr))};  // gs.cu:150
    const v2f q = gp * v, qB = q * Bv;
    // (M0 = sum q is not accumulated: it is <colour, colour sums>, formed once per column strip)
    R.m1 += qB;
    R.m2 += qB * Bv;
    // Only {a0, a1} is an aligned register pair as the two 12-byte loads land; the other four take scalar FMAs
    // (pairing them up costs five v_mov per trip -- more than the two packed operations save).
    R.k01 += (v2f){g.a0, g.a1} * v.x;
    R.ka2 = fmaf(g.a2, v.x, R.ka2);
    R.kb0 = fmaf(g.b0, v.y, R.kb0);
    R.kb1 = fmaf(g.b1, v.y, R.kb1);
    R.kb2 = fmaf(g.b2, v.y, R.kb2);
}

template <bool TEST, int LXLOG, bool UNROLL>
__device__ __forceinline__ void bwd_sweep(int c0, int bw, int r0, int r1, int lane, const Params &P,
                                          const float *__restrict__ pxt, const float *__restrict__ pyt,
                                          const float *__restrict__ grad, float x, float y, float cr, float cg,
                                          float cb, float cinv, float rho, float kappa, float isx, float isy,
                                          float *spy, float (&acc)[8])
{
    constexpr int LX = 1 << LXLOG, RPI = 64 >> LXLOG;
    constexpr float HALF_LOG2E = 0.72134752044448170368f;
    const int col = lane & (LX - 1), rsub = lane >> LXLOG;
    const unsigned pitchb = (unsigned)P.w * 12u;   // bytes per gradient row (< 2^19); all offsets below are unsigned 32 x 32 -> 64
    const float nK1 = -HALF_LOG2E * cinv;
    // issued together with the px load below: one round trip for both tables instead of two dependent ones
    const float py_first = pyt[min(r0 + lane, r1)];
    for (int strip = 0; strip < bw; strip += 64) {
        const int cc = strip + col;
        const int X = c0 + min(cc, bw - 1);
        const float dx = pxt[X] - x;
        // lanes outside the window (or, with TEST, outside the dmax box in x) are switched off through K0:
        // the exponent becomes -inf, v = 0 exactly, and every product with it is 0
        const bool inx = cc < bw && (!TEST || fabsf(dx) <= P.dmax);
        const float u = dx * isx, rho_u = rho * u;
        const float K0 = inx ? -HALF_LOG2E * u * u : -INFINITY;
        BwdRow R;
        R.m1 = R.m2 = R.k01 = (v2f){0.f, 0.f};
        R.ka2 = R.kb0 = R.kb1 = R.kb2 = 0.f;
        const int voff = (int)((unsigned)X * 12u + (unsigned)rsub * pitchb);
        const int halfb = (int)((unsigned)RPI * pitchb);
        for (int rb = r0; rb <= r1; rb += 64) {
            const int rend = min(r1, rb + 63);
            __builtin_amdgcn_wave_barrier();
            {   // per-row values of the block in LDS: v = dy/sy, and (TEST only) the raw dy for the exact box test
                const float dyr = (rb == r0 ? py_first : pyt[min(rb + lane, r1)]) - y;
                spy[lane] = dyr * isy;
                if (TEST) spy[64 + lane] = dyr;
            }
            __builtin_amdgcn_wave_barrier();
            const float *sp = spy + rsub;
            // buffer resource over the slab from row `rb` on (offsets stay far below 2^31 within a 64-row block)
            const char *blk = reinterpret_cast<const char *>(grad) + (unsigned long long)(unsigned)(rb - P.row0) * pitchb;
            const unsigned long long left = (unsigned long long)(unsigned)(P.row1 - rb) * pitchb;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(blk), 0, (int)(left < 0x7fffffffu ? left : 0x7fffffffu), 0x00020000);
            int soff = 0;
            // trip counts up front: the loops below count down (one scalar add + compare + branch per iteration)
            constexpr int TRIP_SHIFT = 7 - LXLOG;                 // log2(rows per trip) = log2(2 RPI)
            const int nrows = rend - rb + 1, ntrip = nrows >> TRIP_SHIFT;
            const int voff_b = voff + halfb;
            // Lanes outside the window sit the trips out (exec mask): the backward is co-limited by the CU's
            // vector-memory pipe (two 768-byte loads per trip and wave, four SIMDs behind one L1), and idle
            // lanes would fetch gradient pixels only to multiply them by zero.
            if (inx) {
            // UNROLL: two trips per iteration, four gradient loads in flight before the first is consumed.  Pays
            // for windows of many trips (x8 and up); costs 18 VGPRs = two waves per SIMD, which small windows
            // (x4, 6 trips) need more: the host picks the instantiation (gsasr_splat_backward).
            int t = ntrip;
            for (; UNROLL && t >= 2; t -= 2, soff += 4 * halfb, sp += 4 * RPI) {
                const Grad6 g0 = bwd_load(rsrc, voff, voff_b, soff);
                const Grad6 g1 = bwd_load(rsrc, voff, voff_b, soff + 2 * halfb);
                const v2f n0 = {sp[0], sp[RPI]}, n1 = {sp[2 * RPI], sp[3 * RPI]};
                const v2f w0 = TEST ? (v2f){sp[64], sp[64 + RPI]} : n0, w1 = TEST ? (v2f){sp[64 + 2 * RPI], sp[64 + 3 * RPI]} : n1;
                bwd_trip<TEST, false>(R, g0, n0, w0, true, true, K0, nK1, rho_u, cr, cg, cb, P.dmax);
                bwd_trip<TEST, false>(R, g1, n1, w1, true, true, K0, nK1, rho_u, cr, cg, cb, P.dmax);
            }
            if (!UNROLL && t > 0) {
                // The plain loop runs ONE TRIP AHEAD: the two loads of trip k+1 are in flight while trip k is summed (two
                // register sets used alternately: no copies).  Left to the compiler every trip was a dependent round trip
                // -- issue, wait, sum -- and a wave's life at x4 is six of them: -5% at config 2, -4% on the config-5 crops,
                // at 71 VGPRs (seven waves per SIMD kept).  Two trips ahead spills (72-VGPR budget): +8%; the same rotation
                // in the unrolled instantiation: no gain (profiles/r03_bwd_experiments.txt).
#define GSASR_TRIP(G) { const v2f n0 = {sp[0], sp[RPI]}; const v2f w0 = TEST ? (v2f){sp[64], sp[64 + RPI]} : n0; \
                        bwd_trip<TEST, false>(R, G, n0, w0, true, true, K0, nK1, rho_u, cr, cg, cb, P.dmax); sp += 2 * RPI; }
                Grad6 ga = bwd_load(rsrc, voff, voff_b, soff);
                soff += 2 * halfb;
                for (; t >= 3; t -= 2) {
                    const Grad6 gb = bwd_load(rsrc, voff, voff_b, soff);
                    soff += 2 * halfb;
                    GSASR_TRIP(ga)
                    ga = bwd_load(rsrc, voff, voff_b, soff);
                    soff += 2 * halfb;
                    GSASR_TRIP(gb)
                }
                if (t == 2) {
                    const Grad6 gb = bwd_load(rsrc, voff, voff_b, soff);
                    soff += 2 * halfb;
                    GSASR_TRIP(ga)
                    GSASR_TRIP(gb)
                } else {
                    GSASR_TRIP(ga)
                }
#undef GSASR_TRIP
                t = 0;
            }
            for (; t > 0; --t, soff += 2 * halfb, sp += 2 * RPI) {   // (odd trip of the unrolled instantiation)
                const v2f n0 = {sp[0], sp[RPI]};
                const v2f w0 = TEST ? (v2f){sp[64], sp[64 + RPI]} : n0;
                bwd_trip<TEST, false>(R, bwd_load(rsrc, voff, voff_b, soff), n0, w0, true, true, K0, nK1, rho_u, cr,
                                      cg, cb, P.dmax);
            }
            if (nrows & ((1 << TRIP_SHIFT) - 1)) {  // ragged last trip: rows past the window are masked (reads past the slab give 0)
                const int Yb = rb + (ntrip << TRIP_SHIFT);
                const int Ya = Yb + rsub, Yc = Ya + RPI;
                const int ia = min(Ya, rend) - rb, ic = min(Yc, rend) - rb;
                const v2f n0 = {spy[ia], spy[ic]};
                const v2f w0 = TEST ? (v2f){spy[64 + ia], spy[64 + ic]} : n0;
                bwd_trip<TEST, true>(R, bwd_load(rsrc, voff, voff_b, soff), n0, w0, Ya <= rend, Yc <= rend, K0, nK1,
                                     rho_u, cr, cg, cb, P.dmax);
            }
            }
        }
        // Expand the column's three sums M0 = sum q, N1 = sum q B, N2 = sum q B^2 (u = dx/sx is a lane constant,
        // A = u - rho v = u kappa - rho B, v = B + rho u):  sum qA, sum qB, sum q u A, sum q v B, sum q A B.
        // Every difference is formed between quantities of its own size, so nothing cancels as |rho| -> 1
        // (the plain monomial moments sum q dx^2, q dx dy, q dy^2 lose 1/(1-rho) digits there).
        const float Kr = R.k01.x + R.kb0, Kg = R.k01.y + R.kb1, Kb = R.ka2 + R.kb2;
        const float M0 = fmaf(Kb, cb, fmaf(Kg, cg, Kr * cr)), N1 = R.m1.x + R.m1.y, N2 = R.m2.x + R.m2.y;
        // (switched-off lanes have M0 = N1 = N2 = 0, but their u is meaningless: use 0)
        const float ue = inx ? u : 0.f, uk = ue * kappa;
        const float sA = uk * M0 - rho * N1;
        const float e[8] = {sA, N1, ue * sA, N2 + rho * ue * N1, uk * N1 - rho * N2,
                            Kr, Kg, Kb};
        // the first (usually only) 64-column strip assigns, so acc[] is not live during its sweep
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = strip == 0 ? e[k] : acc[k] + e[k];
    }
}

// lane i reads lane i+N of its row of 16 (0 past the row end)
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}

// Sum eight per-lane values over the wave through LDS: lanes park their 8 partials ([8][64] floats per wave),
// lane l then adds the 8 consecutive partials {l&7} of value {l>>3} (two ds_read_b128) and three butterfly
// steps (DPP) finish inside each 8-lane group.  Afterwards lane 8k holds the total of value k.  ~14 VALU
// instructions instead of ~45 for a register-only exchange network; the LDS pipe is otherwise idle here.
__device__ __forceinline__ float wave_sum8(const float (&a)[8], int lane, float *red)
{
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < 8; ++k) red[k * 64 + lane] = a[k];
    __builtin_amdgcn_wave_barrier();
    const float4 u = *reinterpret_cast<const float4 *>(red + lane * 8);
    const float4 v = *reinterpret_cast<const float4 *>(red + lane * 8 + 4);
    float d = ((u.x + u.y) + (u.z + u.w)) + ((v.x + v.y) + (v.z + v.w));
    // lane 8k += lanes 8k+4, then +2, then +1, as DPP row shifts folded into the adds (a __shfl_xor is a
    // ds_bpermute round trip plus five address instructions each)
    d += dpp_row_shl<4>(d);
    d += dpp_row_shl<2>(d);
    d += dpp_row_shl<1>(d);
    return d;   // valid in lanes 8k only
}

// Epilogue of one Gaussian (gs.cu:139-146).  With u = dx/sx, v = dy/sy, A = u - rho v, B = v - rho u each gradient
// component is ONE of the eight sums {qA, qB, quA, qvB, qAB, Cr, Cg, Cb} times a Gaussian constant:
//   d/dx = c/sx qA,  d/dy = c/sy qB,  d/dsx = c/sx quA,  d/dsy = c/sy qvB,  d/drho = c^2 qAB,  c = 1/(1-rho^2),
// and the colour gradients are the sums themselves.  The constants are applied to the per-lane partials
// (bwd_scale, five full-wave multiplies by a scalar) so that after the wave reduction lane 8k simply holds
// output k of {x, y | sx, sy, rho | r, g, b} and stores it through a per-lane pointer (bwd_write).
__device__ __forceinline__ void bwd_scale(float (&a)[8], float c, float isx, float isy)
{
    const float fx = c * isx, fy = c * isy;
    a[0] *= fx; a[1] *= fy; a[2] *= fx; a[3] *= fy; a[4] *= c * c;
}

__device__ __forceinline__ void bwd_write(float v, int lane, const Params &P, unsigned i, float *__restrict__ g_sigmas,
                                          float *__restrict__ g_coords, float *__restrict__ g_colors)
{
    if (lane & 7) return;
    const int k = lane >> 3;
    // one store (or atomic) through a per-lane pointer: the three arrays' bases are wave-uniform, pre-biased so that each
    // is indexed by k, and selected per lane -- three exec-masked branches cost twice the instructions
    float *pc = g_coords + (size_t)i * stride2(P), *ps = g_sigmas + (size_t)i * stride3(P) - 2,
          *pk = g_colors + (size_t)i * stride3(P) - 5;
    float *dst = (k < 2 ? pc : (k < 5 ? ps : pk)) + k;
    if (P.flags & GSASR_FLAG_OVERWRITE_GRADS) *dst = v;
    else atomicAdd(dst, v);   // fire-and-forget: the wave must not end on a load-add-store round trip
}

typedef unsigned u2v __attribute__((ext_vector_type(2)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u8v __attribute__((ext_vector_type(8)));

// Everything the sweep needs about Gaussian j, fetched by the SCALAR unit in one batch (one round trip).
// Left to the compiler these are vector loads + v_readfirstlane (the kernel also stores to the workspace, so
// it will not use the non-coherent scalar cache) issued as three dependent round trips, which was most of a
// wave's life.  The plan was written by an earlier kernel, so the scalar cache is coherent for it.
struct BwdRec {
    u8v bb;    // both bbox words: {c0|test|c1, r0|r1, spans.. | spans.., padded rows r0|r1 of the sweep, -}
    u8v rec;   // {x, y, IX, NR | IY, r, g, b}
    u8v fin;   // {c, kappa, rho, 1/sx | 1/sy, px-table offset, sample, index}
};

__device__ __forceinline__ void bwd_fetch(const PlanView &V, unsigned j, BwdRec &R)
{
    const uint4 *pb = V.bbox + 2 * (size_t)j;
    const float4 *pr = V.rec + 2 * (size_t)j, *pf = V.fin + 2 * (size_t)j;
    asm volatile("s_load_dwordx8 %0, %3, 0x0\n\t"
                 "s_load_dwordx8 %1, %4, 0x0\n\t"
                 "s_load_dwordx8 %2, %5, 0x0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(R.bb), "=&s"(R.rec), "=&s"(R.fin)
                 : "s"(pb), "s"(pr), "s"(pf)
                 : "memory");
}

// the same plus the two class boundaries cell_start[ncells], cell_start[ncells+1]
__device__ __forceinline__ void bwd_fetch_first(const PlanView &V, const unsigned *bounds, unsigned j, BwdRec &R, u2v &lim)
{
    const uint4 *pb = V.bbox + 2 * (size_t)j;
    const float4 *pr = V.rec + 2 * (size_t)j, *pf = V.fin + 2 * (size_t)j;
    asm volatile("s_load_dwordx2 %3, %7, 0x0\n\t"
                 "s_load_dwordx8 %0, %4, 0x0\n\t"
                 "s_load_dwordx8 %1, %5, 0x0\n\t"
                 "s_load_dwordx8 %2, %6, 0x0\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(R.bb), "=&s"(R.rec), "=&s"(R.fin), "=&s"(lim)
                 : "s"(pb), "s"(pr), "s"(pf), "s"(bounds)
                 : "memory");
}

template <bool BOUNDED, bool UNROLL>
__device__ __forceinline__ void bwd_item(unsigned j, const BwdRec &G, int chunk, bool atomic, int lane, const Params &P,
                                         const PlanView &V, const float *__restrict__ grad, float *spy, float *red,
                                         float *__restrict__ g_sigmas, float *__restrict__ g_coords,
                                         float *__restrict__ g_colors)
{
    const unsigned bbx = G.bb[0];
    const int c0 = (int)(bbx & 0x7fffu), c1 = (int)(bbx >> 16);
    if (c0 > c1) return;  // dead class (handled by the caller)
    int r0, r1;
    bool empty = false;
    if (chunk >= 0) {  // (row chunks of a large Gaussian must not overlap: they split the window's own rows)
        r0 = (int)(G.bb[1] & 0x7fffu);
        r1 = (int)(G.bb[1] >> 16);
        const int rpc = (r1 - r0 + NCH) / NCH;
        r0 = r0 + chunk * rpc;
        r1 = min(r1, r0 + rpc - 1);
        empty = r0 > r1;   // still counted as a finished chunk below
    } else {           // the plan's padded row range (whole trips; k_bin)
        r0 = (int)(G.bb[6] & 0xffffu);
        r1 = (int)(G.bb[6] >> 16);
    }
    const float x = __uint_as_float(G.rec[0]), y = __uint_as_float(G.rec[1]);
    const float cr = __uint_as_float(G.rec[5]), cg = __uint_as_float(G.rec[6]), cb = __uint_as_float(G.rec[7]);
    const float4 fa = make_float4(__uint_as_float(G.fin[0]), __uint_as_float(G.fin[1]), __uint_as_float(G.fin[2]),
                                  __uint_as_float(G.fin[3]));   // {c, kappa, rho, 1/sx}
    const float4 fb = make_float4(__uint_as_float(G.fin[4]), 0.f, 0.f, 0.f);   // {1/sy, ..}
    float a[8];
    const int bw = c1 - c0 + 1;
    const bool test = BOUNDED && (bbx & 0x8000u);
    float d = 0.f;
    if (!empty) {
#define GSASR_SWEEP(T, L) \
    bwd_sweep<T, L, UNROLL>(c0, bw, r0, r1, lane, P, V.px + G.fin[5], V.py, grad, x, y, cr, cg, cb, fa.x, fa.z, fa.y, fa.w, fb.x, spy, a)
        if (bw <= 16) { if (test) GSASR_SWEEP(true, 4); else GSASR_SWEEP(false, 4); }
        else if (bw <= 32) { if (test) GSASR_SWEEP(true, 5); else GSASR_SWEEP(false, 5); }
        else { if (test) GSASR_SWEEP(true, 6); else GSASR_SWEEP(false, 6); }
#undef GSASR_SWEEP
        bwd_scale(a, fa.x, fa.w, fb.x);
        d = wave_sum8(a, lane, red);   // lane 8k now holds gradient component k
    }
    if (atomic) {
        // Large class: the row chunks add into sums[] and count themselves; the wave that finishes the
        // last chunk takes the totals (re-arming the accumulators for the next backward) and writes the
        // gradient, so no separate finalize pass exists.
        if (!empty && (lane & 7) == 0) atomicAdd(V.sums + 8 * (size_t)j + (lane >> 3), d);
        __threadfence();
        unsigned prev = 0;
        if (lane == 0) prev = atomicAdd(&V.done[j], 1u);
        prev = (unsigned)__builtin_amdgcn_readfirstlane((int)prev);
        if (prev != (unsigned)(NCH - 1)) return;
        __threadfence();
        if ((lane & 7) == 0) d = atomicExch(V.sums + 8 * (size_t)j + (lane >> 3), 0.f);
        if (lane == 0) V.done[j] = 0u;
    }
    bwd_write(d, lane, P, G.fin[7], g_sigmas, g_coords, g_colors);
}

// (occupancy targets: the unrolled sweep fits 6 waves per SIMD at the price of five spilled dwords, -2.7% at config 4;
// forcing the plain sweep to 8 costs more in spills than it gains)
template <bool BOUNDED, bool UNROLL>
__global__ __launch_bounds__(64 * BWD_WAVES) __attribute__((amdgpu_waves_per_eu(UNROLL ? BWD_UNROLL_OCC : BWD_OCC))) void k_render_bwd(Params P, PlanView V, const float *__restrict__ grad,
                                                    float *__restrict__ g_sigmas, float *__restrict__ g_coords,
                                                    float *__restrict__ g_colors)
{
    const int lane = threadIdx.x & 63;
    // XCD-aware order (block b runs on XCD b%8): each XCD sweeps a contiguous run of the cell-ordered
    // Gaussians, i.e. one band of the image, so the grad_img rows it re-reads stay in ITS 4 MiB L2
    const unsigned nb = gridDim.x, b = blockIdx.x;
    const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u;
    const unsigned t = xcd * q + min(xcd, r) + (b >> 3);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned gw = t * (unsigned)BWD_WAVES + (unsigned)wv;
    const unsigned nwaves = nb * (unsigned)BWD_WAVES;
    __shared__ float s_py[BWD_WAVES][128];  // per wave: v = dy/sy of a 64-row block, then (TEST) the raw dy
    __shared__ __attribute__((aligned(16))) float s_red[BWD_WAVES][512];
    float *spy = s_py[wv], *red = s_red[wv];
    // one Gaussian per wave, dispatched by the hardware (a persistent-workgroup variant with a static
    // partition was measured 13% slower at config 2 and 60% slower at config 3: load imbalance)
    // (two or four Gaussians per wave, one after the other, measured the same: wave launch is not the cost)
    BwdRec G;
    u2v lim;
    bwd_fetch_first(V, V.cell_start + P.ncells, min(gw, (unsigned)P.s - 1u), G, lim);  // speculative: class checked below
    const unsigned large_beg = lim.x, large_end = lim.y;
    if (gw < large_beg)
        bwd_item<BOUNDED, UNROLL>(gw, G, -1, false, lane, P, V, grad, spy, red, g_sigmas, g_coords, g_colors);
    else if (gw < large_end)
        bwd_item<BOUNDED, UNROLL>(gw, G, 0, true, lane, P, V, grad, spy, red, g_sigmas, g_coords, g_colors);
    else if (gw < (unsigned)P.s && (P.flags & GSASR_FLAG_OVERWRITE_GRADS))   // dead class: the gradient is zero
        bwd_write(0.f, lane, P, G.fin[7], g_sigmas, g_coords, g_colors);
    // remaining row chunks of the large class, spread over all waves
    const unsigned extra = (large_end - large_beg) * (unsigned)(NCH - 1);
    for (unsigned it = gw; it < extra; it += nwaves) {
        const unsigned j = large_beg + it / (unsigned)(NCH - 1);
        const int chunk = 1 + (int)(it % (unsigned)(NCH - 1));
        bwd_fetch(V, j, G);
        bwd_item<BOUNDED, UNROLL>(j, G, chunk, true, lane, P, V, grad, spy, red, g_sigmas, g_coords, g_colors);
    }
}


// ---------------------------------------------------------------------------------------------------
// backward, TILE-stationary (BASELINE.json north_star's shape: a workgroup owns an HR tile, stages its grad_img ONCE
// in LDS and streams the Gaussians binned near it).  Measured against the Gaussian-stationary k_render_bwd above in
// DESIGN.md 3c; the host picks between the two (gsasr_splat_backward).
//
//   tile      32 x 16 px = 8 "quadrants" of 8 x 8 px, one workgroup of two waves per tile; from x8 up (bt_tall) 32 x 32 px =
//             16 quadrants and four waves.  XCD-banded tile order.
//   stage     the tile's gradient (HWC or planar CHW, zero outside the image / the sample / the row band) goes to LDS as
//             packed row pairs {r_a, r_b, g_a, g_b, b_a, b_b} per (column, row pair) of each quadrant, with the px / py
//             table entries of the tile.  Every pixel of grad_img is read once per tile that holds it -- exactly once.
//   level 1   as in the forward (fwd_block): the four waves test the windows of the Gaussians binned within reach of
//             the tile, 64 per wave and chunk, and append the survivors to a list in LDS.  A survivor also appends one
//             ITEM per quadrant its window touches (1..8 or 16), survivor-major, the last one marked.
//   level 2   LANE = ITEM = (Gaussian, quadrant): a lane loads its Gaussian's records once and evaluates it at the 64
//             pixels of its quadrant -- gradients read from LDS (lanes of different quadrants hit disjoint banks), two
//             rows per packed-fp32 operation, columns in the outer loop so that u = dx/sx is constant in the inner one
//             and the same residual-form sums as bwd_sweep apply.  No cross-lane reduction of pixels, no masks: a pixel
//             outside the Gaussian's window adds a term below exp(-tau), a pixel outside the image adds 0 * v.
//             The items of one Gaussian sit in adjacent lanes (chunks are cut at the last marked lane, so a Gaussian never
//             straddles two chunks): three (four) shuffle steps add them up, and the first lane of each run stores the eight raw
//             sums into the Gaussian's slot for THIS tile (PlanView::part) -- plain 32-byte stores, no atomics, no
//             dependence on scheduling.  (Measured on this chip: fp32 global atomics retire ~19 G cache-line requests/s
//             chip-wide and ds_add_f32 ~3 cycles per lane; tools/atomic_rate.hip.  One atomic set per (tile, Gaussian)
//             would be 14 us of atomic traffic at config 2.)
//   gather    k_bwd_gather (or the fused k_prologue_bwd_gather of the step entry points): one thread per Gaussian adds
//             the slots of its window's tiles in order, applies the Gaussian's constants and writes the gradient.
// A Gaussian whose window spans more tiles than it has slots (or the "large" class) adds into PlanView::sums with
// atomics instead; the gather adds those as well.
// ---------------------------------------------------------------------------------------------------
constexpr int BT_W = 32;                            // tile width; its height 1 << HLOG = 16 or 32 is a template parameter (bt_tall)
#ifndef BT_WAVES_N
#define BT_WAVES_N 2
#endif
constexpr int BT_WAVES = BT_WAVES_N;                // waves per workgroup (= per tile)
// candidate chunks per wave and round (level 1) = a template parameter of the kernel: 8 / BT_WAVES (rounds of 512 candidates,
// 19 KB of LDS per tile = four waves per SIMD), or half of that for plans at 32 HR pixels per Gaussian and more -- a tile
// then sees ~300 candidates, rounds of 256 cost it nothing, and 15 KB of LDS + 96 VGPRs put FIVE waves on a SIMD: -8% at
// config 4 (at x4 the smaller rounds cost +5%, at 16 Gaussians per LR pixel +8%: they keep the large ones)
constexpr int BT_THREADS = 64 * BT_WAVES;
constexpr int BT_QSTRIDE = 32 * 8 + 8;              // floats per quadrant block: 32 entries of 8 floats, +8 so that the
                                                    // blocks of the eight quadrants start 8 banks apart
constexpr unsigned BT_WIDE = 0xffu;                 // slot code: window spans more tiles than part_k -> atomics into sums

__device__ __forceinline__ int bt_tile_span(unsigned wx, unsigned wy, int row0, int hlog, int &ntx, int &tx0, int &ty0)
{
    const int c0 = (int)(wx & 0x7fffu), c1 = (int)(wx >> 16), r0 = (int)(wy & 0x7fffu), r1 = (int)(wy >> 16);
    tx0 = c0 >> 5;
    ty0 = (r0 - row0) >> hlog;
    ntx = (c1 >> 5) - tx0 + 1;
    return ntx * (((r1 - row0) >> hlog) - ty0 + 1);
}

// One item: Gaussian j (cell order) at the 64 pixels of one quadrant.  gq = the quadrant's block of staged gradients,
// pxq / pyq = its 8 column / row coordinates.  a[] = raw sums {qA, qB, quA, qvB, qAB, Cr, Cg, Cb} (cf. bwd_sweep).
template <bool TEST>
__device__ __forceinline__ void bt_eval(const PlanView &V, unsigned j, float dm, const float *gq, const float *pxq,
                                        const float *pyq, float (&a)[8])
{
    constexpr float HALF_LOG2E = 0.72134752044448170368f;
    const float4 ra = V.rec[2 * (size_t)j], rb = V.rec[2 * (size_t)j + 1];
    const float4 fa = V.fin[2 * (size_t)j];
    const float isy = V.fin[2 * (size_t)j + 1].x;
    const float x = ra.x, y = ra.y, cr = rb.y, cg = rb.z, cb = rb.w;
    const float cinv = fa.x, kappa = fa.y, rho = fa.z, isx = fa.w;
    // exponent (log2) = -h u^2 - h c B^2 with u = dx/sx, B = dy/sy - rho u, c = 1/(1-rho^2) (bwd_trip); B is carried
    // pre-scaled by sB = sqrt(h c), so that the exponent is K0(u) - B'^2
    const float sB = __builtin_amdgcn_sqrtf(HALF_LOG2E * cinv), inv_sB = __builtin_amdgcn_rcpf(sB);
    const float isyB = isy * sB, rsB = rho * sB;
    v2f vp[4], rt[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const v2f dy = (v2f){pyq[2 * p], pyq[2 * p + 1]} - y;
        vp[p] = dy * isyB;
        if (TEST) rt[p] = (v2f){fabsf(dy.x) <= dm ? 0.f : -INFINITY, fabsf(dy.y) <= dm ? 0.f : -INFINITY};
    }
    float s_uM = 0.f, s_uuM = 0.f, s_N1 = 0.f, s_uN1 = 0.f, s_N2 = 0.f;
    v2f Cr = {0.f, 0.f}, Cg = {0.f, 0.f}, Cb = {0.f, 0.f};
    for (int c = 0; c < 8; ++c) {
        const float dx = pxq[c] - x;
        const float u = dx * isx, ru = rsB * u;
        float K0 = -HALF_LOG2E * u * u;
        if (TEST) K0 = fabsf(dx) <= dm ? K0 : -INFINITY;   // exponent -inf: v = 0 exactly, every product with it is 0
        v2f M0 = {0.f, 0.f}, N1 = {0.f, 0.f}, N2 = {0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float4 d0 = *reinterpret_cast<const float4 *>(gq + (c * 4 + p) * 8);
            const float2 d1 = *reinterpret_cast<const float2 *>(gq + (c * 4 + p) * 8 + 4);
            const v2f Bv = vp[p] - ru;
            v2f pw = K0 - Bv * Bv;
            if (TEST) pw += rt[p];
            const v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
            const v2f gr = {d0.x, d0.y}, gn = {d0.z, d0.w}, gb = {d1.x, d1.y};
            const v2f gp = gb * cb + (gn * cg + gr * cr);   // gs.cu:150
            const v2f qq = gp * v, qB = qq * Bv;
            M0 += qq;
            N1 += qB;
            N2 += qB * Bv;
            Cr += v * gr;
            Cg += v * gn;
            Cb += v * gb;
        }
        // the column's sums, as polynomials in its u (expanded after the last column)
        const float m0 = M0.x + M0.y, n1 = N1.x + N1.y, n2 = N2.x + N2.y;
        const float um = u * m0;
        s_uM += um;
        s_uuM = fmaf(u, um, s_uuM);
        s_N1 += n1;
        s_uN1 = fmaf(u, n1, s_uN1);
        s_N2 += n2;
    }
    // undo the scale of B, then  sum qA = kappa sum(u M0) - rho sum N1  etc.: bwd_sweep's per-column expansion summed
    // over the columns (A = kappa u - rho B, v = B + rho u)
    const float N1t = s_N1 * inv_sB, uN1t = s_uN1 * inv_sB, N2t = s_N2 * inv_sB * inv_sB;
    a[0] = kappa * s_uM - rho * N1t;
    a[1] = N1t;
    a[2] = kappa * s_uuM - rho * uN1t;
    a[3] = N2t + rho * uN1t;
    a[4] = kappa * uN1t - rho * N2t;
    a[5] = Cr.x + Cr.y;
    a[6] = Cg.x + Cg.y;
    a[7] = Cb.x + Cb.y;
}

// (the 32-row tile runs twice the waves per workgroup with half the chunks each: the same rounds, the same waves per CU under
// its 27 KB of LDS)
template <bool BOUNDED, int BT_CHUNKS, int HLOG>
__global__ __launch_bounds__(BT_THREADS << (HLOG - 4)) __attribute__((amdgpu_waves_per_eu((BT_CHUNKS * BT_WAVES << (HLOG - 4)) <= 4 ? 5 : 4, 5))) void k_render_bwd_tile(
    Params P, PlanView V, const float *__restrict__ grad, int tiles_x, int use_atomics)
{
    constexpr int BT_H = 1 << HLOG, NQY = BT_H / 8, NQ = 4 * NQY;   // tile height, quadrant rows, quadrants (8 or 16)
    constexpr int WAVES = BT_WAVES << (HLOG - 4), THREADS = 64 * WAVES;
    constexpr int BT_LIST = WAVES * BT_CHUNKS * 64;     // survivors per round at most (512 / 256)
    __shared__ __attribute__((aligned(16))) float s_g[NQ * BT_QSTRIDE];
    __shared__ float s_px[BT_W], s_py[BT_H];
    __shared__ unsigned s_list[BT_LIST];            // survivor: index in cell order | needs the dmax test << 31
    __shared__ unsigned char s_slot[BT_LIST];       // its slot in part[] for this tile, or BT_WIDE
    __shared__ unsigned short s_items[BT_LIST * NQ]; // item: survivor (9 bits) | quadrant << 9 | last of its survivor << 13
    __shared__ unsigned s_cnt[3];                   // survivors, items of the round; head of the item queue (level 2)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tt = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tx = (int)(tt % (unsigned)tiles_x), ty = (int)(tt / (unsigned)tiles_x);
    const int bx0 = tx * BT_W, by0 = P.row0 + ty * BT_H;
    const int bx1 = min(bx0 + BT_W - 1, P.w - 1), by1 = min(by0 + BT_H - 1, P.row1 - 1);
    const int smp = P.batch > 1 ? by0 / P.slot : 0;
    const Geo g = sample_geo(P, V, smp);

    // ---- stage the tile ---------------------------------------------------------------------------------
    {
        const int ylim = min(P.row1, g.base + g.h);
        const bool chw = (P.flags & GSASR_FLAG_CHW_GRAD) != 0u;
        size_t plane = (size_t)(P.row1 - P.row0) * P.w, org = 0;     // planar: [3, rows, w]; batched [B, 3, grad_rows, w]
        int yoff = P.row0;
        if (chw && P.batch > 1) {
            plane = (size_t)P.grad_rows * P.w;
            org = (size_t)smp * 3 * plane;
            yoff = g.base;
        }
#pragma unroll
        for (int i = tid; i < BT_W * BT_H; i += THREADS) {
            const int row = i >> 5, col = i & 31, X = bx0 + col, Y = by0 + row;
            float r = 0.f, gg = 0.f, b = 0.f;
            if (X < g.w && Y < ylim) {
                if (chw) {
                    const float *q = grad + org + (size_t)(Y - yoff) * P.w + X;
                    r = q[0]; gg = q[plane]; b = q[2 * plane];
                } else {
                    const float *q = grad + ((size_t)(Y - P.row0) * P.w + X) * 3;
                    r = q[0]; gg = q[1]; b = q[2];
                }
            }
            float *e = s_g + ((row >> 3) * 4 + (col >> 3)) * BT_QSTRIDE + (((col & 7) * 4 + ((row & 7) >> 1)) * 8) + (row & 1);
            e[0] = r; e[2] = gg; e[4] = b;
        }
        if (tid < BT_W) s_px[tid] = V.px[g.pxo + min(bx0 + tid, P.w - 1)];
        else if (tid < BT_W + BT_H) s_py[tid - BT_W] = V.py[min(by0 + tid - BT_W, P.h - 1)];
        if (tid < 3) s_cnt[tid] = 0u;
    }

    // ---- segment table of the tile (every wave builds the same one; cf. fwd_block) ------------------------
    const unsigned *__restrict__ cs = V.cell_start;
    const int rx = (int)V.hdr[8], ry = (int)V.hdr[9];
    int nseg = 0;
    unsigned sbeg = 0, send = 0;
    if (rx > 0) {
        const int cx0 = max(bx0 - rx, 0) >> CELL_SHIFT, cx1 = min((bx1 + rx) >> CELL_SHIFT, P.ncx - 1);
        const int cy0 = max(by0 - ry, 0) >> CELL_SHIFT, cy1 = min((by1 + ry) >> CELL_SHIFT, P.ncy - 1);
        nseg = cy1 - cy0 + 1;
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
    const unsigned len = send - sbeg;
    unsigned pin = len;
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = (unsigned)__shfl_up((int)pin, o);
        if (lane >= o) pin += v;
    }
    const unsigned pex = pin - len;
    const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)pin, nseg - 1);
    const unsigned nchunks = (total + 63u) >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int rseg = 0;
    __syncthreads();

    for (unsigned base = 0; base < nchunks; base += (unsigned)(WAVES * BT_CHUNKS)) {
        // ---- level 1: candidates -> survivors + items ----------------------------------------------------
        unsigned cj[BT_CHUNKS];
        uint2 cw[BT_CHUNKS];
#pragma unroll
        for (int k = 0; k < BT_CHUNKS; ++k) {
            const unsigned c = base + (unsigned)wv + (unsigned)(WAVES * k);
            cj[k] = c < nchunks ? fwd_candidate(c, lane, nseg, rseg, sbeg, pex, pin) : 0xffffffffu;
            cw[k] = make_uint2(0x7fffu, 0x7fffu);
            if (cj[k] != 0xffffffffu) cw[k] = V.win[cj[k]];
        }
#pragma unroll
        for (int k = 0; k < BT_CHUNKS; ++k) {
            const int c0 = (int)(cw[k].x & 0x7fffu), c1 = (int)(cw[k].x >> 16);
            const int r0 = (int)(cw[k].y & 0x7fffu), r1 = (int)(cw[k].y >> 16);
            const bool hit = (c0 <= bx1) & (c1 >= bx0) & (r0 <= by1) & (r1 >= by0);
            const unsigned long long m = __ballot(hit);
            if (m == 0ull) continue;
            unsigned at = 0;
            if (lane == 0) at = atomicAdd(&s_cnt[0], (unsigned)__builtin_popcountll(m));
            at = (unsigned)__builtin_amdgcn_readfirstlane((int)at);
            const unsigned pos = at + (unsigned)__builtin_popcountll(m & below);
            // quadrants of the tile the window touches, row by row trimmed to the columns the ellipse reaches (k_bin's
            // per-8-row spans): the corners of the window are empty for every Gaussian, most of it for a correlated one
            const int qx0 = max(c0 - bx0, 0) >> 3, qx1 = min(c1 - bx0, BT_W - 1) >> 3;
            const int qy0 = max(r0 - by0, 0) >> 3, qy1 = min(r1 - by0, BT_H - 1) >> 3;
            int xl[NQY], xh[NQY];
#pragma unroll
            for (int qy = 0; qy < NQY; ++qy) { xl[qy] = 1; xh[qy] = 0; }
            if (hit) {
                // per-8-row spans (qspan) when the window has at most eight such bands; a taller window (x12 and up) still has
                // the forward's per-16-row spans in its window words: both quadrant rows of this tile then share one band
                uint4 qs = make_uint4(0u, 0xffffffffu, 0u, 0xffffffffu);
                const int q0 = (r0 - P.row0) >> 3, b8 = (by0 - P.row0) >> 3;
                const bool fine = ((r1 - P.row0) >> 3) - q0 < 8;
                if (fine) {
                    if (V.qspan) qs = V.qspan[cj[k]];
                } else if (cw[k].y & 0x8000u) {
                    const uint2 *sp = reinterpret_cast<const uint2 *>(V.bbox + 2 * (size_t)cj[k]);
                    const uint2 s0 = sp[1], s1 = sp[2];
                    qs = make_uint4(s0.x, s0.y, s1.x, s1.y);
                }
                const int cu = (c0 >> 3) - (bx0 >> 3);
#pragma unroll
                for (int qy = 0; qy < NQY; ++qy) {
                    const unsigned t = (unsigned)(fine ? b8 - q0 + qy : ((b8 + qy) >> 1) - (q0 >> 1)) & 7u, sh = (t & 3u) * 8u;
                    const int lo = (int)(((t < 4u ? qs.x : qs.z) >> sh) & 0xffu), hi = (int)(((t < 4u ? qs.y : qs.w) >> sh) & 0xffu);
                    if (qy >= qy0 && qy <= qy1) {
                        // (hi = 255 is "as far as the window goes": the default of a window k_bin computed no spans for --
                        // one wider than 255 columns of 8 px among them, whose far tiles would otherwise lose their quadrants)
                        xl[qy] = max(qx0, cu + lo);
                        xh[qy] = hi == 255 ? qx1 : min(qx1, cu + hi);
                    }
                }
            }
            unsigned n_i = 0u;
#pragma unroll
            for (int qy = 0; qy < NQY; ++qy) n_i += (unsigned)max(xh[qy] - xl[qy] + 1, 0);
            unsigned inc = n_i;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = (unsigned)__shfl_up((int)inc, o);
                if (lane >= o) inc += v;
            }
            unsigned ib = 0;
            if (lane == 63) ib = atomicAdd(&s_cnt[1], inc);
            ib = (unsigned)__builtin_amdgcn_readlane((int)ib, 63);
            if (hit) {
                s_list[pos] = cj[k] | ((cw[k].x & 0x8000u) << 16);
                int ntx, wtx0, wty0;
                const int nt = bt_tile_span(cw[k].x, cw[k].y, P.row0, HLOG, ntx, wtx0, wty0);
                const unsigned slot = nt <= P.part_k ? (unsigned)((ty - wty0) * ntx + (tx - wtx0)) : BT_WIDE;
                s_slot[pos] = (unsigned char)slot;
                unsigned off = ib + inc - n_i;
                const unsigned last = off + n_i - 1u;
#pragma unroll
                for (int qy = 0; qy < NQY; ++qy)
                    for (int qx = xl[qy]; qx <= xh[qy]; ++qx, ++off)
                        s_items[off] = (unsigned short)((unsigned)pos | (unsigned)(qy * 4 + qx) << 9 | (off == last ? 0x2000u : 0u));
                if (n_i == 0u && slot != BT_WIDE && !use_atomics) {   // the ellipse misses the tile: its slot is still read
                    float4 *o = reinterpret_cast<float4 *>(V.part + ((size_t)cj[k] * P.part_k + slot) * 8);
                    o[0] = make_float4(0.f, 0.f, 0.f, 0.f);
                    o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        __syncthreads();
        const unsigned nsurv = (unsigned)__builtin_amdgcn_readfirstlane((int)s_cnt[0]);
        const unsigned nitems = (unsigned)__builtin_amdgcn_readfirstlane((int)s_cnt[1]);
        (void)nsurv;
        // ---- level 2: chunks of <= 64 items, cut where a Gaussian's items end, CLAIMED by the waves from one queue ----
        // (a static split of the item list between the waves leaves every wave a ragged last chunk: with ~150 items per wave
        // that is 3-3.5 chunk iterations for 2.4 chunks of work, the largest single loss of this kernel.  A chunk's end
        // depends on its items, so a wave reads the queue head, finds its cut and claims [head, cut] with a compare-and-swap.)
        for (;;) {
            unsigned p0, it = 0u;
            int tlast = 0;
            for (;;) {
                p0 = (unsigned)__builtin_amdgcn_readfirstlane((int)*(volatile unsigned *)&s_cnt[2]);
                if (p0 >= nitems) break;
                const unsigned idx = p0 + (unsigned)lane;
                it = idx < nitems ? s_items[idx] : 0u;
                const unsigned long long tails = __ballot(idx < nitems && (it & 0x2000u));
                tlast = 63 - __builtin_clzll(tails);                  // (the list ends on a marked item: tails != 0)
                unsigned got = 0u;
                if (lane == 0) got = atomicCAS(&s_cnt[2], p0, p0 + (unsigned)tlast + 1u);
                if ((unsigned)__builtin_amdgcn_readfirstlane((int)got) == p0) break;
            }
            if (p0 >= nitems) break;
            const bool valid = lane <= tlast;
            const unsigned lidx = it & 0x1ffu, q = (it >> 9) & 15u;
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            unsigned j = 0u;
            unsigned e = 0u;
            if (valid) {
                e = s_list[lidx];
                j = e & 0x7fffffffu;
            }
            // (the dmax test costs an instruction per pixel pair: only chunks holding a Gaussian that needs it pay)
            if (BOUNDED && __ballot(valid && (e >> 31)) != 0ull) {
                if (valid) bt_eval<true>(V, j, (e >> 31) ? P.dmax : INFINITY, s_g + q * BT_QSTRIDE, s_px + (q & 3u) * 8u, s_py + (q >> 2) * 8u, a);
            } else {
                if (valid) bt_eval<false>(V, j, INFINITY, s_g + q * BT_QSTRIDE, s_px + (q & 3u) * 8u, s_py + (q >> 2) * 8u, a);
            }
            // add up the items of each Gaussian (adjacent lanes, at most NQ): three or four shuffle steps; its first lane gets the total
            const unsigned key = valid ? lidx : 0xffffu;
#pragma unroll
            for (int o = 1; o < NQ; o <<= 1) {
                const bool same = (unsigned)__shfl_down((int)key, o) == key && lane + o < 64;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float other = __shfl_down(a[k], o);
                    a[k] += same ? other : 0.f;
                }
            }
            // (the shuffle must run with every lane enabled: a lane that has been switched off by a short-circuit
            // supplies 0 to its neighbour)
            const unsigned prev = (unsigned)__shfl_up((int)key, 1);
            const bool head = valid && (lane == 0 || prev != key);
            if (head) {
                const unsigned slot = s_slot[lidx];
                if (slot != BT_WIDE && !use_atomics) {
                    float4 *o = reinterpret_cast<float4 *>(V.part + ((size_t)j * P.part_k + slot) * 8);
                    o[0] = make_float4(a[0], a[1], a[2], a[3]);
                    o[1] = make_float4(a[4], a[5], a[6], a[7]);
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) atomicAdd(V.sums + 8 * (size_t)j + k, a[k]);
                }
            }
        }
        __syncthreads();
        if (tid < 3) s_cnt[tid] = 0u;
        __syncthreads();
    }
}

// The eight gradient components of the Gaussian in cell-order slot j after k_render_bwd_tile: the slots of its window's
// tiles in order (+ whatever went through the atomic accumulators, which are re-armed), times the Gaussian's
// constants (bwd_scale).  Output order {x, y | sx, sy, rho | r, g, b}; returns the Gaussian's original index.
__device__ __forceinline__ unsigned bwd_gather(const Params &P, const PlanView &V, unsigned j, bool use_atomics, float (&o)[8])
{
    const uint2 w = V.win[j];
    const float4 fa = V.fin[2 * (size_t)j], fb = V.fin[2 * (size_t)j + 1];
    float4 *sm = reinterpret_cast<float4 *>(V.sums) + 2 * (size_t)j;
    float4 a = sm[0], b = sm[1];
    if (a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f || b.x != 0.f || b.y != 0.f || b.z != 0.f || b.w != 0.f ||
        a.x != a.x) {
        sm[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        sm[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool dead = (int)(w.x & 0x7fffu) > (int)(w.x >> 16);
    if (!dead && !use_atomics) {
        int ntx, tx0, ty0;
        const int nt = bt_tile_span(w.x, w.y, P.row0, P.bt_hlog, ntx, tx0, ty0);
        if (nt <= P.part_k) {
            const float4 *pp = reinterpret_cast<const float4 *>(V.part + (size_t)j * P.part_k * 8);
            for (int t = 0; t < nt; ++t) {
                const float4 u = pp[2 * t], v = pp[2 * t + 1];
                a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
                b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
            }
        }
    }
    if (dead) {
        a = b = make_float4(0.f, 0.f, 0.f, 0.f);
        o[0] = o[1] = o[2] = o[3] = o[4] = o[5] = o[6] = o[7] = 0.f;
    } else {
        const float c = fa.x, fx = c * fa.w, fy = c * fb.x;
        o[0] = a.x * fx; o[1] = a.y * fy; o[2] = a.z * fx; o[3] = a.w * fy; o[4] = b.x * c * c;
        o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
    return __float_as_uint(fb.w);
}

__global__ __launch_bounds__(256) void k_bwd_gather(Params P, PlanView V, int use_atomics, float *__restrict__ g_sigmas,
                                                    float *__restrict__ g_coords, float *__restrict__ g_colors)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= (unsigned)P.s) return;
    float o[8];
    const unsigned i = bwd_gather(P, V, j, use_atomics != 0, o);
    float *pc = g_coords + (size_t)i * stride2(P), *ps = g_sigmas + (size_t)i * stride3(P), *pk = g_colors + (size_t)i * stride3(P);
    if (P.flags & GSASR_FLAG_OVERWRITE_GRADS) {
        pc[0] = o[0]; pc[1] = o[1]; ps[0] = o[2]; ps[1] = o[3]; ps[2] = o[4]; pk[0] = o[5]; pk[1] = o[6]; pk[2] = o[7];
    } else {   // (one thread per Gaussian: a plain read-modify-write)
        pc[0] += o[0]; pc[1] += o[1]; ps[0] += o[2]; ps[1] += o[3]; ps[2] += o[4]; pk[0] += o[5]; pk[1] += o[6]; pk[2] += o[7];
    }
}

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

// ---------------------------------------------------------------------------------------------------
// fused host prologue (reference utils/gaussian_splatting.py:174-180 and :121-123) and its backward
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prologue_fwd(const float *__restrict__ p, const float *__restrict__ step_ptr,
                                                      int n, int h, int w, float *__restrict__ sigmas,
                                                      float *__restrict__ coords, float *__restrict__ colors,
                                                      int nper, const int4 *__restrict__ geo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (geo && h == 0) {  // batched canvas: sample i/nper has its own size and step size (h, w given: one size for all)
        const int4 g = geo[i / nper];
        h = g.x;
        w = g.y;
    }
    float o[8];
    prologue_one(p + (size_t)i * 9, step_ptr[geo ? i / nper : 0], h, w, o);
    sigmas[i * 3 + 0] = o[0]; sigmas[i * 3 + 1] = o[1]; sigmas[i * 3 + 2] = o[2];
    coords[i * 2 + 0] = o[3]; coords[i * 2 + 1] = o[4];
    colors[i * 3 + 0] = o[5]; colors[i * 3 + 1] = o[6]; colors[i * 3 + 2] = o[7];
}

// chain rule of k_prologue_fwd for one Gaussian: q = its raw parameters, gs/gc/gk = d/d{sigmas, coords, colors}
__device__ __forceinline__ void prologue_chain(const float *__restrict__ q, float step, int h, int w, float gs0, float gs1,
                                               float gs2, float gc0, float gc1, float k0, float k1, float k2,
                                               float *__restrict__ o)
{
    const float W = (float)w, H = (float)h;
    const float s0 = sigmoidf_(q[0]), s1 = sigmoidf_(q[1]), th = tanhf(q[2]), al = sigmoidf_(q[3]);
    const float r = sigmoidf_(q[4]), g = sigmoidf_(q[5]), b = sigmoidf_(q[6]);
    o[0] = gs1 * (2.f / (H - 1.f) / step) * 0.99999f * s0 * (1.f - s0);
    o[1] = gs0 * (2.f / (W - 1.f) / step) * 0.99999f * s1 * (1.f - s1);
    o[2] = gs2 * 0.999999f * (1.f - th * th);
    o[3] = (k0 * r + k1 * g + k2 * b) * al * (1.f - al);
    o[4] = k0 * al * r * (1.f - r);
    o[5] = k1 * al * g * (1.f - g);
    o[6] = k2 * al * b * (1.f - b);
    o[7] = gc0 * 2.f * W / (W - 1.f);
    o[8] = gc1 * 2.f * H / (H - 1.f);
}

__global__ __launch_bounds__(256) void k_prologue_bwd(const float *__restrict__ p, const float *__restrict__ step_ptr,
                                                      int n, int h, int w, const float *__restrict__ gs,
                                                      const float *__restrict__ gc, const float *__restrict__ gk,
                                                      float *__restrict__ gp, int nper, const int4 *__restrict__ geo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (geo && h == 0) {
        const int4 g = geo[i / nper];
        h = g.x;
        w = g.y;
    }
    const float step = step_ptr[geo ? i / nper : 0];
    prologue_chain(p + (size_t)i * 9, step, h, w, gs[i * 3 + 0], gs[i * 3 + 1], gs[i * 3 + 2], gc[i * 2 + 0], gc[i * 2 + 1],
                   gk[i * 3 + 0], gk[i * 3 + 1], gk[i * 3 + 2], gp + (size_t)i * 9);
}

// the same behind the tile-stationary backward: the gather of the partial-gradient slots (bwd_gather) and the chain rule
// in one kernel, one thread per Gaussian in cell order -- the kernel-frame gradients never go through memory
__global__ __launch_bounds__(256) void k_prologue_bwd_gather(Params P, PlanView V, int use_atomics, const float *__restrict__ p,
                                                             const float *__restrict__ step_ptr, float *__restrict__ gp)
{
    const unsigned j = blockIdx.x * 256u + threadIdx.x;
    if (j >= (unsigned)P.s) return;
    float o[8];
    const unsigned i = bwd_gather(P, V, j, use_atomics != 0, o);
    int h = P.h, w = P.w;
    float step = step_ptr[0];
    if (P.batch > 1) {
        const Geo g = sample_geo(P, V, (int)(i / (unsigned)P.nper));
        h = g.h;
        w = g.w;
        step = step_ptr[i / (unsigned)P.nper];
    }
    prologue_chain(p + (size_t)i * 9, step, h, w, o[2], o[3], o[4], o[0], o[1], o[5], o[6], o[7], gp + (size_t)i * 9);
}

// planar [3, rows, w] (batched canvas: [B, 3, grad_rows, w], sample b's rows at the top of its planes) -> interleaved
// [rows, w, 3] / [B * slot, w, 3]: what autograd hands back for the planar image -> what k_render_bwd sweeps.  One
// thread per pixel: three coalesced plane reads, one 12-byte store.  Rows of a slot beyond grad_rows are left alone:
// the backward never reads outside a sample's own grid.
__global__ __launch_bounds__(256) void k_chw_to_hwc(const float *__restrict__ src, float *__restrict__ dst, int w, int rows,
                                                    int batch, int slot, int grad_rows)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int X = (int)(i % (size_t)w);
    const size_t Y = i / (size_t)w;
    if (Y >= (size_t)rows) return;
    size_t plane = (size_t)rows * w, at = Y * w + X;
    if (batch > 1) {
        const int b = (int)(Y / (size_t)slot), y = (int)(Y - (size_t)b * slot);
        if (y >= grad_rows) return;
        plane = (size_t)grad_rows * w;
        at = (size_t)b * 3 * plane + (size_t)y * w + X;
    }
    float *o = dst + (Y * w + X) * 3;
    o[0] = src[at];
    o[1] = src[at + plane];
    o[2] = src[at + 2 * plane];
}

// ---------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------
#define HIP_TRY(expr)                                    \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

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

int check_ws(const gsasr_dims *dims, const void *ws, size_t ws_bytes, Layout &L, bool planning = false)
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


// ---------------------------------------------------------------------------------------------------
// row-band shard: neighbour exchange (device side of gsasr_amd/shard.py)
// ---------------------------------------------------------------------------------------------------
// One thread per Gaussian: its row window on the FULL grid (the same gaussian_box() the plan uses, so the
// selection is exactly the set of Gaussians the neighbour's plan would keep) against this rank's band.
__global__ __launch_bounds__(256) void k_band_select(Params P, int band0, int band1, int rows_above, int rows_below,
                                                     int cap, const float *__restrict__ packed,
                                                     float *__restrict__ up, float *__restrict__ down,
                                                     int *__restrict__ up_index, int *__restrict__ down_index,
                                                     int *__restrict__ counts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool go_up = false, go_down = false, far = false;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    if (i < P.s) {
        ra = reinterpret_cast<const float4 *>(packed)[2 * (size_t)i];      // sx sy rho x
        rb = reinterpret_cast<const float4 *>(packed)[2 * (size_t)i + 1];  // y r g b
        const Box b = gaussian_box(ra.x, ra.y, ra.w, rb.x, P, Geo{P.h, P.w, 0, 0}, P.kcut);   // P.row0/row1 = whole grid here
        if (b.cls != 2) {
            go_up = rows_above > 0 && b.r0 < band0;
            go_down = rows_below > 0 && b.r1 >= band1;
            far = (go_up && b.r0 < band0 - rows_above) || (go_down && b.r1 >= band1 + rows_below);
        }
    }
    // wave-aggregated slot allocation: one returning atomic per wave and list
    const unsigned long long mu = __ballot(go_up), md = __ballot(go_down), mf = __ballot(far);
    const unsigned long long below = (1ull << lane) - 1ull;
    int bu = 0, bd = 0;
    if (lane == 0) {
        if (mu) bu = atomicAdd(&counts[0], __builtin_popcountll(mu));
        if (md) bd = atomicAdd(&counts[1], __builtin_popcountll(md));
        if (mf) atomicAdd(&counts[2], __builtin_popcountll(mf));
    }
    bu = __shfl(bu, 0);
    bd = __shfl(bd, 0);
    if (go_up) {
        const int slot = bu + __builtin_popcountll(mu & below);
        if (slot < cap) {
            reinterpret_cast<float4 *>(up)[2 * (size_t)slot] = ra;
            reinterpret_cast<float4 *>(up)[2 * (size_t)slot + 1] = rb;
            up_index[slot] = i;
        }
    }
    if (go_down) {
        const int slot = bd + __builtin_popcountll(md & below);
        if (slot < cap) {
            reinterpret_cast<float4 *>(down)[2 * (size_t)slot] = ra;
            reinterpret_cast<float4 *>(down)[2 * (size_t)slot + 1] = rb;
            down_index[slot] = i;
        }
    }
}

// 8 threads per returned record; a Gaussian can sit in both lists, hence atomics (two adds at most per word)
__global__ __launch_bounds__(256) void k_band_merge(int s, int cap, float *__restrict__ g_packed,
                                                    const float *__restrict__ g_up, const float *__restrict__ g_down,
                                                    const int *__restrict__ up_index, const int *__restrict__ down_index,
                                                    const int *__restrict__ counts)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = t >> 3, k = t & 7;
    if (j >= 2 * cap) return;
    const bool is_down = j >= cap;
    const int jj = is_down ? j - cap : j;
    if (jj >= min(counts[is_down ? 1 : 0], cap)) return;
    const int i = (is_down ? down_index : up_index)[jj];
    if ((unsigned)i >= (unsigned)s) return;
    atomicAdd(&g_packed[(size_t)i * 8 + k], (is_down ? g_down : g_up)[(size_t)jj * 8 + k]);
}

}  // namespace

extern "C" {

int gsasr_abi_version(void) { return GSASR_SPLAT_ABI_VERSION; }

const char *gsasr_last_error(void) { return tl_err; }

void gsasr_set_default_cutoff(float tau) { g_default_cutoff.store(tau, std::memory_order_relaxed); }

float gsasr_get_default_cutoff(void) { return default_cutoff(); }

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

}  // extern "C"

namespace {
// The plan: [memset of this parity's counters unless the caller vouches for them] -> classify (with the host prologue
// fused in when `raw` is given: sigmas/coords/colors are then OUTPUTS) -> [scan] -> bin.
int plan_impl(const float *sigmas, const float *coords, const float *colors, const gsasr_dims *dims, void *workspace,
              size_t workspace_bytes, void *stream, const float *raw, const StepSrc &SS)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L, true)) return rc;
    note_plan(workspace, dims, L.part_k, L.tl_hlog);
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
}  // namespace

extern "C" {

int gsasr_splat_plan(const float *sigmas, const float *coords, const float *colors, const gsasr_dims *dims,
                     void *workspace, size_t workspace_bytes, void *stream)
{
    return plan_impl(sigmas, coords, colors, dims, workspace, workspace_bytes, stream, nullptr, StepSrc{});
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
    const int subs_x = (dims->w + SUBX - 1) / SUBX, tiles_y = (rows + SUBY - 1) / SUBY;
    hipStream_t st = (hipStream_t)stream;
    const long nsub = (long)subs_x * tiles_y;
    if (fwd_wants_wide(dims)) {
        const int wx = (dims->w + 2 * WIDE - 1) / (2 * WIDE), wy = (rows + 2 * WIDE - 1) / (2 * WIDE);
        const dim3 grid((unsigned)wx * (unsigned)wy), block(256);
        if (L.tl_ok && L.tl_hlog == 5) {    // the plan's tile lists (32 x 32-px tiles)
            if (P.bounded) hipLaunchKernelGGL(k_render_fwd16_list<true>, grid, block, 0, st, P, V, img, wx);
            else hipLaunchKernelGGL(k_render_fwd16_list<false>, grid, block, 0, st, P, V, img, wx);
        } else if (P.bounded) hipLaunchKernelGGL(k_render_fwd16<true>, grid, block, 0, st, P, V, img, wx);
        else hipLaunchKernelGGL(k_render_fwd16<false>, grid, block, 0, st, P, V, img, wx);
    } else if (nsub < 4096 && !(L.tl_ok && L.tl_hlog == 4)) {
        // fewer sub-tiles than half the chip's 8192 wave slots: split each sub-tile's Gaussian list over
        // 2..16 waves so that about one full set of waves is in flight
        int nw = 2;
        while (nw < 16 && nsub * nw < 8192) nw *= 2;
        const dim3 grid((unsigned)nsub), block((unsigned)nw * 64u);
        if (P.bounded)
            hipLaunchKernelGGL(k_render_fwd_split<true>, grid, block, 0, st, P, V, img, subs_x);
        else
            hipLaunchKernelGGL(k_render_fwd_split<false>, grid, block, 0, st, P, V, img, subs_x);
    } else if (L.tl_ok && L.tl_hlog == 4) {
        // the plan's tile lists (32 x 16-px tiles: the same workgroup tile as the two-level walk, and its two shapes)
        const int tx4 = (subs_x + 3) / 4;
        const bool two = nsub < 8192;
        const dim3 grid((unsigned)tx4 * (unsigned)tiles_y), block(two ? 512 : 256);
        if (P.bounded) {
            if (two) hipLaunchKernelGGL((k_render_fwd_list<true, 2>), grid, block, 0, st, P, V, img, tx4);
            else hipLaunchKernelGGL((k_render_fwd_list<true, 1>), grid, block, 0, st, P, V, img, tx4);
        } else {
            if (two) hipLaunchKernelGGL((k_render_fwd_list<false, 2>), grid, block, 0, st, P, V, img, tx4);
            else hipLaunchKernelGGL((k_render_fwd_list<false, 1>), grid, block, 0, st, P, V, img, tx4);
        }
    } else {
        // two-level walk; images with fewer sub-tiles than the chip has wave slots (4096..8191, e.g. the batched canvas
        // of config 5) get two waves per sub-tile (measured -13% at 4608 sub-tiles, +2..14% above 8192)
        const int tx4 = (subs_x + 3) / 4;
        const bool two = nsub < 8192;
        const dim3 grid((unsigned)tx4 * (unsigned)tiles_y), block(two ? 512 : 256);
        if (P.bounded) {
            if (two) hipLaunchKernelGGL((k_render_fwd2<true, 2>), grid, block, 0, st, P, V, img, tx4);
            else hipLaunchKernelGGL((k_render_fwd2<true, 1>), grid, block, 0, st, P, V, img, tx4);
        } else {
            if (two) hipLaunchKernelGGL((k_render_fwd2<false, 2>), grid, block, 0, st, P, V, img, tx4);
            else hipLaunchKernelGGL((k_render_fwd2<false, 1>), grid, block, 0, st, P, V, img, tx4);
        }
    }
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_forward_subtile_width(const gsasr_dims *dims)
{
    if (!dims || dims->s <= 0 || dims->w <= 0 || dims->row1 < dims->row0) return fail(GSASR_ERR_ARG, "bad dims");
    return fwd_wants_wide(dims) ? WIDE : SUBX;
}

}  // extern "C"

namespace {
// mode: 0 = Gaussian-stationary, 1 = tile-stationary with slots, 2 = tile-stationary with atomics
int bwd_mode(const gsasr_dims *dims, const Layout &L)
{
    // Default: whatever the plan was made for (bwd_wants_tile).  Measured on MI355X the two kernels are
    // within ~10% of each other at every scale -- both are bound by VALU issue: Gaussian-stationary ahead for GSASR's
    // LR-pixel sized Gaussians at x4 (38.4 vs 38.8 + 5.0 us gather at config 2), tile-stationary ahead from x8 up
    // (config 4: 1.67 vs 1.79 ms); the tile-stationary one is deterministic and reads the planar gradient autograd returns.
    const unsigned f = dims->flags;
    int mode = L.part_k > 0 ? 1 : 0;      // a plan with slots was made for the tile-stationary kernel (bwd_wants_tile)
    if (f & GSASR_FLAG_BWD_GAUSSIAN) mode = 0;
    else if (f & GSASR_FLAG_BWD_ATOMIC) mode = 2;
    else if (f & (GSASR_FLAG_BWD_TILE | GSASR_FLAG_CHW_GRAD)) mode = 1;
    else if (bwd_env()) mode = bwd_env() - 1;
    if (L.part_k == 0 && mode == 1) mode = (f & GSASR_FLAG_CHW_GRAD) ? 2 : 0;   // a forward-only plan has no slots
    return mode;
}

// Backward of the splat.  With `gather` the kernel-frame gradients are written (or added) to g_*; without it a
// tile-stationary run stops after the tile kernel and the caller fuses the gather into its next kernel
// (k_prologue_bwd_gather) -- *mode_out tells which kernel ran.
int splat_backward(const float *sigmas, const float *coords, const float *colors, const float *grad_img, float *g_sigmas,
                   float *g_coords, float *g_colors, const gsasr_dims *dims, const void *workspace, size_t workspace_bytes,
                   void *stream, bool gather, int *mode_out)
{
    Layout L;
    if (int rc = check_ws(dims, workspace, workspace_bytes, L)) return rc;
    const int mode = bwd_mode(dims, L);
    if (mode_out) *mode_out = mode;
    if (dims->flags & GSASR_FLAG_FORWARD_ONLY) return fail(GSASR_ERR_PLAN, "the plan was made with GSASR_FLAG_FORWARD_ONLY: it holds no backward records");
    if (dims->s == 0) return GSASR_OK;
    if (gather && (!g_sigmas || !g_coords || !g_colors)) return fail(GSASR_ERR_ARG, "null pointer");
    if (mode == 0 && (dims->flags & GSASR_FLAG_CHW_GRAD))
        return fail(GSASR_ERR_ARG, "GSASR_FLAG_CHW_GRAD needs the tile-stationary backward");
    hipStream_t st = (hipStream_t)stream;
    const Params P = make_params(dims, L);
    const PlanView V = make_view(L, const_cast<void *>(workspace));
    const int rows = dims->row1 - dims->row0;
    if (rows > 0 && !grad_img) return fail(GSASR_ERR_ARG, "null pointer");
    if (mode == 0) {
        if (!sigmas || !coords || !colors || !g_sigmas || !g_coords || !g_colors) return fail(GSASR_ERR_ARG, "null pointer");
        if (rows == 0) {  // empty band: the gradient is zero
            if (dims->flags & GSASR_FLAG_OVERWRITE_GRADS) {
                const size_t e3 = (dims->flags & GSASR_FLAG_STRIDE8) ? 0 : sizeof(float) * 3 * (size_t)dims->s;
                if (!e3) {
                    HIP_TRY(hipMemsetAsync(g_sigmas, 0, sizeof(float) * 8 * (size_t)dims->s, st));   // one packed [s,8] array
                } else {
                    HIP_TRY(hipMemsetAsync(g_sigmas, 0, e3, st));
                    HIP_TRY(hipMemsetAsync(g_coords, 0, sizeof(float) * 2 * (size_t)dims->s, st));
                    HIP_TRY(hipMemsetAsync(g_colors, 0, e3, st));
                }
            }
            return GSASR_OK;
        }
        const dim3 grid((unsigned)((dims->s + BWD_WAVES - 1) / BWD_WAVES)), block(64 * BWD_WAVES);
        // Two instantiations of the same sweep (identical results): with the two-trip unrolled loop (88 VGPRs, 5 waves
        // per SIMD) for windows of many trips, without it (70 VGPRs, 7 waves) for small windows.  The window sizes are on
        // the device; GSASR's Gaussians are LR-pixel sized, so pixels per Gaussian is a good proxy (x4: 16, x8: 64).
        const bool unroll = (double)rows * (double)dims->w >= BWD_UNROLL_MIN * (double)dims->s;
        // (Measured dead ends, git history: two Gaussians per wave one after the other, side by side in half waves, and --
        // round 3 -- sharing every gradient load over the union of their windows: 44-50 us against 37 us at config 2; rows
        // or a cell's window staged in LDS; a planar-gradient sweep.  A wave's life is its chain of dependent round trips:
        // what helped was running the sweep one trip ahead)
#define GSASR_BWD(B, U) hipLaunchKernelGGL((k_render_bwd<B, U>), grid, block, 0, st, P, V, grad_img, g_sigmas, g_coords, g_colors)
        if (P.bounded) { if (unroll) GSASR_BWD(true, true); else GSASR_BWD(true, false); }
        else { if (unroll) GSASR_BWD(false, true); else GSASR_BWD(false, false); }
#undef GSASR_BWD
        HIP_TRY(hipGetLastError());
        return GSASR_OK;
    }
    if (L.part_k == 0)
        // the atomic variant, or a tile-stationary backward asked of a plan that was not made for it: the plan may have
        // left the accumulators alone (k_bin zeroes them only where it knows they will be used)
        HIP_TRY(hipMemsetAsync(V.sums, 0, (size_t)dims->s * 32, st));
    if (rows > 0) {
        const int bth = 1 << P.bt_hlog;
        const int tiles_x = (dims->w + BT_W - 1) / BT_W, tiles_y = (rows + bth - 1) / bth;
        const dim3 grid((unsigned)tiles_x * (unsigned)tiles_y), block((unsigned)BT_THREADS << (P.bt_hlog - 4));
        // small rounds + five waves per SIMD from 32 HR pixels per Gaussian up (where this kernel is the default)
        const bool sparse = (double)rows * (double)dims->w >= 32.0 * (double)dims->s;
#define GSASR_BT(B, C, H) hipLaunchKernelGGL((k_render_bwd_tile<B, C, H>), grid, block, 0, st, P, V, grad_img, tiles_x, mode == 2)
#define GSASR_BT2(B, C) do { if (P.bt_hlog == 5) GSASR_BT(B, (C) / 2, 5); else GSASR_BT(B, C, 4); } while (0)
        if (P.bounded) { if (sparse) GSASR_BT2(true, 4 / BT_WAVES); else GSASR_BT2(true, 8 / BT_WAVES); }
        else { if (sparse) GSASR_BT2(false, 4 / BT_WAVES); else GSASR_BT2(false, 8 / BT_WAVES); }
#undef GSASR_BT2
#undef GSASR_BT
        HIP_TRY(hipGetLastError());
    }
    if (gather) {
        // (an empty band left no slots behind: the gather then only sees the zero accumulators)
        Params Pg = P;
        if (rows == 0) Pg.part_k = 0;
        hipLaunchKernelGGL(k_bwd_gather, dim3((unsigned)((dims->s + 255) / 256)), dim3(256), 0, st, Pg, V,
                           (int)(mode == 2 || rows == 0), g_sigmas, g_coords, g_colors);
        HIP_TRY(hipGetLastError());
    }
    return GSASR_OK;
}
}  // namespace

extern "C" {

int gsasr_splat_backward(const float *sigmas, const float *coords, const float *colors, const float *grad_img,
                         float *g_sigmas, float *g_coords, float *g_colors, const gsasr_dims *dims,
                         const void *workspace, size_t workspace_bytes, void *stream)
{
    return splat_backward(sigmas, coords, colors, grad_img, g_sigmas, g_coords, g_colors, dims, workspace, workspace_bytes,
                          stream, true, nullptr);
}

int gsasr_prologue_forward(const float *gs_parameters, const float *step_size, int n, int h, int w, float *sigmas,
                           float *coords, float *colors, void *stream)
{
    if (n < 0 || h < 2 || w < 2) return fail(GSASR_ERR_ARG, "bad n/h/w");
    if (n == 0) return GSASR_OK;
    if (!gs_parameters || !step_size || !sigmas || !coords || !colors) return fail(GSASR_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_prologue_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       gs_parameters, step_size, n, h, w, sigmas, coords, colors, 0, (const int4 *)nullptr);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_prologue_backward(const float *gs_parameters, const float *step_size, int n, int h, int w,
                            const float *g_sigmas, const float *g_coords, const float *g_colors, float *g_parameters,
                            void *stream)
{
    if (n < 0 || h < 2 || w < 2) return fail(GSASR_ERR_ARG, "bad n/h/w");
    if (n == 0) return GSASR_OK;
    if (!gs_parameters || !step_size || !g_sigmas || !g_coords || !g_colors || !g_parameters)
        return fail(GSASR_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_prologue_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       gs_parameters, step_size, n, h, w, g_sigmas, g_coords, g_colors, g_parameters, 0, (const int4 *)nullptr);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

// ---- whole-step entry points ----------------------------------------------------------------------
namespace {
struct StepLayout {
    size_t plan_bytes, off_step, off_sig, off_xy, off_col, off_gsig, off_gxy, off_gcol, off_ghwc, total;
};
StepLayout make_step_layout(const gsasr_dims *d, const void *planned_ws = nullptr)
{
    StepLayout S;
    const size_t n = (size_t)d->s;
    S.plan_bytes = (planned_ws ? plan_layout(d, planned_ws) : make_layout(d)).total;   // (the plan's own slot count: plan_layout)
    size_t o = S.plan_bytes;
    S.off_step = o; o += align_up(GSASR_MAX_BATCH * 4, 256);   // the step size of every sample, as the prologue used it
    S.off_sig = o;  o += align_up(n * 12, 256);
    S.off_xy = o;   o += align_up(n * 8, 256);
    S.off_col = o;  o += align_up(n * 12, 256);
    const size_t nb = (d->flags & GSASR_FLAG_FORWARD_ONLY) ? 0 : n;   // (no gradient scratch for a forward-only step)
    S.off_gsig = o; o += align_up(nb * 12, 256);
    S.off_gxy = o;  o += align_up(nb * 8, 256);
    S.off_gcol = o; o += align_up(nb * 12, 256);
    // a planar upstream gradient (GSASR_FLAG_CHW_GRAD) in front of the Gaussian-stationary backward is interleaved into
    // this scratch by k_chw_to_hwc (the tile-stationary backward stages the planes directly and needs none)
    S.off_ghwc = o;
    if ((d->flags & GSASR_FLAG_CHW_GRAD) && !(d->flags & (GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_ATOMIC | GSASR_FLAG_FORWARD_ONLY)))
        o += align_up((size_t)(d->row1 - d->row0) * (size_t)d->w * 12, 256);
    S.total = o;
    return S;
}
}  // namespace

size_t gsasr_step_workspace_bytes(const gsasr_dims *dims)
{
    if (!dims_ok(dims)) {
        fail(GSASR_ERR_ARG, "bad dims");
        return 0;
    }
    return make_step_layout(dims).total;
}

namespace {
// prologue (per-sample sizes and step sizes on a batched canvas) + plan of a whole-step call
int step_prologue_plan(const float *gs_parameters, StepSrc SS, const gsasr_dims *dims, void *workspace,
                       size_t workspace_bytes, void *stream, StepLayout &S)
{
    if (!dims_ok(dims)) return fail(GSASR_ERR_ARG, "bad dims");
    if (dims->flags & GSASR_FLAG_STRIDE8) return fail(GSASR_ERR_ARG, "GSASR_FLAG_STRIDE8 does not apply to the step entry points");
    S = make_step_layout(dims);
    if (!workspace || ((uintptr_t)workspace & 255u) || workspace_bytes < S.total)
        return fail(GSASR_ERR_WORKSPACE, "workspace null, misaligned or smaller than gsasr_step_workspace_bytes()");
    char *b = (char *)workspace;
    float *sig = (float *)(b + S.off_sig), *xy = (float *)(b + S.off_xy), *col = (float *)(b + S.off_col);
    if (dims->s > 0 && (!gs_parameters || (!SS.step && !SS.sm))) return fail(GSASR_ERR_ARG, "null pointer");
    if (SS.sm && SS.stride < 2) return fail(GSASR_ERR_ARG, "scale_modify stride must be >= 2");
    SS.keep = (float *)(b + S.off_step);
    if (dims->batch > 1 && dims->s > 0) {  // the per-sample geometry must be in place before the classify kernel reads it
        const PlanView V = make_view(make_layout(dims), workspace, dims->flags);
        if (int rc = launch_batch_geo(dims, V, (hipStream_t)stream)) return rc;
    }
    // (the prologue runs inside the plan's first kernel: k_classify<true>)
    return plan_impl(sig, xy, col, dims, workspace, S.plan_bytes, stream, dims->s > 0 ? gs_parameters : nullptr, SS);
}
}  // namespace

int gsasr_step_forward(const float *gs_parameters, const float *step_size, const gsasr_dims *dims, void *workspace,
                       size_t workspace_bytes, float *img, void *stream)
{
    StepLayout S;
    StepSrc SS{};
    SS.step = step_size;
    if (int rc = step_prologue_plan(gs_parameters, SS, dims, workspace, workspace_bytes, stream, S)) return rc;
    return gsasr_splat_forward(dims, workspace, S.plan_bytes, img, stream);
}

int gsasr_step_forward_sm(const float *gs_parameters, const float *scale_modify, int sm_stride, float default_step_size,
                          int *mismatch, const gsasr_dims *dims, void *workspace, size_t workspace_bytes, float *img,
                          void *stream)
{
    StepLayout S;
    StepSrc SS{};
    SS.sm = scale_modify; SS.stride = sm_stride; SS.def_step = default_step_size; SS.mismatch = mismatch;
    if (!scale_modify && dims && dims->s > 0) return fail(GSASR_ERR_ARG, "null pointer");
    if (int rc = step_prologue_plan(gs_parameters, SS, dims, workspace, workspace_bytes, stream, S)) return rc;
    return gsasr_splat_forward(dims, workspace, S.plan_bytes, img, stream);
}

int gsasr_step_backward(const float *gs_parameters, const float *step_size, const float *grad_img,
                        float *g_parameters, const gsasr_dims *dims, void *workspace, size_t workspace_bytes,
                        void *stream)
{
    if (!dims_ok(dims)) return fail(GSASR_ERR_ARG, "bad dims");
    if (dims->flags & GSASR_FLAG_STRIDE8) return fail(GSASR_ERR_ARG, "GSASR_FLAG_STRIDE8 does not apply to the step entry points");
    const StepLayout S = make_step_layout(dims, workspace);
    if (!workspace || ((uintptr_t)workspace & 255u) || workspace_bytes < S.total)
        return fail(GSASR_ERR_WORKSPACE, "workspace null, misaligned or smaller than gsasr_step_workspace_bytes()");
    char *b = (char *)workspace;
    float *sig = (float *)(b + S.off_sig), *xy = (float *)(b + S.off_xy), *col = (float *)(b + S.off_col);
    float *gs = (float *)(b + S.off_gsig), *gc = (float *)(b + S.off_gxy), *gk = (float *)(b + S.off_gcol);
    if (!step_size) step_size = (const float *)(b + S.off_step);   // what the forward's prologue used (gsasr_step_forward_sm)
    gsasr_dims d = *dims;
    d.flags |= GSASR_FLAG_OVERWRITE_GRADS;
    if ((d.flags & GSASR_FLAG_CHW_GRAD) && S.total > S.off_ghwc && dims->s > 0 && d.row1 > d.row0) {
        // Gaussian-stationary kernel behind a planar gradient: interleave it into the scratch first
        if (!grad_img) return fail(GSASR_ERR_ARG, "null pointer");
        float *hwc = (float *)(b + S.off_ghwc);
        const int rows = d.row1 - d.row0;
        const size_t px = (size_t)rows * d.w;
        hipLaunchKernelGGL(k_chw_to_hwc, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, (hipStream_t)stream, grad_img, hwc,
                           d.w, rows, batch_of(&d), d.batch > 1 ? d.slot : rows, d.grad_rows > 0 ? d.grad_rows : (d.batch > 1 ? d.slot : rows));
        HIP_TRY(hipGetLastError());
        grad_img = hwc;
        d.flags &= ~GSASR_FLAG_CHW_GRAD;
        d.flags |= GSASR_FLAG_BWD_GAUSSIAN;
    }
    int mode = 0;
    if (int rc = splat_backward(sig, xy, col, grad_img, gs, gc, gk, &d, workspace, S.plan_bytes, stream, false, &mode)) return rc;
    if (dims->s == 0) return GSASR_OK;
    if (!gs_parameters || !step_size || !g_parameters) return fail(GSASR_ERR_ARG, "null pointer");
    const dim3 grid((unsigned)((dims->s + 255) / 256)), block(256);
    if (mode != 0) {   // tile-stationary: gather of the slots + chain rule in one kernel
        const Layout L = plan_layout(dims, workspace);
        const PlanView V = make_view(L, workspace);
        hipLaunchKernelGGL(k_prologue_bwd_gather, grid, block, 0, (hipStream_t)stream, make_params(&d, L), V, (int)(mode == 2 || d.row1 == d.row0),
                           gs_parameters, step_size, g_parameters);
        HIP_TRY(hipGetLastError());
        return GSASR_OK;
    }
    if (dims->batch > 1) {
        const PlanView V = make_view(make_layout(dims), workspace);
        int uh, uw;
        batch_uniform(dims, uh, uw);
        hipLaunchKernelGGL(k_prologue_bwd, grid, block, 0, (hipStream_t)stream, gs_parameters, step_size, dims->s, uh, uw, gs, gc,
                           gk, g_parameters, dims->s / dims->batch, (const int4 *)V.geo);
        HIP_TRY(hipGetLastError());
        return GSASR_OK;
    }
    return gsasr_prologue_backward(gs_parameters, step_size, dims->s, dims->h, dims->w, gs, gc, gk, g_parameters, stream);
}


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
    if (int rc = step_prologue_plan(gs_parameters, SS, dims, workspace, workspace_bytes, stream, S)) return rc;
    return gsasr_splat_sample_forward(dims, workspace, S.plan_bytes, points, n_points, out, sample_ws, sample_ws_bytes, stream);
}

int gsasr_step_sample_forward_sm(const float *gs_parameters, const float *scale_modify, int sm_stride, float default_step_size,
                                 int *mismatch, const gsasr_dims *dims, void *workspace, size_t workspace_bytes, const int *points,
                                 int n_points, float *out, void *sample_ws, size_t sample_ws_bytes, void *stream)
{
    StepLayout S;
    StepSrc SS{};
    SS.sm = scale_modify; SS.stride = sm_stride; SS.def_step = default_step_size; SS.mismatch = mismatch;
    if (!scale_modify && dims && dims->s > 0) return fail(GSASR_ERR_ARG, "null pointer");
    if (int rc = step_prologue_plan(gs_parameters, SS, dims, workspace, workspace_bytes, stream, S)) return rc;
    return gsasr_splat_sample_forward(dims, workspace, S.plan_bytes, points, n_points, out, sample_ws, sample_ws_bytes, stream);
}

int gsasr_step_sample_backward(const float *gs_parameters, const float *step_size, const float *grad_out,
                               float *g_parameters, const gsasr_dims *dims, void *workspace, size_t workspace_bytes,
                               const int *points, int n_points, void *sample_ws, size_t sample_ws_bytes, void *stream)
{
    if (!dims_ok(dims)) return fail(GSASR_ERR_ARG, "bad dims");
    if (dims->flags & GSASR_FLAG_STRIDE8) return fail(GSASR_ERR_ARG, "GSASR_FLAG_STRIDE8 does not apply to the step entry points");
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
        const PlanView V = make_view(make_layout(dims), workspace);
        int uh, uw;
        batch_uniform(dims, uh, uw);
        hipLaunchKernelGGL(k_prologue_bwd, dim3((unsigned)((dims->s + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           gs_parameters, step_size, dims->s, uh, uw, gs, gc, gk, g_parameters, dims->s / dims->batch,
                           (const int4 *)V.geo);
        HIP_TRY(hipGetLastError());
        return GSASR_OK;
    }
    return gsasr_prologue_backward(gs_parameters, step_size, dims->s, dims->h, dims->w, gs, gc, gk, g_parameters, stream);
}


int gsasr_band_select(const float *packed, const gsasr_dims *dims, int rows_above, int rows_below, int cap,
                      float *up, float *down, int *up_index, int *down_index, int *counts, void *stream)
{
    if (!dims_ok(dims) || dims->batch > 1) return fail(GSASR_ERR_ARG, "bad dims (the band exchange does not take a batched canvas)");
    if (cap < 0 || rows_above < 0 || rows_below < 0 || !counts || (cap > 0 && (!up || !down || !up_index || !down_index)) ||
        (dims->s > 0 && !packed))
        return fail(GSASR_ERR_ARG, "gsasr_band_select: null pointer or negative size");
    hipStream_t st = (hipStream_t)stream;
    gsasr_dims whole = *dims;   // the footprint is taken on the full grid, then compared with the band
    whole.row0 = 0;
    whole.row1 = dims->h;
    const Layout L = make_layout(&whole);
    const Params P = make_params(&whole, L);
    HIP_TRY(hipMemsetAsync(counts, 0, 4 * sizeof(int), st));
    if (cap > 0) {  // 0xff.. = NaN records: dead Gaussians for every kernel of this library
        HIP_TRY(hipMemsetAsync(up, 0xff, (size_t)cap * 32, st));
        HIP_TRY(hipMemsetAsync(down, 0xff, (size_t)cap * 32, st));
    }
    if (dims->s > 0)
        hipLaunchKernelGGL(k_band_select, dim3((dims->s + 255) / 256), dim3(256), 0, st, P, dims->row0, dims->row1,
                           rows_above, rows_below, cap, packed, up, down, up_index, down_index, counts);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
}

int gsasr_band_merge(float *g_packed, int s, const float *g_up, const float *g_down, const int *up_index,
                     const int *down_index, const int *counts, int cap, void *stream)
{
    if (s < 0 || cap < 0 || !counts || (cap > 0 && ((s > 0 && !g_packed) || !g_up || !g_down || !up_index || !down_index)))   // (a rank may own no Gaussian)
        return fail(GSASR_ERR_ARG, "gsasr_band_merge: null pointer or negative size");
    if (cap == 0 || s == 0) return GSASR_OK;
    hipLaunchKernelGGL(k_band_merge, dim3((2 * cap * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, s, cap,
                       g_packed, g_up, g_down, up_index, down_index, counts);
    HIP_TRY(hipGetLastError());
    return GSASR_OK;
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
