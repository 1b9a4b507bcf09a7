// splat_common.h -- what the translation units of libgsasr_splat.so share: constants, the plan's layout (Params, PlanView,
// Layout), host helpers that decide which kernel runs, the geometry and wave helpers of the device code.  Everything lives in
// namespace gsasr_detail (hidden visibility: the library exports only the extern "C" entry points of include/gsasr_splat.h).
// gsasr_splat.hip #includes all parts as ONE translation unit (the micro-benchmark tools/mb.hip builds that); gsasr_amd/build.py
// compiles the parts separately.
#ifndef GSASR_SPLAT_COMMON_H
#define GSASR_SPLAT_COMMON_H
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "gsasr_splat.h"

namespace gsasr_detail {

// ---- state and services with ONE instance in the library (defined in splat_api.hip) ----
int fail(int code, const char *msg);                 // sets the thread-local message of gsasr_last_error(), returns `code`
int hip_fail(hipError_t e, const char *where);
const char *last_error_message();
struct KernelChoice { unsigned flags; int list_cap; };
KernelChoice registered_choice(const gsasr_dims *d);   // gsasr_set_kernel_choice's entry for this shape, {0, 0} when none (splat_api.hip)
float default_cutoff();                              // process default of the support cutoff (0 = adaptive)
void store_default_cutoff(float tau);


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
                             // x5 -3..-7%, x8 -4..-10%, x12 -13..-15%, x16 -20%, x32 -25%; x4: +8% (profiles/history/r04_fwd_wide.txt)
#endif
#ifndef TL_MIN_SUB
#define TL_MIN_SUB 2048      // sub-tiles (8 x 16 px) from which a dense plan carries tile lists by default: 512^2 and up.  Measured at 16
                             // Gaussians per LR pixel (profiles/history/r05_small_dense.txt): 512^2 forward 134 -> 72 us for +8 us of plan, 640^2
                             // 207 -> 126; 384^2 level (+7 us of plan, nothing off the forward), 256^2 and below the split search kernel
                             // wins (33 against 48 us; four waves per sub-tile instead of two: 42)
#endif
#ifndef BWD_LX21_MAX
#define BWD_LX21_MAX 21      // windows of 17..21 columns sweep 21 columns x 3 row slots (bwd_sweep; k_bin pads their rows to 6); 16 = off
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
    float kb_max;    // > 0 (adaptive default only): sqrt(2 GSASR_SPLAT_GRAD_TAU) -- the backward sweeps the window of min(tau', that)
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

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

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

inline bool dims_ok(const gsasr_dims *d)
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

inline int batch_of(const gsasr_dims *d) { return d->batch > 1 ? d->batch : 1; }

inline int classify_blocks(const gsasr_dims *d)
{
    const int pxn = d->w * batch_of(d);  // one px table per sample
    const int n = d->s > pxn ? (d->s > d->h ? d->s : d->h) : (pxn > d->h ? pxn : d->h);
    return (n + 255) / 256;
}

// DEVELOPMENT switches (A/B runs of one build on one box; tools/collect_profiles.sh): environment variables that override a
// kernel choice the library makes by shape.  They are read ONLY when GSASR_SPLAT_DEV=1 is set as well -- a production process
// that happens to carry one of the names in its environment is not affected -- and each is read once.
//   GSASR_SPLAT_FWD_WIDE=0|1   GSASR_SPLAT_BWD=gaussian|tile|atomic   GSASR_SPLAT_BT_TALL=0|1   GSASR_SPLAT_ADAPT=0
//   GSASR_SPLAT_FUSED_MAX=<k_bin blocks>   GSASR_SPLAT_LISTS=0|1   GSASR_SPLAT_BWD8=0|1   GSASR_SPLAT_FWD_SPLIT=0
//   GSASR_SPLAT_FWD_PARTS=1|2 (waves per sub-tile of the 8 x 16 forward)   GSASR_SPLAT_BWD_UNROLL=0|1 (Gaussian-stationary sweep)
// (GSASR_SPLAT_CUTOFF is not one of them: it is the documented process default of the support cutoff, INTEGRATION.md.)
inline const char *dev_switch(const char *name)
{
    static const bool on = [] { const char *e = getenv("GSASR_SPLAT_DEV"); return e && atoi(e) != 0; }();
    return on ? getenv(name) : nullptr;
}

// which backward kernel: explicit flag > development switch GSASR_SPLAT_BWD (gaussian | tile | atomic) > default
// development switch: GSASR_SPLAT_FWD_WIDE=0 / 1 forces the wide forward (16 x 16 sub-tiles) off / on
inline int fwd_wide_env()
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

// development switch: GSASR_SPLAT_FWD_SPLIT=0 sends small images (< 4096 sub-tiles) to the two-level kernels instead of k_render_fwd_split
inline bool fwd_split_env()
{
    static const bool on = [] { const char *e = dev_switch("GSASR_SPLAT_FWD_SPLIT"); return !e || atoi(e) != 0; }();
    return on;
}

// Which forward kernel.  Scale factors from x5 up (FWD_WIDE_MIN HR pixels per Gaussian: windows of ~25 px and more), single
// images of at least 2 Mpx (16 384 sub-tiles of 8 x 16): 16 x 16 sub-tiles, four pixels per lane (k_render_fwd16).
// GSASR_FLAG_FWD_WIDE / _NARROW (or the environment) override the rule for A/B runs and tests -- the wide kernel renders any
// single image.
inline bool fwd_wants_wide(const gsasr_dims *d)
{
    int want = (d->flags & GSASR_FLAG_FWD_WIDE) ? 1 : (d->flags & GSASR_FLAG_FWD_NARROW) ? 0 : -1;
    if (want < 0) {   // no explicit flag: the registered choice of this shape, then the development switch
        const unsigned rf = registered_choice(d).flags;
        want = (rf & GSASR_FLAG_FWD_WIDE) ? 1 : (rf & GSASR_FLAG_FWD_NARROW) ? 0 : fwd_wide_env();
    }
    if (d->batch > 1 || want == 0) return false;
    if (want == 1) return true;
    const int rows = d->row1 - d->row0;
    const long nsub = (long)((d->w + SUBX - 1) / SUBX) * ((rows + SUBY - 1) / SUBY);
    // (pixels of the WHOLE grid per Gaussian = the scale factor squared: a row band that is handed every Gaussian of the image
    // has the image's window sizes, not those of rows * w / s)
    return nsub >= 2 * 8192 && (double)d->h * (double)d->w >= FWD_WIDE_MIN * (double)d->s;
}

inline int bwd_env()
{
    static std::atomic<int> cached{-1};
    int v = cached.load(std::memory_order_relaxed);
    if (v < 0) {
        const char *e = dev_switch("GSASR_SPLAT_BWD");
        v = !e ? 0 : !strcmp(e, "gaussian") ? 1 : !strcmp(e, "tile") ? 2 : !strcmp(e, "atomic") ? 3 : !strcmp(e, "home") ? 4 : 0;
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
inline int lists_env();                          // (defined with the tile lists below)
inline int list_cap_of(const gsasr_dims *d);

inline bool bwd_wants_tile(const gsasr_dims *d)
{
    if (d->flags & (GSASR_FLAG_FORWARD_ONLY | GSASR_FLAG_BWD_GAUSSIAN | GSASR_FLAG_BWD_ATOMIC | GSASR_FLAG_BWD_HOME)) return false;
    if (d->flags & GSASR_FLAG_BWD_TILE) return true;
    {   // the registered choice of this shape (gsasr_set_kernel_choice)
        const unsigned rf = registered_choice(d).flags;
        if (rf & GSASR_FLAG_BWD_GAUSSIAN) return false;
        if (rf & GSASR_FLAG_BWD_TILE) return true;
    }
    if (bwd_env() == 2) return true;
    // (by default only for whole images: a row band of a sharded image may hold all the Gaussians or just its own, so
    // its pixels per Gaussian say nothing about the window size -- the shard's caller knows the scale and sets the flag)
    if (bwd_env() != 0 || d->batch > 1 || d->row0 != 0 || d->row1 != d->h) return false;
    const double px_per_gaussian = (double)d->h * (double)d->w / (double)(d->s > 0 ? d->s : 1);
    // Round 5: one Gaussian per 2..4 pixels on a megapixel and more (x2 at one per LR pixel, x4 at four) -- a dense plan, whose
    // tile lists this kernel then reads (same 32 x 16-px tiles as the 8 x 16 forward): 1024^2 x2 -16% per step, 2048^2 x2 -21%,
    // 1024^2 x4 at 4 per LR pixel -4%, level at x8 / 16 per LR pixel; 512^2 images lose 4..15% and keep the Gaussian-stationary
    // kernel, as do denser plans (16 per LR pixel at x4: level to +1%).  profiles/history/r05_pxg4.txt; the fused host path has drawn
    // the same line since round 2 (gaussian_splatting._tile_backward).
    if (px_per_gaussian >= 2.0 && px_per_gaussian <= 4.0 && (double)d->h * (double)d->w >= 1048576.0 && list_cap_of(d) >= 0 && lists_env() != 0)
        return true;
    // (round 4, with the windows of the data-derived cutoff: at 2048^2 x8 the Gaussian-stationary kernel is 7% ahead, at
    // 3072^2 x6 the tile-stationary one 3%, from 5120^2 up 4..10%: the line is drawn at 8 Mpx)
    return px_per_gaussian >= 32.0 && (double)d->h * (double)d->w >= 8388608.0;
}

// Tile height of the tile-stationary backward: 32 rows from 64 whole-grid HR pixels per Gaussian (x8 and up: windows of
// 45 px and more) on single images -- the per-tile search is shared by twice the pixels and a window meets 40% fewer tiles
// (slots written, and read by the gather): x8 -2% (the gather 94 -> 68 us, the tile kernel level; HBM traffic 1.72 -> 1.56 GB),
// x12 -2%, x16 -9%, x24 -17%; nothing at x6 (profiles/history/r04_bwd_experiments.txt (6)).  16 rows below that and on the batched
// canvas (slots are multiples of 16 rows).
// development switch: GSASR_SPLAT_BT_TALL=0 / 1
inline bool bt_tall(const gsasr_dims *d)
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

// The home-tile backward by default (round 6, after the backward's own cutoff): whole images and batched canvases denser than one Gaussian per
// two pixels (GSASR's 16 per LR pixel up to x5) with at least 1024 tiles of 32 x 16 px.  Measured against the Gaussian-stationary
// kernel on three boxes (profiles/r06_home_default.txt): 1024^2 at 16 per LR pixel -8..-9% (bench c2x16 backward 362 -> 331 us),
// 1280^2 -8%, 1408^2 -12%, 1536^2 -16%, x3 -15%; 896^2 and 768^2 -7% with one-cell tiles (launch_bwd_home's variant 3), 640^2 level; eight per LR
// pixel -4% but the tile-stationary kernel's range; four per LR pixel +21%, 512^2 +7%: those keep what they had.
inline bool bwd_wants_home(const gsasr_dims *d)
{
    if (d->row0 != 0 || d->row1 != d->h) return false;
    // (a batched canvas counts as the image it is: h = batch x slot rows)
    const long tiles = (long)((d->w + 31) / 32) * (long)((d->h + 15) / 16);
    return (double)d->h * (double)d->w < 2.0 * (double)d->s && tiles >= 1024;
}

inline int bwd_part_k(const gsasr_dims *d)
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
inline int lists_env()
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
// Gaussian per LR pixel the atomics outweigh the search: config 2 +5%, config 3 +20% (profiles/history/r05_lists_ab.txt).
inline bool tl_dense(const gsasr_dims *d)
{
    const double rows = (double)(d->row1 - d->row0 > 0 ? d->row1 - d->row0 : 1);
    return 4.0 * (double)d->s >= (double)d->w * rows;
}

// log2 of the list tiles' height for a plan of these dims: 5 where the forward will be the wide kernel (32 x 32-px tiles),
// 4 for the two-level 8 x 16 kernels (32 x 16), 0 = no lists (small images: the split kernel; list_cap < 0; no Gaussians)
// gsasr_dims.list_cap, or when that is 0 the registered choice of the shape
inline int list_cap_of(const gsasr_dims *d) { return d->list_cap != 0 ? d->list_cap : registered_choice(d).list_cap; }

inline int tl_hlog_for(const gsasr_dims *d)
{
    const int list_cap = list_cap_of(d);
    if (list_cap < 0 || d->s <= 0 || lists_env() == 0) return 0;
    // by default for dense plans (tl_dense) and for plans whose BACKWARD reads them too -- the tile-stationary kernel on the same
    // tiles (x8 and up: config 4, the shard's bands): two consumers pay for k_bin's atomics; an explicit capacity (or the
    // development switch) asks for them anywhere
    // (up to ~x10: at x12 a window meets 16 and more 32 x 32-px tiles and k_bin's appends cost more than the two kernels save --
    // 3072^2 x12 fwd+bwd: plan +33 us for -13 us of forward, profiles/history/r05_policy_sweep.txt)
    const bool both = !(d->flags & GSASR_FLAG_FORWARD_ONLY) && bwd_wants_tile(d) && fwd_wants_wide(d) == bt_tall(d) &&
                      (double)d->h * (double)d->w < 100.0 * (double)d->s;
    if (list_cap == 0 && lists_env() != 1 && !tl_dense(d) && !both) return 0;
    const int rows = d->row1 - d->row0;
    if (fwd_wants_wide(d)) return 5;
    const long nsub = (long)((d->w + SUBX - 1) / SUBX) * ((rows + SUBY - 1) / SUBY);
    return (nsub >= TL_MIN_SUB || list_cap > 0) ? 4 : 0;      // (an explicit capacity asks for lists on any image: tests)
}

// Entries per tile.  gsasr_dims.list_cap when given; else four times what a tile of GSASR-shaped Gaussians (about one LR
// pixel in size: half-extent ~ 2.5 / sqrt(Gaussians per pixel), twice that allowed for) collects, + 64.
inline int tl_cap_for(const gsasr_dims *d, int hlog)
{
    if (!hlog) return 0;
    long cap = list_cap_of(d);
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

// part_k / tl_hlog / tl_cap >= 0: the values the PLAN of this workspace was made with (its note, plan_layout) -- a forward or
// backward must lay the workspace out as its plan did, whatever kernel choice has been registered or cleared for the shape since
inline Layout make_layout(const gsasr_dims *d, int part_k = -1, int tl_hlog = -1, int tl_cap = -1)
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
    L.tl_cap = (tl_cap >= 0 && L.tl_hlog) ? tl_cap : tl_cap_for(d, L.tl_hlog);
    L.tl_ntx = (d->w + TL_W - 1) / TL_W;
    L.tl_ntiles = L.tl_hlog ? L.tl_ntx * ((d->row1 - d->row0 + (1 << L.tl_hlog) - 1) >> L.tl_hlog) : 0;
    L.off_tlc = o;    o += align_up((size_t)L.tl_ntiles * TL_STRIDE * 4, 256);
    L.off_tle = o;    o += align_up((size_t)L.tl_ntiles * (size_t)L.tl_cap * 8, 256);
    L.tl_ok = L.tl_hlog != 0;
    L.total = o;
    return L;
}

inline PlanView make_view(const Layout &L, void *ws, unsigned flags = 0u)
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

// development switch: GSASR_SPLAT_ADAPT=0 keeps the conservative tau = ln(s / eps) in the windows (A/B of adapt_kcut)
inline bool adapt_env()
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
inline float resolve_cutoff(float cutoff, int s)
{
    if (cutoff == 0.f) cutoff = default_cutoff();
    if (cutoff != 0.f) return cutoff;
    const double tau = std::log((double)(s > 1 ? s : 1) / (double)GSASR_SPLAT_DEFAULT_EPS);
    return (float)(tau < 16.0 ? 16.0 : tau > (double)GSASR_SPLAT_EXACT_CUTOFF ? (double)GSASR_SPLAT_EXACT_CUTOFF : tau);
}

// batched canvas whose samples all have one size (the training crops): that size; false otherwise
inline bool batch_uniform(const gsasr_dims *d, int &h, int &w)
{
    h = w = 0;
    if (d->batch <= 1) return false;
    for (int b = 1; b < d->batch; ++b)
        if (d->sample_hw[2 * b] != d->sample_hw[0] || d->sample_hw[2 * b + 1] != d->sample_hw[1]) return false;
    h = d->sample_hw[0];
    w = d->sample_hw[1];
    return true;
}

inline Params make_params(const gsasr_dims *d, const Layout &L)
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
    // the backward's own cutoff (include/gsasr_splat.h: GSASR_SPLAT_GRAD_TAU); development: GSASR_SPLAT_GRADTAU=0 switches it off
    static const bool gradtau_on = !(dev_switch("GSASR_SPLAT_GRADTAU") && atoi(dev_switch("GSASR_SPLAT_GRADTAU")) == 0);
    // (whole images only: "complete to 1e-5 of the Gaussian's own mass" is a statement about all of its pixels -- on a row band
    // that holds nothing but a Gaussian's tail the same truncation is a large part of THAT band's share, and a band's gradient
    // is compared and reduced on its own)
    const bool whole = d->row0 == 0 && d->row1 == d->h;
    P.kb_max = (adapt && gradtau_on && whole) ? (float)(std::sqrt(2.0 * (double)GSASR_SPLAT_GRAD_TAU) * (1.0 + 1e-6)) : 0.f;
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
    constexpr unsigned BWD_ANY = GSASR_FLAG_BWD_GAUSSIAN | GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_ATOMIC | GSASR_FLAG_BWD_HOME;
    if (!(P.flags & BWD_ANY))   // (the registered choice; the development A/B switch)
        P.flags |= registered_choice(d).flags & (GSASR_FLAG_BWD_GAUSSIAN | GSASR_FLAG_BWD_TILE | GSASR_FLAG_BWD_HOME);
    if (!(P.flags & BWD_ANY))
        P.flags |= bwd_env() == 2 ? GSASR_FLAG_BWD_TILE : bwd_env() == 3 ? GSASR_FLAG_BWD_ATOMIC : bwd_env() == 4 ? GSASR_FLAG_BWD_HOME : 0u;
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

// ---- plan notes (splat_api.hip) ----
void note_plan(const void *ws, const gsasr_dims *d, int part_k, int tl_hlog, int tl_cap);
Layout plan_layout(const gsasr_dims *d, const void *ws);     // layout of the plan in `ws`: from the note its plan left, else from these dims
int check_ws(const gsasr_dims *dims, const void *ws, size_t ws_bytes, Layout &L, bool planning = false);

#define HIP_TRY(expr)                                    \
    do {                                                 \
        hipError_t _e = (expr);                          \
        if (_e != hipSuccess) return hip_fail(_e, #expr); \
    } while (0)

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

// ---- pieces shared by several translation units ----
struct BatchSizes {   // kernel argument: the host's per-sample sizes
    unsigned short h[GSASR_MAX_BATCH], w[GSASR_MAX_BATCH];
};

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

typedef float v2f __attribute__((ext_vector_type(2)));

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

// XCD-aware tile order: blocks are dealt round-robin to the 8 XCDs (block b -> XCD b%8), so give each XCD
// a contiguous band of tile rows: neighbouring tiles then share records in ONE L2.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned b, unsigned nb)
{
    const unsigned q = nb >> 3, r = nb & 7u, xcd = b & 7u;
    return xcd * q + min(xcd, r) + (b >> 3);
}

// lane i reads lane i+N of its row of 16 (0 past the row end)
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
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

// ---- host functions that cross translation units ----
// splat_plan.hip
int launch_batch_geo(const gsasr_dims *dims, const PlanView &V, hipStream_t st);
int plan_impl(const float *sigmas, const float *coords, const float *colors, const gsasr_dims *dims, void *workspace,
              size_t workspace_bytes, void *stream, const float *raw, const StepSrc &SS);
// splat_backward.hip
int splat_backward(const float *sigmas, const float *coords, const float *colors, const float *grad_img, float *g_sigmas,
                   float *g_coords, float *g_colors, const gsasr_dims *dims, const void *workspace, size_t workspace_bytes,
                   void *stream, bool gather, int *mode_out);
// splat_backward_home.hip
int launch_bwd_home(const Params &P, const PlanView &V, const float *grad_img, float *g_sigmas, float *g_coords, float *g_colors,
                    int variant, hipStream_t st);
// splat_step.hip
struct StepLayout {
    size_t plan_bytes, off_step, off_sig, off_xy, off_col, off_gsig, off_gxy, off_gcol, off_ghwc, total;
};
StepLayout make_step_layout(const gsasr_dims *d, const void *planned_ws = nullptr);
int step_prologue_plan(const float *gs_parameters, StepSrc SS, const gsasr_dims *dims, void *workspace,
                       size_t workspace_bytes, void *stream, StepLayout &S);
int prologue_backward_batched(const float *gs_parameters, const float *step_size, const gsasr_dims *dims, void *workspace,
                              const float *gs, const float *gc, const float *gk, float *g_parameters, void *stream);

}  // namespace gsasr_detail

#endif  // GSASR_SPLAT_COMMON_H
