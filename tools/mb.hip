// tools/mb.hip -- standalone micro-benchmark of libgsasr_splat's kernels (development aid, not product).
//
//   bash tools/build_mb.sh [name] [-DVARIANT... | -DGSASR_SRC='"/tmp/variant.hip"']     -> tools/bin/mb[_name]
//   tools/bin/mb [lr_h lr_w scale dmax tau iters gpp flags]
//
// It #includes the library SOURCE (it does not link libgsasr_splat.so: LD_LIBRARY_PATH tricks do nothing, the
// binary must be rebuilt after every kernel change -- tools/build_mb.sh uses the library's own flags) so that
// -D switches or an alternative source file can select kernel variants, generates
// GSASR-shaped Gaussians (LR raster + jitter, sigmoid/tanh activations) with its own
// RNG, and times plan / forward / backward with hipEvents on one stream.
#ifdef GSASR_SRC
#include GSASR_SRC
#else
#include "../gsasr_amd/csrc/gsasr_splat.hip"
#endif

#include <algorithm>
#include <random>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

template <class F>
static float time_us(F &&f, int iters, hipStream_t st)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipStreamSynchronize(st));
    std::vector<float> ts;
    for (int i = 0; i < iters; ++i) {
        CK(hipEventRecord(a, st));
        f();
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main(int argc, char **argv)
{
    int lrh = argc > 1 ? atoi(argv[1]) : 256, lrw = argc > 2 ? atoi(argv[2]) : 256;
    float scale = argc > 3 ? atof(argv[3]) : 4.f, dmax = argc > 4 ? atof(argv[4]) : 0.1f;
    float tau = argc > 5 ? atof(argv[5]) : 0.f;
    int iters = argc > 6 ? atoi(argv[6]) : 30, gpp = argc > 7 ? atoi(argv[7]) : 1;
    const int H = (int)lroundf(lrh * scale), W = (int)lroundf(lrw * scale), n = lrh * lrw * gpp;
    std::mt19937 rng(0);
    std::normal_distribution<float> nd(0.f, 0.5f);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    auto sigm = [](float v) { return 1.f / (1.f + expf(-v)); };
    std::vector<float> sig(3 * (size_t)n), xy(2 * (size_t)n), col(3 * (size_t)n), grad((size_t)H * W * 3);
    const float step = 1.2f / scale;
    // MB_DIST: which Gaussians (tools/policy_sweep.sh).  0 = SURVEY 8(d)'s (sigma = sigmoid(N(0, 0.5)) of the decoder's range);
    // 1 small (0.1..0.3 of the range); 2 saturated (0.85..1); 3 log-uniform over 0.05..1; 4 needles (0.9 x 0.1, |rho| 0.9);
    // 5 bimodal (90% at 0.15, 10% at 0.95); 6 = 0 with the means jittered over +-3 LR pixels (clusters and holes)
    const int dist = getenv("MB_DIST") ? atoi(getenv("MB_DIST")) : 0;
    std::mt19937 rng2(1);
    for (int k = 0; k < n; ++k) {
        const int i = k / (lrw * gpp), j = (k / gpp) % lrw;
        float sx = 0.99999f * sigm(nd(rng)) + 1e-6f, sy = 0.99999f * sigm(nd(rng)) + 1e-6f;
        float rho = 0.999999f * tanhf(nd(rng)), jit = 1.f;
        const float r1 = ud(rng2), r2 = ud(rng2), r3 = ud(rng2);   // (own stream: MB_DIST=0 draws what it always drew)
        switch (dist) {
        case 1: sx = 0.1f + 0.2f * r1; sy = 0.1f + 0.2f * r2; break;
        case 2: sx = 0.85f + 0.15f * r1; sy = 0.85f + 0.15f * r2; break;
        case 3: sx = 0.05f * powf(20.f, r1); sy = 0.05f * powf(20.f, r2); break;
        case 4: sx = r3 < 0.5f ? 0.9f : 0.1f; sy = r3 < 0.5f ? 0.1f : 0.9f; rho = (r1 < 0.5f ? -0.9f : 0.9f) * (0.5f + 0.5f * r2); break;
        case 5: sx = sy = r3 < 0.9f ? 0.15f : 0.95f; break;
        case 6: jit = 6.f; break;
        default: break;
        }
        sig[3 * k + 0] = sy / step * 2 / (W - 1);
        sig[3 * k + 1] = sx / step * 2 / (H - 1);
        sig[3 * k + 2] = rho;
        const float a = sigm(nd(rng));
        for (int c = 0; c < 3; ++c) col[3 * k + c] = sigm(nd(rng)) * a;
        const float mx = ((j + 0.5f + jit * (ud(rng) - 0.5f)) / lrw) * 2 - 1, my = ((i + 0.5f + jit * (ud(rng) - 0.5f)) / lrh) * 2 - 1;
        xy[2 * k + 0] = (mx + 1 - 1.f / W) * W / (W - 1) - 1.f;
        xy[2 * k + 1] = (my + 1 - 1.f / H) * H / (H - 1) - 1.f;
    }
    for (auto &g : grad) g = ud(rng);
    if (getenv("MB_SHUFFLE") && atoi(getenv("MB_SHUFFLE"))) {   // the Gaussians in random order instead of the decoder's raster order
        std::vector<int> perm(n);
        for (int k = 0; k < n; ++k) perm[k] = k;
        std::shuffle(perm.begin(), perm.end(), rng2);
        std::vector<float> s2(sig.size()), x2(xy.size()), c2(col.size());
        for (int k = 0; k < n; ++k) {
            for (int c = 0; c < 3; ++c) { s2[3 * k + c] = sig[3 * (size_t)perm[k] + c]; c2[3 * k + c] = col[3 * (size_t)perm[k] + c]; }
            x2[2 * k] = xy[2 * (size_t)perm[k]]; x2[2 * k + 1] = xy[2 * (size_t)perm[k] + 1];
        }
        sig.swap(s2); xy.swap(x2); col.swap(c2);
    }

    const unsigned flags = argc > 8 ? (unsigned)atoi(argv[8]) : 0u;   // 6 = overwrite image + grads
    gsasr_dims d{n, H, W, 3, dmax, 0, H, tau, flags};
    if (getenv("MB_LIST_CAP")) d.list_cap = atoi(getenv("MB_LIST_CAP"));   // > 0: tile lists of that capacity on any image; < 0: none
    const size_t wsb = gsasr_splat_workspace_bytes(&d);
    float *dsig, *dxy, *dcol, *dgrad, *dimg, *dgs, *dgc, *dgk;
    void *ws;
    CK(hipMalloc(&dsig, sig.size() * 4)); CK(hipMalloc(&dxy, xy.size() * 4)); CK(hipMalloc(&dcol, col.size() * 4));
    CK(hipMalloc(&dgrad, grad.size() * 4)); CK(hipMalloc(&dimg, grad.size() * 4));
    CK(hipMalloc(&dgs, sig.size() * 4)); CK(hipMalloc(&dgc, xy.size() * 4)); CK(hipMalloc(&dgk, col.size() * 4));
    CK(hipMalloc(&ws, wsb));
    CK(hipMemcpy(dsig, sig.data(), sig.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dxy, xy.data(), xy.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgrad, grad.data(), grad.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dimg, 0, grad.size() * 4));
    CK(hipMemset(dgs, 0, sig.size() * 4)); CK(hipMemset(dgc, 0, xy.size() * 4)); CK(hipMemset(dgk, 0, col.size() * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto chk = [](int rc, const char *w) { if (rc) { fprintf(stderr, "%s: %d %s\n", w, rc, gsasr_last_error()); exit(1); } };
    chk(gsasr_splat_plan(dsig, dxy, dcol, &d, ws, wsb, st), "plan");
    // (as bench.py and the Python pool plan: the workspace persists, every plan zeroes the other parity's counters on the side)
    int nplans = 1;
    const float t_plan = time_us([&] {
        gsasr_dims dp = d;
        dp.flags |= GSASR_FLAG_COUNTERS_CLEAN | ((++nplans & 1) ? 0u : GSASR_FLAG_PARITY);
        chk(gsasr_splat_plan(dsig, dxy, dcol, &dp, ws, wsb, st), "plan"); }, iters, st);
    const float t_fwd = time_us([&] { chk(gsasr_splat_forward(&d, ws, wsb, dimg, st), "fwd"); }, iters, st);
    const bool fwd_only = (flags & GSASR_FLAG_FORWARD_ONLY) != 0u;      // (flags 70 = inference: a forward-only plan, no backward timed)
    const float t_bwd = fwd_only ? 0.f : time_us([&] { chk(gsasr_splat_backward(dsig, dxy, dcol, dgrad, dgs, dgc, dgk, &d, ws, wsb, st), "bwd"); }, iters, st);
    CK(hipStreamSynchronize(st));
    // checksums so variants can be compared for (approximate) equality
    std::vector<float> img(grad.size()), gs(sig.size());
    CK(hipMemset(dimg, 0, grad.size() * 4)); CK(hipMemset(dgs, 0, sig.size() * 4));
    CK(hipMemset(dgc, 0, xy.size() * 4)); CK(hipMemset(dgk, 0, col.size() * 4));
    chk(gsasr_splat_forward(&d, ws, wsb, dimg, st), "fwd");
    if (!fwd_only) chk(gsasr_splat_backward(dsig, dxy, dcol, dgrad, dgs, dgc, dgk, &d, ws, wsb, st), "bwd");
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(img.data(), dimg, img.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gs.data(), dgs, gs.size() * 4, hipMemcpyDeviceToHost));
    double si = 0, sg = 0;
    for (float v : img) si += v;
    for (float v : gs) sg += fabs(v);
    // MB_ALT_FLAGS=<flags>: the backward again with these flags OR-ed in (another kernel on the same plan), timed and compared
    // element by element with the first one: worst difference relative to the tensor's max-abs and to the Gaussian's own row
    if (getenv("MB_ALT_FLAGS") && !fwd_only) {
        gsasr_dims d2 = d;
        d2.flags |= (unsigned)atoi(getenv("MB_ALT_FLAGS"));
        std::vector<float> gc(xy.size()), gk(col.size()), gs2(sig.size()), gc2(xy.size()), gk2(col.size());
        CK(hipMemcpy(gc.data(), dgc, gc.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gk.data(), dgk, gk.size() * 4, hipMemcpyDeviceToHost));
        const float t_alt = time_us([&] { chk(gsasr_splat_backward(dsig, dxy, dcol, dgrad, dgs, dgc, dgk, &d2, ws, wsb, st), "bwd alt"); }, iters, st);
        CK(hipMemset(dgs, 0, sig.size() * 4)); CK(hipMemset(dgc, 0, xy.size() * 4)); CK(hipMemset(dgk, 0, col.size() * 4));
        chk(gsasr_splat_backward(dsig, dxy, dcol, dgrad, dgs, dgc, dgk, &d2, ws, wsb, st), "bwd alt");
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(gs2.data(), dgs, gs2.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gc2.data(), dgc, gc2.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gk2.data(), dgk, gk2.size() * 4, hipMemcpyDeviceToHost));
        auto cmp = [&](const std::vector<float> &a, const std::vector<float> &b, int w, const char *name) {
            double mx = 0, worst = 0, worst_row = 0;
            long bad = -1, nan = 0;
            for (float v : a) mx = std::max(mx, (double)fabs(v));
            for (size_t k = 0; k < a.size() / w; ++k) {
                double rm = 0, re = 0;
                for (int c = 0; c < w; ++c) {
                    rm = std::max(rm, (double)fabs(a[k * w + c]));
                    const double e = fabs((double)a[k * w + c] - (double)b[k * w + c]);
                    if (!(e == e)) ++nan;
                    re = std::max(re, e);
                }
                worst = std::max(worst, re);
                const double rr = re / (rm + 1e-5 * mx + 1e-30);
                if (rr > worst_row) { worst_row = rr; bad = (long)k; }
            }
            printf("  alt %s: max|a|=%.4e worst abs diff / max = %.3e, worst row-relative = %.3e (Gaussian %ld) nan=%ld\n", name, mx, worst / (mx + 1e-30), worst_row, bad, nan);
        };
        printf("alt flags %u: bwd %.1f us (first %.1f us)\n", d2.flags, t_alt, t_bwd);
        cmp(gs, gs2, 3, "g_sigmas"); cmp(gc, gc2, 2, "g_coords"); cmp(gk, gk2, 3, "g_colors");
    }
    unsigned hdr[16];
    CK(hipMemcpy(hdr, ws, 64, hipMemcpyDeviceToHost));
    float tau_w;
    memcpy(&tau_w, &hdr[4], 4);
    printf("N=%d %dx%d dmax=%g tau=%g | plan %.1f us  fwd %.1f us  bwd %.1f us | sum(img)=%.6e sum|gs|=%.6e | rx=%u ry=%u reach=%u,%u maxcell=%u tau'=%.3f K=%u near-dead=%u\n", n, H, W,
           dmax, tau, t_plan, t_fwd, t_bwd, si, sg, hdr[0], hdr[1], hdr[8], hdr[9], hdr[2], tau_w, hdr[5], hdr[7]);
    return 0;
}
