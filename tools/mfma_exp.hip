// mfma_exp.hip -- micro-benchmark: the forward's exponent on the MFMA pipe.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_exp.hip -o tools/bin/mfma_exp && tools/bin/mfma_exp
//
// For a block of 32 pixels x 32 Gaussians the exponent  E = -U^2 - Bq^2,  U = IX dx,  Bq = IY dy + NR U  (k_bin's completed
// square, log2 units) is a rank-6 bilinear form in coordinates LOCAL to the wave's sub-tile:  with xi = px - pxc, eta = py - pyc
// (pixel side) and xg = x - pxc, yg = y - pyc (Gaussian side), a = IX, b = IY, c = NR IX, u0 = a xg, q0 = b yg + c xg:
//     E = [1, xi, eta, xi^2, xi eta, eta^2] . [-(u0^2 + q0^2), 2 (a u0 + c q0), 2 b q0, -(a^2 + c^2), -2 b c, -b^2]
// i.e. three v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 64 cycles each on the matrix pipe, which issues beside the VALU)
// replace the ~6 VALU instructions per pair that form the exponent; the VALU keeps v_exp_f32 and the three colour FMAs.
//
// Three evaluators of the SAME work (one wave = one 8 x 16-px sub-tile, NG Gaussians staged in LDS, REP repetitions):
//   v0  fwd_eval_one of gsasr_splat.hip (rounds 1-4): two PIXELS per lane packed, 12 VALU + 2 v_exp_f32 per record
//   v1  record-PAIR packed (round 5, fwd_eval_pair): 16 packed + 4 v_exp_f32 per record pair
//   v2  MFMA exponent: per group of 32 Gaussians the lanes build the coefficient operand from the staged records, then per
//       32-pixel block 3 MFMA + 16 v_exp_f32 + 24 v_pk_fma_f32 (colours from a transposed LDS copy)
// Reported: ns per 1024 (Gaussian, pixel) pairs per SIMD-equivalent, speed-ups, and the error of each against an fp64 evaluation of
// the same sums, on GSASR-shaped Gaussians (sigma 0.8..2.8 px, |rho| <= 0.9, centres within 20 px of the sub-tile) and on a
// hard set (sigma 0.15..0.5 px hairlines, |rho| up to 0.999).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

constexpr int NG = 256;          // Gaussians per wave (multiple of 32)

struct Rec { float x, y, IX, NR, IY, r, g, b; };

// ---- v0: two pixels per lane ------------------------------------------------------------------------
__device__ __forceinline__ void eval_one(const float4 a, const float4 b, float px, v2f py, v2f &ar, v2f &ag, v2f &ab)
{
    const float dx = px - a.x;
    const v2f dy = py - a.y;
    const float u = a.z * dx;
    const float k0 = -u * u, ru = a.w * u;
    const v2f bq = b.x * dy + ru;
    const v2f pw = k0 - bq * bq;
    const v2f v = {__builtin_amdgcn_exp2f(pw.x), __builtin_amdgcn_exp2f(pw.y)};
    ar += v * b.y;
    ag += v * b.z;
    ab += v * b.w;
}

__global__ __launch_bounds__(256) void k_v0(const Rec *__restrict__ recs, const float *__restrict__ pxt, const float *__restrict__ pyt,
                                            float *__restrict__ out, int rep)
{
    __shared__ float4 st[4][2 * NG];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float4 *src = reinterpret_cast<const float4 *>(recs);
    for (int i = lane; i < 2 * NG; i += 64) st[wv][i] = src[i];
    __builtin_amdgcn_wave_barrier();
    const float px = pxt[lane & 7];
    const v2f py = {pyt[lane >> 3], pyt[(lane >> 3) + 8]};
    v2f ar = {0.f, 0.f}, ag = ar, ab = ar;
    for (int r = 0; r < rep; ++r) {
        for (int i = 0; i < NG; i += 2) {
            const float4 a0 = st[wv][2 * i], b0 = st[wv][2 * i + 1], a1 = st[wv][2 * i + 2], b1 = st[wv][2 * i + 3];
            eval_one(a0, b0, px, py, ar, ag, ab);
            eval_one(a1, b1, px, py, ar, ag, ab);
        }
    }
    float *o = out + ((size_t)blockIdx.x * 4 + wv) * 384;
    o[lane * 3 + 0] = ar.x; o[lane * 3 + 1] = ag.x; o[lane * 3 + 2] = ab.x;
    o[192 + lane * 3 + 0] = ar.y; o[192 + lane * 3 + 1] = ag.y; o[192 + lane * 3 + 2] = ab.y;
}

// ---- v1: record pairs ----------------------------------------------------------------------------------
__device__ __forceinline__ void eval_pair(const float4 q0, const float4 q1, const float4 q2, const float4 q3, float px, v2f py, v2f (&acc)[6])
{
    const v2f x = {q0.x, q0.y}, y = {q0.z, q0.w}, ix = {q1.x, q1.y}, nr = {q1.z, q1.w}, iy = {q2.x, q2.y};
    const v2f cr = {q2.z, q2.w}, cg = {q3.x, q3.y}, cb = {q3.z, q3.w};
    const v2f dx = px - x;
    const v2f u = ix * dx;
    const v2f k0 = -u * u, ru = nr * u;
    const v2f dyA = py.x - y, dyB = py.y - y;
    const v2f bqA = iy * dyA + ru, bqB = iy * dyB + ru;
    const v2f pwA = k0 - bqA * bqA, pwB = k0 - bqB * bqB;
    const v2f vA = {__builtin_amdgcn_exp2f(pwA.x), __builtin_amdgcn_exp2f(pwA.y)};
    const v2f vB = {__builtin_amdgcn_exp2f(pwB.x), __builtin_amdgcn_exp2f(pwB.y)};
    acc[0] += vA * cr; acc[1] += vA * cg; acc[2] += vA * cb;
    acc[3] += vB * cr; acc[4] += vB * cg; acc[5] += vB * cb;
}

__global__ __launch_bounds__(256) void k_v1(const Rec *__restrict__ recs, const float *__restrict__ pxt, const float *__restrict__ pyt,
                                            float *__restrict__ out, int rep)
{
    __shared__ float4 st[4][2 * NG];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *sf = reinterpret_cast<float *>(st[wv]);
    for (int i = lane; i < NG; i += 64) {      // interleave pairs: {x0,x1,y0,y1,IX0,IX1,NR0,NR1 | IY0,IY1,r0,r1,g0,g1,b0,b1}
        const Rec q = recs[i];
        float *p = sf + (i >> 1) * 16 + (i & 1);
        p[0] = q.x; p[2] = q.y; p[4] = q.IX; p[6] = q.NR; p[8] = q.IY; p[10] = q.r; p[12] = q.g; p[14] = q.b;
    }
    __builtin_amdgcn_wave_barrier();
    const float px = pxt[lane & 7];
    const v2f py = {pyt[lane >> 3], pyt[(lane >> 3) + 8]};
    v2f acc[6];
    for (int k = 0; k < 6; ++k) acc[k] = (v2f){0.f, 0.f};
    for (int r = 0; r < rep; ++r) {
        for (int i = 0; i < NG / 2; i += 2) {
            const float4 *q = st[wv] + 4 * i;
            eval_pair(q[0], q[1], q[2], q[3], px, py, acc);
            eval_pair(q[4], q[5], q[6], q[7], px, py, acc);
        }
    }
    float *o = out + ((size_t)blockIdx.x * 4 + wv) * 384;
    o[lane * 3 + 0] = acc[0].x + acc[0].y; o[lane * 3 + 1] = acc[1].x + acc[1].y; o[lane * 3 + 2] = acc[2].x + acc[2].y;
    o[192 + lane * 3 + 0] = acc[3].x + acc[3].y; o[192 + lane * 3 + 1] = acc[4].x + acc[4].y; o[192 + lane * 3 + 2] = acc[5].x + acc[5].y;
}

// ---- v2: exponent by MFMA ------------------------------------------------------------------------------
// C/D layout of v_mfma_f32_32x32x2_f32: lane l, register r: column j = l & 31, row i = (r & 3) + 8 (r >> 2) + 4 (l >> 5).
// Here rows = Gaussians of the group, columns = pixels of the block: A = coefficients [32 Gaussians x 2 k], B = monomials
// [2 k x 32 pixels]: a lane ends with 16 Gaussians' exponents for ONE pixel.
__global__ __launch_bounds__(256) void k_v2(const Rec *__restrict__ recs, const float *__restrict__ pxt, const float *__restrict__ pyt,
                                            float *__restrict__ out, int rep)
{
    __shared__ float4 st[4][2 * NG];
    __shared__ __attribute__((aligned(16))) float colT[4][3][NG];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, half = lane >> 5, j = lane & 31;
    const float4 *src = reinterpret_cast<const float4 *>(recs);
    for (int i = lane; i < 2 * NG; i += 64) st[wv][i] = src[i];
    for (int i = lane; i < NG; i += 64) {
        const Rec q = recs[i];
        colT[wv][0][i] = q.r; colT[wv][1][i] = q.g; colT[wv][2][i] = q.b;
    }
    __builtin_amdgcn_wave_barrier();
    // monomials of this lane's pixel in each of the four 8 x 4-px blocks (pixel j of block b: column j & 7, row 4 b + (j >> 3)),
    // relative to the sub-tile's centre pixel (column 4, row 8); lane half h holds k = 2 kp + h
    const float pxc = pxt[4], pyc = pyt[8];
    float mono[4][3];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const float xi = pxt[j & 7] - pxc, eta = pyt[4 * b + (j >> 3)] - pyc;
        mono[b][0] = half ? xi : 1.f;            // k = 0 | 1
        mono[b][1] = half ? xi * xi : eta;       // k = 2 | 3
        mono[b][2] = half ? eta * eta : xi * eta;  // k = 4 | 5
    }
    v2f acc[4][3];
#pragma unroll
    for (int b = 0; b < 4; ++b)
        for (int c = 0; c < 3; ++c) acc[b][c] = (v2f){0.f, 0.f};
    for (int r = 0; r < rep; ++r) {
        for (int g0 = 0; g0 < NG; g0 += 32) {
            // coefficient operand of Gaussian g0 + j: [-(u0^2 + q0^2), 2 (a u0 + c q0), 2 b q0, -(a^2 + c^2), -2 b c, -b^2]
            const float4 ra = st[wv][2 * (g0 + j)], rb = st[wv][2 * (g0 + j) + 1];
            const float xg = ra.x - pxc, yg = ra.y - pyc, a = ra.z, bb = rb.x, c = ra.w * ra.z;
            const float u0 = a * xg, q0 = bb * yg + c * xg;
            const float k0 = -(u0 * u0 + q0 * q0), kX = 2.f * (a * u0 + c * q0), kY = 2.f * bb * q0;
            const float kXX = -(a * a + c * c), kXY = -2.f * bb * c, kYY = -bb * bb;
            const float c0 = half ? kX : k0, c1 = half ? kXX : kY, c2 = half ? kYY : kXY;
            // colours of this lane's 16 Gaussians (rows (r & 3) + 8 (r >> 2) + 4 half of the group): 4 consecutive per read
            float4 cr[4], cg[4], cb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int g = g0 + 8 * t + 4 * half;
                cr[t] = *reinterpret_cast<const float4 *>(&colT[wv][0][g]);
                cg[t] = *reinterpret_cast<const float4 *>(&colT[wv][1][g]);
                cb[t] = *reinterpret_cast<const float4 *>(&colT[wv][2][g]);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                f16v d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(c0, mono[b][0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(c1, mono[b][1], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(c2, mono[b][2], d, 0, 0, 0);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const v2f v01 = {__builtin_amdgcn_exp2f(d[4 * t + 0]), __builtin_amdgcn_exp2f(d[4 * t + 1])};
                    const v2f v23 = {__builtin_amdgcn_exp2f(d[4 * t + 2]), __builtin_amdgcn_exp2f(d[4 * t + 3])};
                    acc[b][0] += v01 * (v2f){cr[t].x, cr[t].y}; acc[b][0] += v23 * (v2f){cr[t].z, cr[t].w};
                    acc[b][1] += v01 * (v2f){cg[t].x, cg[t].y}; acc[b][1] += v23 * (v2f){cg[t].z, cg[t].w};
                    acc[b][2] += v01 * (v2f){cb[t].x, cb[t].y}; acc[b][2] += v23 * (v2f){cb[t].z, cb[t].w};
                }
            }
        }
    }
    // pixel j of block b: the two lane halves hold the two halves of its Gaussians
    float *o = out + ((size_t)blockIdx.x * 4 + wv) * 384;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s = acc[b][c].x + acc[b][c].y;
            s += __shfl_xor(s, 32);
            // same output order as v0 / v1: lane L = column + 8 (row & 7), rows 0..7 first, then rows 8..15
            const int row = 4 * b + (j >> 3), L = (j & 7) + 8 * (row & 7);
            if (!half) o[(row >> 3) * 192 + L * 3 + c] = s;
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------
static void reference(const std::vector<Rec> &g, const std::vector<float> &px, const std::vector<float> &py, std::vector<double> &ref,
                      std::vector<double> &mag)
{
    ref.assign(384, 0.0);
    mag.assign(384, 0.0);
    for (int L = 0; L < 64; ++L)
        for (int h = 0; h < 2; ++h) {
            const double X = px[L & 7], Y = py[(L >> 3) + 8 * h];
            for (const Rec &q : g) {
                const double dx = X - (double)q.x, dy = Y - (double)q.y, u = (double)q.IX * dx, bq = (double)q.IY * dy + (double)q.NR * u;
                const double v = std::exp2(-u * u - bq * bq);
                const double c[3] = {q.r, q.g, q.b};
                for (int k = 0; k < 3; ++k) { ref[h * 192 + L * 3 + k] += v * c[k]; mag[h * 192 + L * 3 + k] += std::fabs(v * c[k]); }
            }
        }
}

int main(int argc, char **argv)
{
    const int W = 1024, H = 1024;             // the image the sub-tile sits in (config 2); sub-tile at (504, 496)
    const int X0 = 504, Y0 = 496;
    std::vector<float> px(8), py(16);
    for (int i = 0; i < 8; ++i) px[i] = (float)(2.0 * (X0 + i) / (W - 1) - 1.0);
    for (int i = 0; i < 16; ++i) py[i] = (float)(2.0 * (Y0 + i) / (H - 1) - 1.0);
    float *d_px, *d_py, *d_out;
    Rec *d_rec;
    const int blocks = 2048;                   // 8192 waves: 8 per SIMD
    CHECK(hipMalloc(&d_px, 32));
    CHECK(hipMalloc(&d_py, 64));
    CHECK(hipMalloc(&d_out, (size_t)blocks * 4 * 384 * 4));
    CHECK(hipMalloc(&d_rec, NG * sizeof(Rec)));
    CHECK(hipMemcpy(d_px, px.data(), 32, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_py, py.data(), 64, hipMemcpyHostToDevice));
    const double hl = 0.5 * 1.4426950408889634074;
    for (int hard = 0; hard < 2; ++hard) {
        std::mt19937 rng(1234 + hard);
        std::uniform_real_distribution<double> U(0.0, 1.0);
        std::vector<Rec> g(NG);
        for (Rec &q : g) {
            const double sx_px = hard ? 0.15 + 0.35 * U(rng) : 0.8 + 2.0 * U(rng), sy_px = hard ? 0.15 + 0.35 * U(rng) : 0.8 + 2.0 * U(rng);
            const double rho = (hard ? 0.999 : 0.9) * (2.0 * U(rng) - 1.0);
            const double reach = hard ? 4.0 : 20.0;
            const double cx = X0 + 4 + reach * (2.0 * U(rng) - 1.0), cy = Y0 + 8 + reach * (2.0 * U(rng) - 1.0);
            const double sx = sx_px * 2.0 / (W - 1), sy = sy_px * 2.0 / (H - 1), cinv = 1.0 / (1.0 - rho * rho);
            q.x = (float)(2.0 * cx / (W - 1) - 1.0);
            q.y = (float)(2.0 * cy / (H - 1) - 1.0);
            q.IX = (float)(std::sqrt(hl) / sx);
            q.IY = (float)(std::sqrt(hl * cinv) / sy);
            q.NR = (float)(-rho * std::sqrt(cinv));
            q.r = (float)U(rng); q.g = (float)U(rng); q.b = (float)U(rng);
        }
        CHECK(hipMemcpy(d_rec, g.data(), NG * sizeof(Rec), hipMemcpyHostToDevice));
        std::vector<double> ref, mag;
        reference(g, px, py, ref, mag);
        double top = 0;
        for (double v : ref) top = std::fmax(top, std::fabs(v));
        printf("%s set: %d Gaussians, largest pixel value %.3f\n", hard ? "HARD (hairlines, |rho| <= 0.999)" : "GSASR-shaped", NG, top);
        const int rep = 64;
        double t_ns[3];
        for (int v = 0; v < 3; ++v) {
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            auto launch = [&](int r) {
                if (v == 0) hipLaunchKernelGGL(k_v0, dim3(blocks), dim3(256), 0, 0, d_rec, d_px, d_py, d_out, r);
                else if (v == 1) hipLaunchKernelGGL(k_v1, dim3(blocks), dim3(256), 0, 0, d_rec, d_px, d_py, d_out, r);
                else hipLaunchKernelGGL(k_v2, dim3(blocks), dim3(256), 0, 0, d_rec, d_px, d_py, d_out, r);
            };
            launch(1);
            CHECK(hipDeviceSynchronize());
            std::vector<float> got(384);
            CHECK(hipMemcpy(got.data(), d_out, 384 * 4, hipMemcpyDeviceToHost));
            double err = 0, rel = 0;
            for (int i = 0; i < 384; ++i) {
                err = std::fmax(err, std::fabs(got[i] - ref[i]));
                rel = std::fmax(rel, std::fabs(got[i] - ref[i]) / std::fmax(mag[i], 1e-30));
            }
            launch(rep);
            CHECK(hipDeviceSynchronize());
            float best = 1e30f;
            for (int it = 0; it < 5; ++it) {
                CHECK(hipEventRecord(e0));
                launch(rep);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = std::fmin(best, ms);
            }
            const double pairs = (double)blocks * 4 * 128.0 * NG * rep;
            t_ns[v] = best * 1e6 / (pairs / 1024.0) * 1024.0;      // ns per 1024 pairs on ONE SIMD-equivalent (1024 SIMDs busy)
            printf("  v%d %-28s %8.3f ms  %7.1f G pairs/s  %6.1f ns (= cycles at 1 GHz; x clock) per 1024 pairs per SIMD   max |err| %.2e (%.2e of the pixel's sum of |terms|)\n",
                   v, v == 0 ? "pixel-packed (rounds 1-4)" : v == 1 ? "record-pair packed" : "MFMA exponent", best,
                   pairs / (best * 1e-3) / 1e9, t_ns[v], err, rel);
        }
        printf("  speed-up of v1 over v0 %.2fx, of v2 over v0 %.2fx, of v2 over v1 %.2fx\n", t_ns[0] / t_ns[1], t_ns[0] / t_ns[2], t_ns[1] / t_ns[2]);
    }
    return 0;
}
