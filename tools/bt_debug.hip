// debug harness: counts items / survivors seen by k_render_bwd_tile (build with -DBT_DEBUG)
#include "../gsasr_amd/csrc/gsasr_splat.hip"
#include <vector>
#include <random>
int main(int argc, char **argv)
{
    int lr = argc > 1 ? atoi(argv[1]) : 32;
    const float scale = 4.f;
    const int H = lr * 4, W = lr * 4, n = lr * lr;
    std::mt19937 rng(0);
    std::normal_distribution<float> nd(0.f, 0.5f);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    auto sigm = [](float v) { return 1.f / (1.f + expf(-v)); };
    std::vector<float> sig(3 * n), xy(2 * n), col(3 * n), grad((size_t)H * W * 3);
    const float step = 1.2f / scale;
    for (int k = 0; k < n; ++k) {
        const int i = k / lr, j = k % lr;
        sig[3 * k] = (0.99999f * sigm(nd(rng)) + 1e-6f) / step * 2 / (W - 1);
        sig[3 * k + 1] = (0.99999f * sigm(nd(rng)) + 1e-6f) / step * 2 / (H - 1);
        sig[3 * k + 2] = 0.999999f * tanhf(nd(rng));
        for (int c = 0; c < 3; ++c) col[3 * k + c] = ud(rng);
        xy[2 * k] = ((j + ud(rng)) / lr) * 2 - 1;
        xy[2 * k + 1] = ((i + ud(rng)) / lr) * 2 - 1;
    }
    for (auto &g : grad) g = ud(rng);
    gsasr_dims d{n, H, W, 3, 0.1f, 0, H, 0.f, GSASR_FLAG_OVERWRITE_GRADS | GSASR_FLAG_BWD_TILE};
    const size_t wsb = gsasr_splat_workspace_bytes(&d);
    float *dsig, *dxy, *dcol, *dgrad, *dgs, *dgc, *dgk;
    void *ws;
    hipMalloc(&dsig, sig.size() * 4); hipMalloc(&dxy, xy.size() * 4); hipMalloc(&dcol, col.size() * 4);
    hipMalloc(&dgrad, grad.size() * 4); hipMalloc(&dgs, sig.size() * 4); hipMalloc(&dgc, xy.size() * 4); hipMalloc(&dgk, col.size() * 4);
    hipMalloc(&ws, wsb);
    hipMemcpy(dsig, sig.data(), sig.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dxy, xy.data(), xy.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dgrad, grad.data(), grad.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        gsasr_splat_plan(dsig, dxy, dcol, &d, ws, wsb, nullptr);
        Layout L = make_layout(&d);
        PlanView V = make_view(L, ws);
        hipMemset(V.done, 0, 32);
        gsasr_splat_backward(dsig, dxy, dcol, dgrad, dgs, dgc, dgk, &d, ws, wsb, nullptr);
        hipDeviceSynchronize();
        unsigned c[8];
        hipMemcpy(c, V.done, 32, hipMemcpyDeviceToHost);
        printf("items %u processed %u | survivors %u heads %u | unwritten %u\n", c[0], c[1], c[2], c[3], c[4]);
    }
    return 0;
}
