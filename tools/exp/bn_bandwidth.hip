// Experiment: what the generic engine's BatchNorm passes reach of the HBM roof at NLSPN's layer sizes (links libptta_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tta-depth-completion_amd/csrc -o /tmp/bn_bandwidth tools/exp/bn_bandwidth.hip \
//         -L tta-depth-completion_amd/proxytta -lptta_hip -Wl,-rpath,'$ORIGIN/../../tta-depth-completion_amd/proxytta'
#include "ptta_common.h"
#include "ptta_kernels.h"
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    struct Case { int B, H, W, C; const char* what; } cases[] = {{2, 352, 1216, 64, "layer1 (train, real+proxy)"}, {1, 352, 1216, 64, "layer1 (one pass)"},
                                                                   {2, 176, 608, 128, "layer2"}, {2, 88, 304, 256, "layer3"}, {2, 44, 152, 512, "layer4"}, {2, 352, 1216, 48, "conv1_rgb"}};
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Case& c : cases) {
        const size_t n = (size_t)c.B * c.H * c.W * c.C;
        float *x, *y, *r, *g, *gx, *gr, *gamma, *beta, *part, *st, *bw, *dg, *db;
        CK(hipMalloc(&x, n * 4)); CK(hipMalloc(&y, n * 4)); CK(hipMalloc(&r, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&gx, n * 4)); CK(hipMalloc(&gr, n * 4));
        CK(hipMalloc(&gamma, c.C * 4)); CK(hipMalloc(&beta, c.C * 4)); CK(hipMalloc(&dg, c.C * 4)); CK(hipMalloc(&db, c.C * 4)); CK(hipMalloc(&bw, 3 * c.C * 4));
        CK(hipMalloc(&st, 8 * c.C * 4)); CK(hipMalloc(&part, (size_t)ptta_gbn_part_floats(c.C, 2) * 4));
        std::vector<float> h(n); for (size_t i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) & 4095) / 2048.f - 1.f;
        CK(hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(r, h.data(), n * 4, hipMemcpyHostToDevice));
        std::vector<float> one(c.C, 1.f); CK(hipMemcpy(gamma, one.data(), c.C * 4, hipMemcpyHostToDevice)); CK(hipMemset(beta, 0, c.C * 4));
        GView vx{x, c.B, c.H, c.W, c.C, c.C}, vy{y, c.B, c.H, c.W, c.C, c.C}, vr{r, c.B, c.H, c.W, c.C, c.C}, none{};
        const int npass = c.B >= 2 ? 2 : 1;
        const int nb = c.B / npass;       // frames of the grad pass
        GView vg{g, nb, c.H, c.W, c.C, c.C}, vgx{gx, nb, c.H, c.W, c.C, c.C}, vgr{gr, nb, c.H, c.W, c.C, c.C};
        auto timeit = [&](const char* nm, double bytes, auto fn) {
            std::vector<float> ts;
            for (int k = 0; k < 12; ++k) { (void)hipEventRecord(e0, s); fn(); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); ts.push_back(ms); }
            std::sort(ts.begin(), ts.end());
            printf("  %-44s %8.1f us  %6.2f TB/s of %7.1f MB\n", nm, ts[ts.size() / 2] * 1e3, bytes / (ts[ts.size() / 2] * 1e-3) / 1e12, bytes / 1e6);
        };
        printf("%s: B %d %dx%d C %d\n", c.what, c.B, c.H, c.W, c.C);
        timeit("forward: stats + finalize + apply", 3.0 * n * 4, [&] { ptta_launch_gbn_forward(vx, none, vy, npass, GACT_RELU, 1e-5f, gamma, beta, part, st, s, 0, 0, nullptr, nullptr, nullptr, nullptr, 0.1f, 1, nullptr); });
        timeit("forward: finalize + apply (stats in the conv)", 2.0 * n * 4, [&] { ptta_launch_gbn_forward(vx, none, vy, npass, GACT_RELU, 1e-5f, gamma, beta, part, st, s, 1024, 0, nullptr, nullptr, nullptr, nullptr, 0.1f, 1, nullptr); });
        timeit("forward: finalize + apply + residual", 3.0 * n * 4, [&] { ptta_launch_gbn_forward(vx, vr, vy, npass, GACT_NONE, 1e-5f, gamma, beta, part, st, s, 1024, 0, nullptr, nullptr, nullptr, nullptr, 0.1f, 1, nullptr); });
        timeit("apply only", 2.0 * n * 4, [&] { ptta_launch_gbn_apply(vx, none, vy, npass, GACT_RELU, st, 1, s, 0); });
        const double nb_ = (double)nb * c.H * c.W * c.C * 4;
        timeit("backward: stats + finalize + apply (relu)", 7.0 * nb_, [&] { ptta_launch_gbn_backward(vx, vg, vy, vgx, none, npass, GACT_RELU, 0, 0, 0, gamma, st, part, bw, dg, db, s, 0, nullptr); });
        timeit("backward: ... + residual gradient", 8.0 * nb_, [&] { ptta_launch_gbn_backward(vx, vg, vy, vgx, vgr, npass, GACT_NONE, 1, 0, 0, gamma, st, part, bw, dg, db, s, 0, nullptr); });
        for (float* p : {x, y, r, g, gx, gr, gamma, beta, part, st, bw, dg, db}) (void)hipFree(p);
    }
    return 0;
}
