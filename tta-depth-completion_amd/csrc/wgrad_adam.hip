// Weight gradient of the adapted meta layer conv1_rgb_meta = Conv2d(32,32,3,1,1)
// (network_exp_msg_chn_adapt.py:1065-1071) and the fused Adam update (torch.optim.Adam,
// src/tta_main.py:341-346,633).  Only the adapted parameters get a weight gradient: the reference's
// autograd computes (and DDP all-reduces) weight gradients for all 1.46 M parameters and throws
// them away.
//
// wgrad: dW[co][ci][tap] = sum_pixels gy[pix][co] * x[pix+tap][ci]  ->  per wave nine 32x32 fp32
// MFMA accumulators (one per tap) + one for the bias, K = pixels, two pixels per MFMA; each wave
// reduces a contiguous chunk of pixels and a second, fixed-order pass sums the chunk partials.
#include "ptta_common.h"
#include "ptta_kernels.h"

#define WGRAD_MAX_CHUNKS 512
#define WGRAD_PART (10 * 1024)       // floats per chunk: 9 taps * 32 * 32 + 32 bias (padded)

int ptta_wgrad_chunks(long pixels) {
    long c = (pixels + 63) / 64;
    return (int)(c > WGRAD_MAX_CHUNKS ? WGRAD_MAX_CHUNKS : (c < 1 ? 1 : c));
}

template <typename T>
__global__ __launch_bounds__(64) void wgrad32_kernel(const T* __restrict__ x, const T* __restrict__ gy, int B, int H, int W,
                                                     int nchunks, float* __restrict__ part) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const long P = (long)B * H * W;
    const long per = ((P + nchunks - 1) / nchunks + 7) & ~7L;          // multiple of 8 pixels per chunk
    const long p0 = (long)blockIdx.x * per;
    long p1 = p0 + per; if (p1 > P) p1 = P;
    f32x16 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (long pp = p0; pp < p1; pp += 8) {
        // four pixel pairs per iteration, all 40 loads issued before the first MFMA
        float a[4], bv[4][9];
        bool pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long pix = pp + 2 * u + h;
            pv[u] = pix < p1;
            int px = 0, py = 0, pb = 0;
            a[u] = 0.f;
            if (pv[u]) {
                px = (int)(pix % W); const long t_ = pix / W; py = (int)(t_ % H); pb = (int)(t_ / H);
                a[u] = ld(gy + pix * 32 + i);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
                bv[u][tap] = 0.f;
                if (pv[u] && yy >= 0 && yy < H && xx >= 0 && xx < W) bv[u][tap] = ld(x + (((long)pb * H + yy) * W + xx) * 32 + i);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u][tap], acc[tap], 0, 0, 0);
            acc[9] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], pv[u] ? 1.f : 0.f, acc[9], 0, 0, 0);
        }
    }
    float* out = part + (long)blockIdx.x * WGRAD_PART;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(tap * 32 + acc_row(r, h)) * 32 + i] = acc[tap][r];
    if (i == 0)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[9 * 1024 + acc_row(r, h)] = acc[9][r];
}

// 64 outputs per block, 8 chunk groups summed in parallel then combined through LDS (fixed order)
__global__ __launch_bounds__(512) void wgrad32_reduce_kernel(const float* __restrict__ part, int nchunks, float* __restrict__ gw, float* __restrict__ gb, int co_stride) {
    __shared__ float red[8][64];
    const int ol = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int o = blockIdx.x * 64 + ol;
    float s = 0.f;
    if (o < 9 * 1024 + 32)
        for (int c = cg; c < nchunks; c += 8) s += part[(long)c * WGRAD_PART + o];
    red[cg][ol] = s;
    __syncthreads();
    if (cg != 0 || o >= 9 * 1024 + 32) return;
    s = ((red[0][ol] + red[1][ol]) + (red[2][ol] + red[3][ol])) + ((red[4][ol] + red[5][ol]) + (red[6][ol] + red[7][ol]));
    if (o < 9 * 1024) {
        const int ci = o & 31, co = (o >> 5) & 31, tap = o >> 10;
        gw[(size_t)co * co_stride + ci * 9 + tap] = s;        // co_stride = 9 * (input channels of the full weight)
    } else if (gb) {
        gb[o - 9 * 1024] = s;
    }
}

int ptta_launch_wgrad32(const void* x, const void* gy, int bf16, int B, int H, int W, float* part,
                        float* gw, float* gb, hipStream_t s, int co_stride) {
    const int nchunks = ptta_wgrad_chunks((long)B * H * W);
    if (bf16) hipLaunchKernelGGL((wgrad32_kernel<bf16_t>), dim3(nchunks), dim3(64), 0, s, (const bf16_t*)x, (const bf16_t*)gy, B, H, W, nchunks, part);
    else hipLaunchKernelGGL((wgrad32_kernel<float>), dim3(nchunks), dim3(64), 0, s, (const float*)x, (const float*)gy, B, H, W, nchunks, part);
    hipLaunchKernelGGL(wgrad32_reduce_kernel, dim3((9 * 1024 + 32 + 63) / 64), dim3(512), 0, s, part, nchunks, gw, gb, co_stride);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// torch.optim.Adam single-tensor update: g += wd*p; m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  The step count lives on the device so the
// whole TTA step can be replayed from a hipGraph.
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, const float* __restrict__ g,
                            long n, const float* __restrict__ hyper, const int* __restrict__ step) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
    const int t = *step;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    const float step_size = (float)((double)lr / bc1);
    const float bc2s = (float)sqrt(bc2);
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
        float pk = p[k], mk = m[k], vk = v[k];
        ptta_adam_update(pk, mk, vk, g[k], wd, b1, b2, eps, step_size, bc2s);
        m[k] = mk; v[k] = vk; p[k] = pk;
    }
}

// All adapted tensors in ONE launch (NLSPN adapts 88 tensors, CostDCNet 32: one ~5 us launch each otherwise), the step
// count incremented by the last block to finish: t = *step + 1 is what every block uses, *step = t is written once all
// blocks are done reading it.
__global__ __launch_bounds__(256) void adam_multi_kernel(const PttaAdamEntry* __restrict__ tab, int nt, long total, const float* __restrict__ hyper,
                                                         int* step, unsigned* ticket) {
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4];
    const int t = *step + 1;
    const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
    const float step_size = (float)((double)lr / bc1);
    const float bc2s = (float)sqrt(bc2);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nt - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].off <= idx) lo = mid; else hi = mid - 1; }
        const PttaAdamEntry e = tab[lo];
        if (e.rep == 0) continue;                           // listed, never given a gradient (torch.optim.Adam skips grad = None)
        const long k = idx - e.off;
        const float g0 = e.g[k];
        float pk = e.p[k], mk = e.m[k], vk = e.v[k];
        if (e.rep <= 1) {
            ptta_adam_update(pk, mk, vk, g0, wd, b1, b2, eps, step_size, bc2s);
        } else {
            // a tensor listed `rep` times: `rep` consecutive updates with the same gradient, step counts (t - 1) * rep + 1 ... t * rep
            for (int r = 0; r < e.rep; ++r) {
                const int tt = (t - 1) * e.rep + r + 1;
                const double c1 = 1.0 - pow((double)b1, (double)tt), c2 = 1.0 - pow((double)b2, (double)tt);
                ptta_adam_update(pk, mk, vk, g0, wd, b1, b2, eps, (float)((double)lr / c1), (float)sqrt(c2));
            }
        }
        e.m[k] = mk; e.v[k] = vk; e.p[k] = pk;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(ticket, 1u);
        if (done == gridDim.x - 1) { *step = t; *ticket = 0u; }
    }
}
int ptta_launch_adam_multi(const PttaAdamEntry* tab_dev, int nt, long total, const float* hyper, int* step_dev, unsigned* ticket_dev, hipStream_t s) {
    long blocks = (total + 255) / 256; if (blocks > 256) blocks = 256; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((int)blocks), dim3(256), 0, s, tab_dev, nt, total, hyper, step_dev, ticket_dev);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

__global__ void step_inc_kernel(int* step) { if (threadIdx.x == 0 && blockIdx.x == 0) *step += 1; }

// Small host constants travel as KERNEL ARGUMENTS (copied at launch), so setting hyper-parameters, loss weights or the
// Adam step count never synchronises the stream and never reads host memory after the call returns.
struct F8 { float v[8]; };
__global__ void set_floats_kernel(float* dst, F8 src, int n) { if ((int)threadIdx.x < n) dst[threadIdx.x] = src.v[threadIdx.x]; }
__global__ void set_int_kernel(int* dst, int v) { if (threadIdx.x == 0) *dst = v; }
int ptta_launch_set_floats(float* dst, const float* host_src, int n, hipStream_t s) {
    if (n < 0 || n > 8) return -22;
    F8 f{};
    for (int i = 0; i < n; ++i) f.v[i] = host_src[i];
    hipLaunchKernelGGL(set_floats_kernel, dim3(1), dim3(64), 0, s, dst, f, n);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
int ptta_launch_set_int(int* dst, int v, hipStream_t s) {
    hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(64), 0, s, dst, v);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

int ptta_launch_step_inc(int* step_dev, hipStream_t s) {
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(64), 0, s, step_dev);
    PTTA_CHECK_LAUNCH();
    return 0;
}

int ptta_launch_adam(float* p, float* m, float* v, const float* g, long n, const float* hyper, const int* step_dev, hipStream_t s) {
    int blocks = (int)((n + 255) / 256); if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, p, m, v, g, n, hyper, step_dev);
    PTTA_CHECK_LAUNCH();
    return 0;
}
