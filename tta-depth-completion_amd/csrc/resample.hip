// Resolution changes of the MSG_CHN cascade that are not convolutions
// (network_exp_msg_chn_adapt.py:487-501): sparse-aware average pooling of the depth input,
// bilinear x2 (align_corners=True) of the 1-channel predictions, and the transposes of the
// bilinear x2 used by the backward pass (1-channel planar and 32-channel NHWC).  The forward
// bilinear x2 of 32-channel maps is fused into the producing conv's epilogue (ptta_common.h).
#include "ptta_common.h"
#include "ptta_kernels.h"

// One thread per 4x4 input block: clamp (external_model_adapt.py:108), 1/2 and 1/4 pooled maps
// d_s = avg_pool(d, k) / (avg_pool(d > 0, k) + 1e-4)  (network_exp_msg_chn_adapt.py:487,492).
__global__ void prep_kernel(const float* __restrict__ sparse, float max_d, float* __restrict__ dclamp,
                            float* __restrict__ d12, float* __restrict__ d14, int N, int H, int W) {
    const int H4 = H >> 2, W4 = W >> 2;
    const long total = (long)N * H4 * W4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x4 = (int)(idx % W4);
        long t_ = idx / W4;
        const int y4 = (int)(t_ % H4);
        const int n = (int)(t_ / H4);
        const float* src = sparse + (size_t)n * H * W;
        float* dc = dclamp + (size_t)n * H * W;
        float s4 = 0.f, c4 = 0.f;
        float s2[4] = {0.f, 0.f, 0.f, 0.f}, c2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
            const int y = 4 * y4 + dy;
            float4 v = *(const float4*)(src + (size_t)y * W + 4 * x4);
            float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int dx = 0; dx < 4; ++dx) {
                float d = a[dx];
                if (max_d >= 0.f) d = fminf(fmaxf(d, 0.f), max_d);
                a[dx] = d;
                const float c = d > 0.f ? 1.f : 0.f;
                s4 += d; c4 += c;
                const int q = (dy >> 1) * 2 + (dx >> 1);
                s2[q] += d; c2[q] += c;
            }
            *(float4*)(dc + (size_t)y * W + 4 * x4) = make_float4(a[0], a[1], a[2], a[3]);
        }
        d14[idx] = (s4 / 16.f) / (c4 / 16.f + 0.0001f);
        const int H2 = H >> 1, W2 = W >> 1;
        float* o2 = d12 + (size_t)n * H2 * W2;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o2[(size_t)(2 * y4 + (q >> 1)) * W2 + 2 * x4 + (q & 1)] = (s2[q] / 4.f) / (c2[q] / 4.f + 0.0001f);
    }
}

int ptta_launch_prep(const float* sparse, float max_input_depth, float* dclamp, float* d12, float* d14,
                     int N, int H, int W, hipStream_t s) {
    const long total = (long)N * (H / 4) * (W / 4);
    int blocks = (int)((total + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(prep_kernel, dim3(blocks), dim3(256), 0, s, sparse, max_input_depth, dclamp, d12, d14, N, H, W);
    PTTA_CHECK_LAUNCH();
    return 0;
}

__global__ void up2_1ch_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int Hin, int Win) {
    // one block row = one output row (grid: x blocks, rows, images): the row's taps and both source-row pointers are scalar, a lane's
    // offsets 32-bit.  (The flat 64-bit index form spent ~120 quarter-rate integer multiplies per thread on div / mod and addresses: this
    // launch is two of the latency-bound steps of the real chain.)
    const int Ho = 2 * Hin, Wo = 2 * Win;
    const float sy = up_scale(Hin, Ho), sx = up_scale(Win, Wo);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= Wo) return;
    const Lerp ly = lerp_coef(y, Hin, sy), lx = lerp_coef(x, Win, sx);
    const float* r0 = in + ((size_t)b * Hin + ly.i0) * Win;
    const float* r1 = in + ((size_t)b * Hin + ly.i1) * Win;
    out[((size_t)b * Ho + y) * Wo + x] = ly.l0 * (lx.l0 * r0[lx.i0] + lx.l1 * r0[lx.i1]) + ly.l1 * (lx.l0 * r1[lx.i0] + lx.l1 * r1[lx.i1]);
}

int ptta_launch_up2_1ch(const float* in, float* out, int B, int Hin, int Win, hipStream_t s) {
    hipLaunchKernelGGL(up2_1ch_kernel, dim3((2 * Win + 255) / 256, 2 * Hin, B), dim3(256), 0, s, in, out, B, Hin, Win);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// Transposed bilinear x2 as a GATHER (deterministic, no atomics): the gradient of source index i
// collects every destination d in [2i-3, 2i+3] whose lerp taps touch i.
__device__ __forceinline__ void up2T_weights(int i, int in_size, float scale, int& d0, float w[7]) {
    d0 = 2 * i - 3;
    const int out_size = 2 * in_size;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int d = d0 + k;
        float ww = 0.f;
        if (d >= 0 && d < out_size) {
            const Lerp l = lerp_coef(d, in_size, scale);
            if (l.i0 == i) ww += l.l0;
            if (l.i1 == i) ww += l.l1;
        }
        w[k] = ww;
    }
}

// The destinations whose taps touch source index i lie in the open interval ((i - 1) / s, (i + 1) / s), s = (in - 1) / (2 in - 1): at most
// FIVE consecutive integers.  up2T_window returns the first of them (clamped so that d0 .. d0 + 4 stays inside [0, 2 in)) and their five
// weights -- the same values up2T_weights computes, zero where a candidate does not touch i.  With a fixed 5 x 5 window every load of the
// gather is UNCONDITIONAL: behind `if (w != 0) acc += w * g[...]` hipcc issued each of the ~14 touching taps as its own load + s_waitcnt
// vmcnt(0) (round 5, .s of both kernels: up to 49 dependent round trips per thread).
__device__ __forceinline__ void up2T_window(int i, int in_size, float scale, int& d0, float w[5]) {
    int c0; float w7[7];
    up2T_weights(i, in_size, scale, c0, w7);
    int k0 = 0;
    if (w7[0] == 0.f) { k0 = 1; if (w7[1] == 0.f) k0 = 2; }            // first touching candidate (the touching ones are consecutive)
    const int out_size = 2 * in_size;
    int first = c0 + k0;
    first = max(min(first, out_size - 5), 0);                          // (maps narrower than five: the loads clamp their offsets)
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int c = first + k - c0;                                   // index into the seven candidates
        float ww = 0.f;
#pragma unroll
        for (int q = 0; q < 7; ++q) ww = (q == c) ? w7[q] : ww;
        w[k] = ww;
    }
    d0 = first;
}

__global__ void up2T_1ch_kernel(const float* __restrict__ gout, float* __restrict__ gin, int B, int Hin, int Win) {
    // one block row = one source row (as up2_1ch_kernel): the row window and its weights are scalar
    const int Wo = 2 * Win, Ho = 2 * Hin;
    const float sy = up_scale(Hin, Ho), sx = up_scale(Win, Wo);
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= Win) return;
    int dy0, dx0; float wy[5], wx[5];
    up2T_window(y, Hin, sy, dy0, wy);
    up2T_window(x, Win, sx, dx0, wx);
    const float* g = gout + ((size_t)b * Ho + dy0) * Wo;
    float v[5][5];
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        const float* gr = g + (size_t)min(ky, Ho - 1 - dy0) * Wo;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) v[ky][kx] = gr[dx0 + min(kx, Wo - 1 - dx0)];
    }
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        float row = 0.f;
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) row += wx[kx] * v[ky][kx];
        acc += wy[ky] * row;
    }
    gin[((size_t)b * Hin + y) * Win + x] = acc;
}

int ptta_launch_up2T_1ch(const float* gout, float* gin, int B, int Hin, int Win, hipStream_t s) {
    hipLaunchKernelGGL(up2T_1ch_kernel, dim3((Win + 255) / 256, Hin, B), dim3(256), 0, s, gout, gin, B, Hin, Win);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// 32-channel NHWC version: one thread per (pixel, 4 channels); gin = add + up2^T(gout).
template <typename T>
__global__ void up2T_32_kernel(const T* __restrict__ gout, const T* __restrict__ add, T* __restrict__ gin,
                               int B, int Hin, int Win) {
    // one block row = one source row (as up2T_1ch_kernel): row window, row weights and row pointers are scalar, a lane's offsets 32-bit
    const int Wo = 2 * Win, Ho = 2 * Hin;
    const float sy = up_scale(Hin, Ho), sx = up_scale(Win, Wo);
    const int t = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    const int cq = t & 7, x = t >> 3;
    if (x >= Win) return;
    int dy0, dx0; float wy[5], wx[5];
    up2T_window(y, Hin, sy, dy0, wy);
    up2T_window(x, Win, sx, dx0, wx);
    const T* g = gout + ((size_t)b * Ho + dy0) * Wo * 32;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // row by row: the five taps of a row are in flight together (one 16-B / 8-B load per tap: the channel quad is aligned)
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        const T* gr = g + (size_t)min(ky, Ho - 1 - dy0) * Wo * 32;
        float4 v[5];
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const T* q = gr + (unsigned)((dx0 + min(kx, Wo - 1 - dx0)) * 32 + 4 * cq);
            if constexpr (sizeof(T) == 4) v[kx] = *(const float4*)q;
            else v[kx] = bf4_to_f4(*(const uint2*)q);
        }
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
            const float wgt = wy[ky] * wx[kx];
            acc[0] += wgt * v[kx].x; acc[1] += wgt * v[kx].y; acc[2] += wgt * v[kx].z; acc[3] += wgt * v[kx].w;
        }
    }
    const size_t o = (((size_t)b * Hin + y) * Win + x) * 32 + 4 * cq;
    if constexpr (sizeof(T) == 4) {
        float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (add) { const float4 a = *(const float4*)(add + o); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        *(float4*)(gin + o) = v;
    } else {
        float4 v = make_float4(acc[0], acc[1], acc[2], acc[3]);
        if (add) { const float4 a = bf4_to_f4(*(const uint2*)(add + o)); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        *(uint2*)(gin + o) = f4_to_bf4(v);
    }
}

int ptta_launch_up2T_32(const void* gout, const void* add, void* gin, int B, int Hin, int Win, int bf16, hipStream_t s) {
    const dim3 grid((Win * 8 + 255) / 256, Hin, B);
    if (bf16) hipLaunchKernelGGL((up2T_32_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)gout, (const bf16_t*)add, (bf16_t*)gin, B, Hin, Win);
    else hipLaunchKernelGGL((up2T_32_kernel<float>), grid, dim3(256), 0, s, (const float*)gout, (const float*)add, (float*)gin, B, Hin, Win);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// ---- OutlierRemoval.remove_outliers (src/net_utils.py:766-811), the sparse-depth filter that runs
// on-device before every forward (src/tta_main.py:590,:703) ---------------------------------------
//   max_value = 10 * max(sparse);  filled = validity <= 0 ? max_value : sparse  (padded with max_value)
//   min_k x k(filled) < sparse - threshold  =>  point removed
// One fused stencil replaces the reference's 6 ATen kernels (where, pad, neg, max_pool2d, where, 2 muls);
// the global max is a two-stage reduction (block partials, then every block re-reduces the <=1024
// partials, which is cheaper than a third launch).
__global__ __launch_bounds__(256) void outlier_max_kernel(const float* __restrict__ sparse, long n, float* __restrict__ part) {
    __shared__ float red[4];
    float m = -INFINITY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, sparse[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(256) void outlier_removal_kernel(const float* __restrict__ sparse, const float* __restrict__ validity,
                                                              const float* __restrict__ part, int nparts, float* __restrict__ sparse_out,
                                                              float* __restrict__ validity_out, int N, int H, int W, int ksize, float threshold) {
    __shared__ float smax;
    {
        float m = -INFINITY;
        for (int k = threadIdx.x; k < nparts; k += blockDim.x) m = fmaxf(m, part[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        __shared__ float red[4];
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) smax = 10.0f * fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
    }
    const float max_value = smax;
    const int pad = ksize / 2;
    const long total = (long)N * H * W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const long t_ = idx / W;
        const int y = (int)(t_ % H);
        const long base = (t_ / H) * (long)H * W;
        float mn = INFINITY;
        for (int dy = -pad; dy <= pad; ++dy) {
            const int yy = y + dy;
            for (int dx = -pad; dx <= pad; ++dx) {
                const int xx = x + dx;
                float f = max_value;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const long q = base + (long)yy * W + xx;
                    f = validity[q] <= 0.f ? max_value : sparse[q];
                }
                mn = fminf(mn, f);
            }
        }
        const float sd = sparse[idx], v = validity[idx];
        const float clean = (mn < sd - threshold) ? 0.f : 1.f;
        const float vo = v * clean;
        validity_out[idx] = vo;
        sparse_out[idx] = sd * vo;
    }
}

int ptta_launch_outlier_removal(const float* sparse, const float* validity, float* sparse_out, float* validity_out,
                                int N, int H, int W, int ksize, float threshold, float* scratch, hipStream_t s) {
    const long total = (long)N * H * W;
    int nparts = (int)((total + 255) / 256); if (nparts > 1024) nparts = 1024;
    hipLaunchKernelGGL(outlier_max_kernel, dim3(nparts), dim3(256), 0, s, sparse, total, scratch);
    int blocks = (int)((total + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(outlier_removal_kernel, dim3(blocks), dim3(256), 0, s, sparse, validity, scratch, nparts, sparse_out,
                       validity_out, N, H, W, ksize, threshold);
    PTTA_CHECK_LAUNCH();
    return 0;
}
