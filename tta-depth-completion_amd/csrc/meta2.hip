// Building blocks of the `2layers` meta layer conv1_rgb_meta = Res_Conv(32,128)
// (network_exp_msg_chn_adapt.py:28-36, built at :1073-1077, used by bash/adapt/adapt_msgchn_vkitti.sh):
//   x -> conv3x3(32->128, no bias) -> BatchNorm2d(128) -> LeakyReLU(0.2) -> conv3x3(128->32, bias)
//     -> BatchNorm2d(32) -> + x
// The two convolutions run on the 32->32 matrix-core kernel, four 32-channel groups at a time
// (the 128-channel hidden map is kept as four NHWC-32 tensors); this file holds what is left:
// train-mode BatchNorm2d statistics (per-channel, deterministic two-stage reductions), the fused
// normalise(+LeakyReLU)(+residual) passes and the BatchNorm/LeakyReLU backward passes.
// All of it runs at 1/4 resolution (the meta layer sits on enc_c[2]).
#include "ptta_common.h"
#include "ptta_kernels.h"

#define CS_BLOCKS 128       // partial blocks per statistics launch

// Per-channel partial sums over items [b0, b1) of a [B][H][W][32] tensor.
//   MODE 0: s1 = sum x,  s2 = sum x^2
//   MODE 1: g1 = g * lrelu'(x*fscale+fshift)   (slope < 0: no activation, g1 = g)
//           s1 = sum g1, s2 = sum g1 * (x - mean) * inv
// part layout: [CS_BLOCKS][2][32]
template <typename T, int MODE>
__global__ __launch_bounds__(256) void chan_stats32_kernel(const T* __restrict__ x, const T* __restrict__ g, long pix0, long npix,
                                                           const float* __restrict__ fscale, const float* __restrict__ fshift,
                                                           const float* __restrict__ mean, const float* __restrict__ inv,
                                                           float slope, float* __restrict__ part) {
    __shared__ float red[8][2][32];
    const int ch = threadIdx.x & 31, sub = threadIdx.x >> 5;            // 8 pixels per block-iteration
    float fs = 0.f, fh = 0.f, mu = 0.f, iv = 0.f;
    if (MODE == 1) { mu = mean[ch]; iv = inv[ch]; if (slope >= 0.f) { fs = fscale[ch]; fh = fshift[ch]; } }
    float s1 = 0.f, s2 = 0.f;
    for (long p = (long)blockIdx.x * 8 + sub; p < npix; p += (long)gridDim.x * 8) {
        const float xv = ld(x + (pix0 + p) * 32 + ch);
        if (MODE == 0) { s1 += xv; s2 += xv * xv; }
        else {
            float gv = ld(g + (pix0 + p) * 32 + ch);
            if (slope >= 0.f) gv = (fmaf(xv, fs, fh) > 0.f) ? gv : gv * slope;
            s1 += gv; s2 += gv * (xv - mu) * iv;
        }
    }
    red[sub][0][ch] = s1; red[sub][1][ch] = s2;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int which = threadIdx.x >> 5;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[k][which][ch];
        part[((long)blockIdx.x * 2 + which) * 32 + ch] = v;
    }
}

int ptta_launch_chan_stats32(const void* x, const void* g, int bf16, long pix0, long npix, const float* fscale,
                             const float* fshift, const float* mean, const float* inv, float slope, float* part, hipStream_t s) {
#define L_(T, M) hipLaunchKernelGGL((chan_stats32_kernel<T, M>), dim3(CS_BLOCKS), dim3(256), 0, s, (const T*)x, (const T*)g, pix0, npix, fscale, fshift, mean, inv, slope, part)
    if (g) { if (bf16) L_(bf16_t, 1); else L_(float, 1); }
    else { if (bf16) L_(bf16_t, 0); else L_(float, 0); }
#undef L_
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_chan_stats_blocks() { return CS_BLOCKS; }

// BatchNorm2d in eval mode: scale/shift from the running statistics.
__global__ void bn_eval_affine_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = sc; shift[c] = beta[c] - rm[c] * sc;
}
int ptta_launch_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale, float* shift, int C, hipStream_t s) {
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 63) / 64), dim3(64), 0, s, gamma, beta, rm, rv, eps, scale, shift, C);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// y = act(x*scale[pass][ch] + shift[pass][ch]) (+ res);  pass = b / items_per_pass; slope < 0: no activation
template <typename T>
__global__ void bn_apply32_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, long npix, long pix_per_pass,
                                  const float* __restrict__ scale, const float* __restrict__ shift, int stat_stride, float slope) {
    const long total = npix * 32;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(k & 31);
        const int pass = (int)((k >> 5) / pix_per_pass);
        float v = fmaf(ld(x + k), scale[pass * stat_stride + ch], shift[pass * stat_stride + ch]);
        if (slope >= 0.f) v = v > 0.f ? v : v * slope;
        if (res) v += ld(res + k);
        st(y + k, v);
    }
}
int ptta_launch_bn_apply32(const void* x, const void* res, void* y, int bf16, long npix, long pix_per_pass, const float* scale,
                           const float* shift, int stat_stride, float slope, hipStream_t s) {
    long b = (npix * 32 + 255) / 256; if (b > 4096) b = 4096;
    if (bf16) hipLaunchKernelGGL((bn_apply32_kernel<bf16_t>), dim3((int)b), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)res, (bf16_t*)y, npix, pix_per_pass, scale, shift, stat_stride, slope);
    else hipLaunchKernelGGL((bn_apply32_kernel<float>), dim3((int)b), dim3(256), 0, s, (const float*)x, (const float*)res, (float*)y, npix, pix_per_pass, scale, shift, stat_stride, slope);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// BatchNorm backward (train mode) through an optional LeakyReLU:
//   g1 = g * lrelu'(x*fscale+fshift) ; dx = gscale * (g1 - c1 - (x-mean)*inv * c2)
template <typename T>
__global__ void bn_bwd_apply32_kernel(const T* __restrict__ x, const T* __restrict__ g, T* __restrict__ dx, long npix,
                                      const float* __restrict__ fscale, const float* __restrict__ fshift, const float* __restrict__ mean,
                                      const float* __restrict__ inv, const float* __restrict__ gscale, const float* __restrict__ c1,
                                      const float* __restrict__ c2, float slope) {
    const long total = npix * 32;
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(k & 31);
        const float xv = ld(x + k);
        float gv = ld(g + k);
        if (slope >= 0.f) gv = (fmaf(xv, fscale[ch], fshift[ch]) > 0.f) ? gv : gv * slope;
        st(dx + k, gscale[ch] * (gv - c1[ch] - (xv - mean[ch]) * inv[ch] * c2[ch]));
    }
}
int ptta_launch_bn_bwd_apply32(const void* x, const void* g, void* dx, int bf16, long npix, const float* fscale, const float* fshift,
                               const float* mean, const float* inv, const float* gscale, const float* c1, const float* c2,
                               float slope, hipStream_t s) {
    long b = (npix * 32 + 255) / 256; if (b > 4096) b = 4096;
    if (bf16) hipLaunchKernelGGL((bn_bwd_apply32_kernel<bf16_t>), dim3((int)b), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)g, (bf16_t*)dx, npix, fscale, fshift, mean, inv, gscale, c1, c2, slope);
    else hipLaunchKernelGGL((bn_bwd_apply32_kernel<float>), dim3((int)b), dim3(256), 0, s, (const float*)x, (const float*)g, (float*)dx, npix, fscale, fshift, mean, inv, gscale, c1, c2, slope);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// From the MODE-1 partials: BatchNorm parameter gradients and the constants of the input gradient.
//   dbeta = sum g1 ; dgamma = sum g1*xhat ; c1 = dbeta/R ; c2 = dgamma/R ; gscale = gamma*inv
// one wave per channel: lane l sums partial blocks l, l+64, ... in a fixed order, then xor-shuffles
__global__ __launch_bounds__(256) void bn2d_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, long R, const float* __restrict__ gamma,
                                                                const float* __restrict__ inv, float* dgamma, float* dbeta, float* gscale, float* c1, float* c2) {
    const int ch = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (ch >= 32) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = lane; b < nblocks; b += 64) { s1 += (double)part[((long)b * 2 + 0) * 32 + ch]; s2 += (double)part[((long)b * 2 + 1) * 32 + ch]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
    if (lane) return;
    if (dbeta) dbeta[ch] = (float)s1;
    if (dgamma) dgamma[ch] = (float)s2;
    c1[ch] = (float)(s1 / (double)R); c2[ch] = (float)(s2 / (double)R);
    gscale[ch] = gamma[ch] * inv[ch];
}
int ptta_launch_bn2d_bwd_finalize(const float* part, int nblocks, long R, const float* gamma, const float* inv, float* dgamma,
                                  float* dbeta, float* gscale, float* c1, float* c2, hipStream_t s) {
    hipLaunchKernelGGL(bn2d_bwd_finalize_kernel, dim3(8), dim3(256), 0, s, part, nblocks, R, gamma, inv, dgamma, dbeta, gscale, c1, c2);
    PTTA_CHECK_LAUNCH();
    return 0;
}
