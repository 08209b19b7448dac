// Generic NHWC layers for the NLSPN backbone: convolutions with arbitrary channel counts
// (3/1 -> 48/16, ResNet34 64..512, decoder concatenations 768/384/192/128/96, heads 1/8/24), stride 1/2, 3x3 or
// 1x1, and the 3x3 stride-2 transposed convolution -- external_src/NLSPN/src/model/common.py:45-80 and
// nlspnmodel_adapt.py:59-116,385-452.  Tensors are strided NHWC views (pixel stride `ld`, channel offset folded into
// the pointer) so torch.cat along channels (nlspnmodel_adapt.py:474-490) never materialises: producers write into a
// channel slice of the wider buffer.
//
// Every data gradient is itself one of these two kernels with re-packed weights (gpack):
//   conv s1      -> conv s1 with flipped taps and swapped channel roles
//   conv s2      -> transposed conv with the same weight tensor
//   convT s2     -> conv s2 with the same weight tensor
// Kernels here: the reference-grade direct form (exact fp32 FMA, one thread per output element), used for the small
// layers and as the on-device check of the matrix-core path (gconv_mfma.hip).
#include "ptta_common.h"
#include "ptta_kernels.h"

// P[t][a][b] = src[a * a_stride + b * b_stride + (flip ? KK-1-t : t)]
__global__ void gpack_kernel(const float* __restrict__ src, float* __restrict__ dst, int KK, int A, int B, long a_stride,
                             long b_stride, int flip) {
    const long total = (long)KK * A * B;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx % B); long t_ = idx / B;
        const int a = (int)(t_ % A); const int t = (int)(t_ / A);
        dst[idx] = src[(long)a * a_stride + (long)b * b_stride + (flip ? KK - 1 - t : t)];
    }
}
void ptta_gpack(const float* src, float* dst, int KK, int A, int B, long a_stride, long b_stride, int flip, hipStream_t s) {
    const long total = (long)KK * A * B;
    long blocks = (total + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gpack_kernel, dim3((int)blocks), dim3(256), 0, s, src, dst, KK, A, B, a_stride, b_stride, flip);
}

__device__ __forceinline__ float g_act(float v, int act) {
    if (act == GACT_RELU) return v > 0.f ? v : 0.f;
    if (act == GACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == GACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// y[b][oy][ox][co] (+)= act(bias[co] + sum_{tap,ci} x[b][oy*s+ky-pad][ox*s+kx-pad][ci] * w[tap][ci][co])
__global__ __launch_bounds__(256) void gconv_direct_kernel(GConvArgs a) {
    const int Ho = a.y.H, Wo = a.y.W, Co = a.y.C, Ci = a.x.C, k = a.k, pad = a.k >> 1, st = a.stride;
    const long wld = a.wld ? a.wld : Co, wts = a.wts ? a.wts : (long)Ci * Co;
    const long total = (long)a.y.B * Ho * Wo * Co;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Co); long t_ = idx / Co;
        const int ox = (int)(t_ % Wo); t_ /= Wo;
        const int oy = (int)(t_ % Ho); const int b = (int)(t_ / Ho);
        float* yp = a.y.p + (((long)b * Ho + oy) * Wo + ox) * a.y.ld + co;
        float acc = a.bias ? a.bias[co] : 0.f;
        if (a.accumulate) acc += *yp;
        for (int ky = 0; ky < k; ++ky) {
            const int iy = oy * st + ky - pad;
            if (iy < 0 || iy >= a.x.H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int ix = ox * st + kx - pad;
                if (ix < 0 || ix >= a.x.W) continue;
                const float* xp = a.x.p + (((long)b * a.x.H + iy) * a.x.W + ix) * a.x.ld;
                const float* wp = a.w + (long)(ky * k + kx) * wts + co;
                for (int ci = 0; ci < Ci; ++ci) acc = fmaf(xp[ci], wp[(long)ci * wld], acc);
            }
        }
        *yp = g_act(acc, a.act);
    }
}

// transposed conv, stride st, padding pad = k/2 (3x3: output_padding 1; 1x1: pad 0):
// y[b][oy][ox][co] (+)= sum over (ky,kx,ci) with oy = iy*st - pad + ky of x[b][iy][ix][ci] * w[tap][ci][co]
__global__ __launch_bounds__(256) void gconvT_direct_kernel(GConvArgs a) {
    const int Ho = a.y.H, Wo = a.y.W, Co = a.y.C, Ci = a.x.C, k = a.k, pad = a.k >> 1, st = a.stride;
    const long wld = a.wld ? a.wld : Co, wts = a.wts ? a.wts : (long)Ci * Co;
    const long total = (long)a.y.B * Ho * Wo * Co;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Co); long t_ = idx / Co;
        const int ox = (int)(t_ % Wo); t_ /= Wo;
        const int oy = (int)(t_ % Ho); const int b = (int)(t_ / Ho);
        float* yp = a.y.p + (((long)b * Ho + oy) * Wo + ox) * a.y.ld + co;
        float acc = a.bias ? a.bias[co] : 0.f;
        if (a.accumulate) acc += *yp;
        for (int ky = 0; ky < k; ++ky) {
            const int ny = oy + pad - ky;
            if (ny < 0 || (ny % st) != 0) continue;
            const int iy = ny / st;
            if (iy >= a.x.H) continue;
            for (int kx = 0; kx < k; ++kx) {
                const int nx = ox + pad - kx;
                if (nx < 0 || (nx % st) != 0) continue;
                const int ix = nx / st;
                if (ix >= a.x.W) continue;
                const float* xp = a.x.p + (((long)b * a.x.H + iy) * a.x.W + ix) * a.x.ld;
                const float* wp = a.w + (long)(ky * k + kx) * wts + co;
                for (int ci = 0; ci < Ci; ++ci) acc = fmaf(xp[ci], wp[(long)ci * wld], acc);
            }
        }
        *yp = g_act(acc, a.act);
    }
}

int ptta_launch_gconv_direct(const GConvArgs& a, hipStream_t s) {
    if (a.k != 1 && a.k != 3) return -22;
    if (a.stride != 1 && a.stride != 2) return -22;
    const long total = (long)a.y.B * a.y.H * a.y.W * a.y.C;
    long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536; if (blocks < 1) blocks = 1;
    if (a.transposed) hipLaunchKernelGGL(gconvT_direct_kernel, dim3((int)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gconv_direct_kernel, dim3((int)blocks), dim3(256), 0, s, a);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// g (in place) *= act'(y): relu / leaky-relu from the sign of the output, sigmoid y(1-y); clamp(min=0) is relu
__global__ void gact_bwd_kernel(GView g, GView y, int act) {
    const long total = (long)g.B * g.H * g.W * g.C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % g.C); const long pix = idx / g.C;
        const float yv = y.p[pix * y.ld + c];
        float* gp = g.p + pix * g.ld + c;
        float d = 1.f;
        if (act == GACT_RELU) d = yv > 0.f ? 1.f : 0.f;
        else if (act == GACT_LRELU) d = yv > 0.f ? 1.f : 0.2f;
        else if (act == GACT_SIGMOID) d = yv * (1.f - yv);
        *gp = *gp * d;
    }
}
int ptta_launch_gact_bwd(const GView& g, const GView& y, int act, hipStream_t s) {
    if (act == GACT_NONE) return 0;
    const long total = (long)g.B * g.H * g.W * g.C;
    long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gact_bwd_kernel, dim3((int)blocks), dim3(256), 0, s, g, y, act);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// (N,C,H,W) planar -> NHWC view, with the optional photometric normalisation (v/div - mean[c])/std[c]; zero_from_b:
// items >= zero_from_b are written as zeros (the proxy pass's zero image, nlspnmodel_adapt.py:908)
__global__ void gnchw_to_nhwc_kernel(const float* __restrict__ src, int src_nb, int src_c, GView y, int zero_from_b, int norm, float div,
                                     float m0, float m1, float m2, float s0, float s1, float s2) {
    const long total = (long)y.B * y.H * y.W * y.C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % y.C); long pix = idx / y.C;
        const long hw = (long)y.H * y.W;
        const int b = (int)(pix / hw); const long r = pix % hw;
        float v = 0.f;
        if (b < zero_from_b && c < src_c) {
            v = src[((long)(b % src_nb) * src_c + c) * hw + r];
            if (norm) { const float m = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2); v = (v / div - m) / sd; }
        }
        y.p[pix * y.ld + c] = v;
    }
}
int ptta_launch_gnchw_to_nhwc(const float* src, int src_nb, int src_c, const GView& y, int zero_from_b, int norm, float div, const float* mean,
                              const float* stdv, hipStream_t s) {
    const long total = (long)y.B * y.H * y.W * y.C;
    long blocks = (total + 255) / 256; if (blocks > 16384) blocks = 16384; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gnchw_to_nhwc_kernel, dim3((int)blocks), dim3(256), 0, s, src, src_nb, src_c, y, zero_from_b, norm, div,
                       mean ? mean[0] : 0.f, mean ? mean[1] : 0.f, mean ? mean[2] : 0.f, stdv ? stdv[0] : 1.f, stdv ? stdv[1] : 1.f,
                       stdv ? stdv[2] : 1.f);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// Weight + bias gradient of a stride-1 3x3 convolution (the adapted conv1_rgb_meta, Conv2d(48,48,3,1,1),
// nlspnmodel_adapt.py:1371): dW[co][ci][tap] = sum_p gy[p][co] * x[p+tap][ci].  Stage 1: each block reduces a slab
// of pixels into a [9][Ci][Co] partial (one thread per (ci,co) pair group); stage 2: fixed-order sum of the partials.
__global__ __launch_bounds__(256) void gwgrad_part_kernel(GView x, GView gy, int nslab, float* __restrict__ part) {
    const int Ci = x.C, Co = gy.C, H = x.H, W = x.W;
    const long P = (long)x.B * H * W;
    const long per = (P + nslab - 1) / nslab;
    const long p0 = (long)blockIdx.x * per;
    long p1 = p0 + per; if (p1 > P) p1 = P;
    const int npair = Ci * Co;
    float* out = part + (long)blockIdx.x * (9 * npair + Co);
    for (int e = threadIdx.x; e < 9 * npair + Co; e += blockDim.x) {
        float acc = 0.f;
        if (e < 9 * npair) {
            const int tap = e / npair, r = e % npair, ci = r / Co, co = r % Co;
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            for (long p = p0; p < p1; ++p) {
                const int px = (int)(p % W); const long t_ = p / W; const int py = (int)(t_ % H);
                const int yy = py + dy, xx = px + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                acc = fmaf(gy.p[p * gy.ld + co], x.p[(p + (long)dy * W + dx) * x.ld + ci], acc);
            }
        } else {
            const int co = e - 9 * npair;
            for (long p = p0; p < p1; ++p) acc += gy.p[p * gy.ld + co];
        }
        out[e] = acc;
    }
}
__global__ void gwgrad_reduce_kernel(const float* __restrict__ part, int nslab, int Ci, int Co, float* __restrict__ gw, float* __restrict__ gb) {
    const int npair = Ci * Co, n = 9 * npair + Co;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    double s = 0.0;
    for (int k = 0; k < nslab; ++k) s += (double)part[(long)k * n + e];
    if (e < 9 * npair) {
        const int tap = e / npair, r = e % npair, ci = r / Co, co = r % Co;
        gw[((long)co * Ci + ci) * 9 + tap] = (float)s;
    } else if (gb) gb[e - 9 * npair] = (float)s;
}
int ptta_gwgrad_slabs(long pixels) { long n = (pixels + 255) / 256; return (int)(n > 1024 ? 1024 : (n < 1 ? 1 : n)); }
int ptta_launch_gwgrad(const GView& x, const GView& gy, float* part, float* gw, float* gb, hipStream_t s) {
    const int nslab = ptta_gwgrad_slabs((long)x.B * x.H * x.W);
    hipLaunchKernelGGL(gwgrad_part_kernel, dim3(nslab), dim3(256), 0, s, x, gy, nslab, part);
    const int n = 9 * x.C * gy.C + gy.C;
    hipLaunchKernelGGL(gwgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, part, nslab, x.C, gy.C, gw, gb);
    PTTA_CHECK_LAUNCH();
    return 0;
}
