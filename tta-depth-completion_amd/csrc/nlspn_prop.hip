// NLSPN non-local spatial propagation (external_src/NLSPN/src/model/nlspnmodel_adapt.py:189-373) specialised to
// what the TTA path runs: affinity = 'TGASS', conf_prop, preserve_input, prop_kernel 3, ch_f = 1
// (src/nlspn_model_adapt.py:56-68).  The reference builds it from 8 + prop_time calls of its generic DCN extension
// (one im2col buffer + addmm each) plus ~25 ATen elementwise kernels; here
//   nl_affinity_fwd : offsets/affinities of the 9 taps from the 24-channel conv output -- tanh / gamma scaling,
//                     the 8 confidence gathers (1x1 modulated deformable conv at the detached offsets, :287-311),
//                     abs-sum normalisation and the centre weight (:313-328) in ONE pass,
//   nl_prop_fwd     : one propagation sweep: re-impose the sparse input (:362-364) while sampling, 9 bilinear taps,
//   nl_prop_bwd     : its gradient (tap-affinity and offset gradients as gathers, feature gradient as a scatter
//                     of float atomics like the reference's col2im, modulated_deform_im2col_cuda.cuh:197-254),
//   nl_affinity_bwd : gradient of the fused affinity pass wrt the conv output and the confidence map.
// Layouts: feature / confidence / sparse maps planar [B][H*W]; off9 [B][18][H*W] (h,w per tap), aff9 [B][9][H*W]
// (the reference's NCHW offset / mask tensors: consecutive lanes read consecutive addresses);
// the conv output is a strided NHWC view with 24 channels (o1 | o2 | aff as torch.chunk sees them, :259-260).
#include "ptta_common.h"
#include "ptta_kernels.h"

namespace {

// pair n (0..7) of the 16 offset channels is (ch 2n, ch 2n+1) of cat(o1,o2) (= view(B,num,2,H,W), :262); tap k of
// the 3x3 window maps to pair k (k<4) / k-1 (k>4); the centre tap has zero offset
__device__ __forceinline__ int tap_of_pair(int n) { return n < 4 ? n : n + 1; }

struct AffPix {
    float t[8], c[8], a[8];      // tanh/S, confidence sample, product
    float s, sp;                 // sum|a| + 1e-4 and max(s, 1)
};

__device__ __forceinline__ void aff_forward_pixel(const float* oa, const float* __restrict__ conf, int H, int W, int y, int x,
                                                  float S, int legacy, AffPix& r) {
    float s = 0.f;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const int k = tap_of_pair(n);
        float oh = oa[2 * n], ow = oa[2 * n + 1];
        if (legacy) { oh += (float)(k / 3) - 1.f; ow += (float)(k % 3) - 1.f; }
        const Corner cn = corner_of((float)y + oh, (float)x + ow, H, W);
        r.c[n] = bilinear_at(conf, H, W, cn);
        r.t[n] = tanhf(oa[16 + n]) / (S + 1e-8f);
        r.a[n] = r.t[n] * r.c[n];
        s += fabsf(r.a[n]);
    }
    r.s = s + 1e-4f;
    r.sp = r.s < 1.f ? 1.f : r.s;
}

__global__ __launch_bounds__(256) void nl_affinity_fwd_kernel(GView oa, const float* __restrict__ conf, const float* __restrict__ Sp,
                                                              int legacy, float* __restrict__ off9, float* __restrict__ aff9) {
    const int H = oa.H, W = oa.W;
    const long P = (long)H * W, total = (long)oa.B * P;
    const float S = Sp[0];
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / P); const long pix = idx % P;
        const int y = (int)(pix / W), x = (int)(pix % W);
        float v[24];
        const float* src = oa.p + idx * oa.ld;
#pragma unroll
        for (int k = 0; k < 24; ++k) v[k] = src[k];
        AffPix r;
        aff_forward_pixel(v, conf + (long)b * P, H, W, y, x, S, legacy, r);
        float* o = off9 + (long)b * 18 * P + pix; float* a = aff9 + (long)b * 9 * P + pix;
        float sum = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int k = tap_of_pair(n);
            o[(long)(2 * k) * P] = v[2 * n]; o[(long)(2 * k + 1) * P] = v[2 * n + 1];
            const float an = r.a[n] / r.sp;
            a[(long)k * P] = an; sum += an;
        }
        o[8 * P] = 0.f; o[9 * P] = 0.f;
        a[4 * P] = 1.f - sum;
    }
}

// value of the map the reference propagates: (1-mask_fix)*feat + mask_fix*feat_fix (:362-364), mask_fix = fix > 0
__device__ __forceinline__ float pres(const float* __restrict__ feat, const float* __restrict__ fix, int q) {
    const float f = fix[q];
    return f > 0.f ? f : feat[q];
}

struct Tap { Corner c; float v1, v2, v3, v4; bool o1, o2, o3, o4; };
__device__ __forceinline__ Tap tap_sample(const float* __restrict__ feat, const float* __restrict__ fix, int H, int W, float h, float w) {
    Tap t;
    t.c = corner_of(h, w, H, W);
    const int h1 = t.c.h0 + 1, w1 = t.c.w0 + 1;
    t.o1 = t.c.inside && t.c.h0 >= 0 && t.c.w0 >= 0; t.o2 = t.c.inside && t.c.h0 >= 0 && w1 <= W - 1;
    t.o3 = t.c.inside && h1 <= H - 1 && t.c.w0 >= 0; t.o4 = t.c.inside && h1 <= H - 1 && w1 <= W - 1;
    t.v1 = t.o1 ? pres(feat, fix, t.c.h0 * W + t.c.w0) : 0.f; t.v2 = t.o2 ? pres(feat, fix, t.c.h0 * W + w1) : 0.f;
    t.v3 = t.o3 ? pres(feat, fix, h1 * W + t.c.w0) : 0.f; t.v4 = t.o4 ? pres(feat, fix, h1 * W + w1) : 0.f;
    return t;
}
__device__ __forceinline__ float tap_value(const Tap& t) {
    const float hh = 1.f - t.c.lh, hw = 1.f - t.c.lw;
    return hh * hw * t.v1 + hh * t.c.lw * t.v2 + t.c.lh * hw * t.v3 + t.c.lh * t.c.lw * t.v4;
}

__global__ __launch_bounds__(256) void nl_prop_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ fix,
                                                          const float* __restrict__ off9, const float* __restrict__ aff9,
                                                          float* __restrict__ out, int B, int H, int W) {
    const long P = (long)H * W, total = (long)B * P;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / P); const long pix = idx % P;
        const int y = (int)(pix / W), x = (int)(pix % W);
        const float* fb = feat + (long)b * P; const float* xb = fix + (long)b * P;
        const float* o = off9 + (long)b * 18 * P + pix; const float* a = aff9 + (long)b * 9 * P + pix;
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const Tap t = tap_sample(fb, xb, H, W, (float)(y + k / 3 - 1) + o[(long)(2 * k) * P], (float)(x + k % 3 - 1) + o[(long)(2 * k + 1) * P]);
            acc = fmaf(a[(long)k * P], tap_value(t), acc);
        }
        out[idx] = acc;
    }
}

// one thread per pixel, one block per 16x16 tile: g_aff9 / g_off9 accumulate over the sweeps (same thread every sweep ->
// deterministic); the feature gradient is scattered onto the corners that are not pinned by the sparse input.  Offsets
// are a few pixels at most, so the scatter goes to an LDS copy of the tile + a 4-pixel apron (ds_add_f32) and is
// flushed with one global atomic per touched cell; targets outside the apron fall back to global atomics directly.
#define PB_T 16
#define PB_R 4
#define PB_W (PB_T + 2 * PB_R)
__global__ __launch_bounds__(256) void nl_prop_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ fix,
                                                          const float* __restrict__ off9, const float* __restrict__ aff9,
                                                          const float* __restrict__ gout, float* __restrict__ gfeat,
                                                          float* __restrict__ goff9, float* __restrict__ gaff9, int B, int H, int W) {
    __shared__ float tile[PB_W * PB_W];
    const long P = (long)H * W;
    const int ntx = (W + PB_T - 1) / PB_T, nty = (H + PB_T - 1) / PB_T;
    const int b = blockIdx.x / (ntx * nty), tr = blockIdx.x % (ntx * nty);
    const int ty0 = (tr / ntx) * PB_T, tx0 = (tr % ntx) * PB_T;
    for (int k = threadIdx.x; k < PB_W * PB_W; k += 256) tile[k] = 0.f;
    __syncthreads();
    const int y = ty0 + (threadIdx.x >> 4), x = tx0 + (threadIdx.x & 15);
    const float* fb = feat + (long)b * P; const float* xb = fix + (long)b * P;
    float* gb = gfeat + (long)b * P;
    auto scatter = [&](int qy, int qx, float v) {
        if (xb[qy * W + qx] > 0.f) return;                         // pinned by the sparse input: no gradient to the feature
        const int ly = qy - ty0 + PB_R, lx = qx - tx0 + PB_R;
        if (ly >= 0 && ly < PB_W && lx >= 0 && lx < PB_W) atomicAdd(&tile[ly * PB_W + lx], v);
        else atomicAdd(gb + qy * W + qx, v);
    };
    if (y < H && x < W) {
        const long pix = (long)y * W + x, idx = (long)b * P + pix;
        const float* o = off9 + (long)b * 18 * P + pix; const float* a = aff9 + (long)b * 9 * P + pix;
        float* go = goff9 + (long)b * 18 * P + pix; float* ga = gaff9 + (long)b * 9 * P + pix;
        const float g = gout[idx];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const Tap t = tap_sample(fb, xb, H, W, (float)(y + k / 3 - 1) + o[(long)(2 * k) * P], (float)(x + k % 3 - 1) + o[(long)(2 * k + 1) * P]);
            const float hh = 1.f - t.c.lh, hw = 1.f - t.c.lw, ga_ = g * a[(long)k * P];
            ga[(long)k * P] += g * tap_value(t);
            // mdmcn_get_coordinate_weight (modulated_deform_im2col_cuda.cuh:84-125)
            go[(long)(2 * k) * P] += ga_ * (-hw * t.v1 - t.c.lw * t.v2 + hw * t.v3 + t.c.lw * t.v4);
            go[(long)(2 * k + 1) * P] += ga_ * (-hh * t.v1 + hh * t.v2 - t.c.lh * t.v3 + t.c.lh * t.v4);
            const int h1 = t.c.h0 + 1, w1 = t.c.w0 + 1;
            if (t.o1) scatter(t.c.h0, t.c.w0, hh * hw * ga_);
            if (t.o2) scatter(t.c.h0, w1, hh * t.c.lw * ga_);
            if (t.o3) scatter(h1, t.c.w0, t.c.lh * hw * ga_);
            if (t.o4) scatter(h1, w1, t.c.lh * t.c.lw * ga_);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < PB_W * PB_W; k += 256) {
        const float v = tile[k];
        if (v == 0.f) continue;
        const int qy = ty0 - PB_R + k / PB_W, qx = tx0 - PB_R + k % PB_W;
        if (qy >= 0 && qy < H && qx >= 0 && qx < W) atomicAdd(gb + qy * W + qx, v);
    }
}

// gradient of nl_affinity_fwd: g_oa (24 channels, overwritten) and g_conf (atomics; zero it first)
__global__ __launch_bounds__(256) void nl_affinity_bwd_kernel(GView oa, const float* __restrict__ conf, const float* __restrict__ Sp,
                                                              int legacy, const float* __restrict__ goff9, const float* __restrict__ gaff9,
                                                              GView goa, float* __restrict__ gconf) {
    const int H = oa.H, W = oa.W;
    const long P = (long)H * W, total = (long)oa.B * P;
    const float S = Sp[0];
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / P); const long pix = idx % P;
        const int y = (int)(pix / W), x = (int)(pix % W);
        float v[24];
        const float* src = oa.p + idx * oa.ld;
#pragma unroll
        for (int k = 0; k < 24; ++k) v[k] = src[k];
        AffPix r;
        aff_forward_pixel(v, conf + (long)b * P, H, W, y, x, S, legacy, r);
        const float* go = goff9 + (long)b * 18 * P + pix; const float* ga = gaff9 + (long)b * 9 * P + pix;
        float* dst = goa.p + idx * goa.ld;
        // centre = 1 - sum(a_hat): G_n = g_hat_n - g_centre ; a_hat = a / sp ; sp = max(sum|a| + 1e-4, 1)
        float G[8], dot = 0.f;
#pragma unroll
        for (int n = 0; n < 8; ++n) { G[n] = ga[(long)tap_of_pair(n) * P] - ga[4 * P]; dot += G[n] * (r.a[n] / r.sp); }
        const bool through = !(r.s < 1.f);
        float* gcb = gconf + (long)b * P;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int k = tap_of_pair(n);
            float g_a = G[n] / r.sp;
            if (through) g_a -= dot / r.sp * (r.a[n] > 0.f ? 1.f : (r.a[n] < 0.f ? -1.f : 0.f));
            // a = t * c
            const float g_t = g_a * r.c[n], g_c = g_a * r.t[n];
            const float th = r.t[n] * (S + 1e-8f);                    // tanh(raw)
            dst[16 + n] = g_t * (1.f - th * th) / (S + 1e-8f);
            dst[2 * n] = go[(long)(2 * k) * P]; dst[2 * n + 1] = go[(long)(2 * k + 1) * P];   // offsets are used (not detached) only by the sweeps
            // confidence gather at the detached offset: scatter g_c onto the bilinear corners
            float oh = v[2 * n], ow = v[2 * n + 1];
            if (legacy) { oh += (float)(k / 3) - 1.f; ow += (float)(k % 3) - 1.f; }
            const Corner c = corner_of((float)y + oh, (float)x + ow, H, W);
            if (c.inside) {
                const int h1 = c.h0 + 1, w1 = c.w0 + 1;
                const float hh = 1.f - c.lh, hw = 1.f - c.lw;
                if (c.h0 >= 0 && c.w0 >= 0) atomicAdd(gcb + c.h0 * W + c.w0, hh * hw * g_c);
                if (c.h0 >= 0 && w1 <= W - 1) atomicAdd(gcb + c.h0 * W + w1, hh * c.lw * g_c);
                if (h1 <= H - 1 && c.w0 >= 0) atomicAdd(gcb + h1 * W + c.w0, c.lh * hw * g_c);
                if (h1 <= H - 1 && w1 <= W - 1) atomicAdd(gcb + h1 * W + w1, c.lh * c.lw * g_c);
            }
        }
    }
}

inline int nblocks(long total) { long b = (total + 255) / 256; if (b > 16384) b = 16384; if (b < 1) b = 1; return (int)b; }

}  // namespace

int ptta_launch_nl_affinity_fwd(const GView& oa, const float* conf, const float* S, int legacy, float* off9, float* aff9, hipStream_t s) {
    if (oa.C != 24) return -22;
    hipLaunchKernelGGL(nl_affinity_fwd_kernel, dim3(nblocks((long)oa.B * oa.H * oa.W)), dim3(256), 0, s, oa, conf, S, legacy, off9, aff9);
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_launch_nl_prop_fwd(const float* feat, const float* fix, const float* off9, const float* aff9, float* out, int B, int H, int W,
                            hipStream_t s) {
    hipLaunchKernelGGL(nl_prop_fwd_kernel, dim3(nblocks((long)B * H * W)), dim3(256), 0, s, feat, fix, off9, aff9, out, B, H, W);
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_launch_nl_prop_bwd(const float* feat, const float* fix, const float* off9, const float* aff9, const float* gout, float* gfeat,
                            float* goff9, float* gaff9, int B, int H, int W, hipStream_t s) {
    const int tiles = B * ((W + PB_T - 1) / PB_T) * ((H + PB_T - 1) / PB_T);
    hipLaunchKernelGGL(nl_prop_bwd_kernel, dim3(tiles), dim3(256), 0, s, feat, fix, off9, aff9, gout, gfeat, goff9, gaff9, B, H, W);
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_launch_nl_affinity_bwd(const GView& oa, const float* conf, const float* S, int legacy, const float* goff9, const float* gaff9,
                                const GView& goa, float* gconf, hipStream_t s) {
    hipLaunchKernelGGL(nl_affinity_bwd_kernel, dim3(nblocks((long)oa.B * oa.H * oa.W)), dim3(256), 0, s, oa, conf, S, legacy, goff9,
                       gaff9, goa, gconf);
    PTTA_CHECK_LAUNCH();
    return 0;
}
