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

// The map the reference propagates is (1-mask_fix)*feat + mask_fix*feat_fix (:362-364), mask_fix = fix > 0.  The sweeps keep
// it MATERIALISED ("pinned"): sweep k reads the pinned map of sweep k-1 with ONE load per bilinear corner (feat-or-fix was two
// dependent loads) and writes pin(its raw output); only the last sweep writes the raw output (the network's result).
__global__ __launch_bounds__(256) void nl_pin_kernel(const float* __restrict__ feat, const float* __restrict__ fix, float* __restrict__ out, long n) {
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x) {
        const float f = fix[k];
        out[k] = f > 0.f ? f : feat[k];
    }
}

// Geometry of one bilinear tap: the four corner indices (clamped to a valid address when the corner does not contribute) and
// whether each contributes (deformable-convolution boundary rule, ptta_common.h corner_of).  Loads are then UNCONDITIONAL:
// behind per-corner branches every one of a pixel's 36 (forward) / 99 (backward) loads waited for the previous one.
struct TapG { int q[4]; bool ok[4]; float lh, lw; };
__device__ __forceinline__ TapG tap_geom(int H, int W, float h, float w) {
    TapG t;
    const Corner c = corner_of(h, w, H, W);
    const int h1 = c.h0 + 1, w1 = c.w0 + 1;
    t.ok[0] = c.inside && c.h0 >= 0 && c.w0 >= 0; t.ok[1] = c.inside && c.h0 >= 0 && w1 <= W - 1;
    t.ok[2] = c.inside && h1 <= H - 1 && c.w0 >= 0; t.ok[3] = c.inside && h1 <= H - 1 && w1 <= W - 1;
    t.q[0] = t.ok[0] ? c.h0 * W + c.w0 : 0; t.q[1] = t.ok[1] ? c.h0 * W + w1 : 0;
    t.q[2] = t.ok[2] ? h1 * W + c.w0 : 0; t.q[3] = t.ok[3] ? h1 * W + w1 : 0;
    t.lh = c.lh; t.lw = c.lw;
    return t;
}
__device__ __forceinline__ float tap_blend(const TapG& t, const float* v) {
    const float hh = 1.f - t.lh, hw = 1.f - t.lw;
    return hh * hw * v[0] + hh * t.lw * v[1] + t.lh * hw * v[2] + t.lh * t.lw * v[3];
}

__global__ __launch_bounds__(256) void nl_prop_fwd_kernel(const float* __restrict__ pinned, const float* __restrict__ fix,
                                                          const float* __restrict__ off9, const float* __restrict__ aff9,
                                                          float* __restrict__ out, int pin_out, int B, int H, int W) {
    const long P = (long)H * W, total = (long)B * P;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / P); const long pix = idx % P;
        const int y = (int)(pix / W), x = (int)(pix % W);
        const float* fb = pinned + (long)b * P;
        const float* o = off9 + (long)b * 18 * P + pix; const float* a = aff9 + (long)b * 9 * P + pix;
        float ov[18], av[9];
#pragma unroll
        for (int k = 0; k < 18; ++k) ov[k] = o[(long)k * P];
#pragma unroll
        for (int k = 0; k < 9; ++k) av[k] = a[(long)k * P];
        const float fx = pin_out ? fix[idx] : 0.f;
        TapG t[9];
        float v[9][4];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            t[k] = tap_geom(H, W, (float)(y + k / 3 - 1) + ov[2 * k], (float)(x + k % 3 - 1) + ov[2 * k + 1]);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[k][j] = fb[t[k].q[j]];
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (!t[k].ok[j]) v[k][j] = 0.f;
            acc = fmaf(av[k], tap_blend(t[k], v[k]), acc);
        }
        out[idx] = (pin_out && fx > 0.f) ? fx : acc;
    }
}

// one thread per pixel, one block per 16x16 tile: g_aff9 / g_off9 accumulate over the sweeps (same thread every sweep ->
// deterministic); the feature gradient is scattered onto the corners that are not pinned by the sparse input.  Offsets
// are a few pixels at most, so the scatter goes to an LDS copy of the tile + a 4-pixel apron (ds_add_f32) and is
// flushed with one global atomic per touched cell; targets outside the apron fall back to global atomics directly.
// Loads are batched three taps at a time (offsets / affinities / accumulators first, then 12 corner values + 12 pin masks).
// Round 2 added neighbour merging (below): 91 -> 74 us per sweep on the synthetic frames, whose offset fields are not smooth.
// Ablations at 352x1216 before it (one box): 91.6 us as is; without the accumulator read-modify-writes 92.0; without them AND without the
// corner gathers 92.1; with the scatter on memory-side global atomics instead of LDS 215.9 (half and half: 130.5) -- the 36
// ds_add_f32 per pixel are the whole kernel (SQ_WAIT_INST_LDS = 40 % of its wave cycles), and LDS is still the faster home.
#define PB_T 16
#define PB_R 4
#define PB_W (PB_T + 2 * PB_R)
__global__ __launch_bounds__(256, 4) void nl_prop_bwd_kernel(const float* __restrict__ pinned, const float* __restrict__ fix,
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
    const float* fb = pinned + (long)b * P; const float* xb = fix + (long)b * P;
    float* gb = gfeat + (long)b * P;
    // every lane runs the whole body (the neighbour merging below shuffles across lanes); lanes outside the image work on a
    // clamped pixel and neither store nor scatter
    const bool inimg = y < H && x < W;
    const int col = threadIdx.x & 15, wrow = (threadIdx.x >> 4) & 3;            // position inside the wave's 4 x 16 pixel patch
    {
        const long pix = (long)min(y, H - 1) * W + min(x, W - 1), idx = (long)b * P + pix;
        const float* o = off9 + (long)b * 18 * P + pix; const float* a = aff9 + (long)b * 9 * P + pix;
        float* go = goff9 + (long)b * 18 * P + pix; float* ga = gaff9 + (long)b * 9 * P + pix;
        const float g = gout[idx];
        float ov[18], av[9];
#pragma unroll
        for (int k = 0; k < 18; ++k) ov[k] = o[(long)k * P];
#pragma unroll
        for (int k = 0; k < 9; ++k) av[k] = a[(long)k * P];
#pragma unroll
        for (int k0 = 0; k0 < 9; k0 += 3) {
            TapG t[3];
            float v[3][4], m[3][4], gov[6], gav[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int k = k0 + kk;
                gov[2 * kk] = go[(long)(2 * k) * P]; gov[2 * kk + 1] = go[(long)(2 * k + 1) * P]; gav[kk] = ga[(long)k * P];
                t[kk] = tap_geom(H, W, (float)(y + k / 3 - 1) + ov[2 * k], (float)(x + k % 3 - 1) + ov[2 * k + 1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[kk][j] = fb[t[kk].q[j]]; m[kk][j] = xb[t[kk].q[j]]; }
            }
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                const int k = k0 + kk;
                const TapG& tk = t[kk];
#pragma unroll
                for (int j = 0; j < 4; ++j) if (!tk.ok[j]) v[kk][j] = 0.f;
                const float hh = 1.f - tk.lh, hw = 1.f - tk.lw, ga_ = g * av[k];
                if (inimg) {
                    ga[(long)k * P] = gav[kk] + g * tap_blend(tk, v[kk]);
                    // mdmcn_get_coordinate_weight (modulated_deform_im2col_cuda.cuh:84-125)
                    go[(long)(2 * k) * P] = gov[2 * kk] + ga_ * (-hw * v[kk][0] - tk.lw * v[kk][1] + hw * v[kk][2] + tk.lw * v[kk][3]);
                    go[(long)(2 * k + 1) * P] = gov[2 * kk + 1] + ga_ * (-hh * v[kk][0] + hh * v[kk][1] - tk.lh * v[kk][2] + tk.lh * v[kk][3]);
                }
                // contributions to the four corners; a corner outside the image or pinned by the sparse input receives nothing
                float val[4] = {hh * hw * ga_, hh * tk.lw * ga_, tk.lh * hw * ga_, tk.lh * tk.lw * ga_};
                bool act[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { act[j] = inimg && tk.ok[j] && !(m[kk][j] > 0.f) && val[j] != 0.f; if (!act[j]) val[j] = 0.f; }   // (+0 is not worth an atomic)
                // Neighbour merging.  The LDS float atomics are this kernel's whole cost; where the offset field is smooth, the
                // right-hand corners of a pixel ARE the left-hand corners of its right neighbour (and the lower corners the upper
                // corners of the pixel below): hand the contribution to that lane (DPP row shift / one permute) instead of
                // issuing an atomic of its own.  Exact: a hand-over happens only when the two cell indices are equal.
                auto merge = [&](int src, int dst, bool horizontal) __attribute__((always_inline)) {
                    const int sq = act[src] ? tk.q[src] : -1;
                    int nq, rsq; float rv;
                    if (horizontal) {
                        nq = __builtin_amdgcn_update_dpp(0, tk.q[dst], 0x101, 0xf, 0xf, true);                        // row_shl:1 -> lane + 1
                        rsq = __builtin_amdgcn_update_dpp(-1, sq, 0x111, 0xf, 0xf, false);                             // row_shr:1 -> lane - 1
                        rv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(val[src]), 0x111, 0xf, 0xf, true));
                    } else {
                        nq = __shfl_down(tk.q[dst], 16); rsq = __shfl_up(sq, 16); rv = __shfl_up(val[src], 16);
                    }
                    const bool edge_s = horizontal ? col == 15 : wrow == 3, edge_r = horizontal ? col == 0 : wrow == 0;
                    const bool give = !edge_s && sq >= 0 && nq == sq;
                    const bool take = !edge_r && rsq >= 0 && rsq == tk.q[dst];
                    if (take) { val[dst] += rv; act[dst] = true; }
                    if (give) { act[src] = false; val[src] = 0.f; }
                };
                merge(1, 0, true); merge(3, 2, true);
                merge(2, 0, false); merge(3, 1, false);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!act[j]) continue;
                    const int qy = tk.q[j] / W, qx = tk.q[j] - qy * W;
                    const int ly = qy - ty0 + PB_R, lx = qx - tx0 + PB_R;
                    if (ly >= 0 && ly < PB_W && lx >= 0 && lx < PB_W) atomicAdd(&tile[ly * PB_W + lx], val[j]);
                    else atomicAdd(gb + tk.q[j], val[j]);
                }
            }
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
int ptta_launch_nl_pin(const float* feat, const float* fix, float* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(nl_pin_kernel, dim3(nblocks(n)), dim3(256), 0, s, feat, fix, out, n);
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_launch_nl_prop_fwd(const float* pinned, const float* fix, const float* off9, const float* aff9, float* out, int pin_out, int B, int H,
                            int W, hipStream_t s) {
    hipLaunchKernelGGL(nl_prop_fwd_kernel, dim3(nblocks((long)B * H * W)), dim3(256), 0, s, pinned, fix, off9, aff9, out, pin_out, B, H, W);
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_launch_nl_prop_bwd(const float* pinned, const float* fix, const float* off9, const float* aff9, const float* gout, float* gfeat,
                            float* goff9, float* gaff9, int B, int H, int W, hipStream_t s) {
    const int tiles = B * ((W + PB_T - 1) / PB_T) * ((H + PB_T - 1) / PB_T);
    hipLaunchKernelGGL(nl_prop_bwd_kernel, dim3(tiles), dim3(256), 0, s, pinned, fix, off9, aff9, gout, gfeat, goff9, gaff9, B, H, W);
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
