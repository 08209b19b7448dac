// Batch-statistics normalisation for the NLSPN backbone, any channel count, on strided NHWC views.
// adapt_parameters('meta_bn') (src/nlspn_model_adapt.py:322-337) drops the running statistics of every
// BatchNorm2d, so train AND eval forwards normalise with the statistics of the current batch; the MLP heads'
// BatchNorm1d (nlspnmodel_adapt.py:1398-1404) are in train mode on the TTA path: same arithmetic with H*W = rows.
//   forward : mean/var (biased) per (pass, channel) -> y = act(x*scale+shift) [+ res -> relu]
//   backward: g1 = gy*act'(y);  dbeta = sum g1, dgamma = sum g1*xhat;  gx += gamma*inv*(g1 - dbeta/R - xhat*dgamma/R)
// "pass" = an equal slice of the batch with its own statistics: the grad pass and the no-grad proxy pass of one
// step are batched as [real | proxy] and must not share statistics (they are separate forward calls in the
// reference, nlspnmodel_adapt.py:866-916).  All reductions are two-stage with a fixed order (deterministic).
#include "ptta_common.h"
#include "ptta_kernels.h"

#define GBN_BLOCKS 1024

// activation gradient from the SAVED OUTPUT yv: relu / leaky-relu by sign, ELU: 1 for y > 0, y + 1 (= e^x) otherwise
__device__ __forceinline__ float gbn_act_grad(float gv, float yv, int act) {
    if (act == GACT_RELU) return yv > 0.f ? gv : 0.f;
    if (act == GACT_LRELU) return yv > 0.f ? gv : 0.2f * gv;
    if (act == GACT_ELU) return yv > 0.f ? gv : gv * (yv + 1.f);
    return gv;
}
__device__ __forceinline__ float gbn_act(float v, int act) {
    if (act == GACT_RELU) return v > 0.f ? v : 0.f;
    if (act == GACT_LRELU) return v > 0.f ? v : 0.2f * v;
    if (act == GACT_ELU) return v > 0.f ? v : expm1f(v);
    return v;
}

// partial sums over the pixels of one pass: MODE 0: {sum x, sum x^2}; MODE 1: {sum g1, sum g1*xhat}
// part layout [pass][block][2][C]
// YLESS (MODE 1, plain ReLU without a residual): the ReLU mask is recomputed as fma(x, scale, shift) > 0 -- the very expression whose
// sign the forward stored as y > 0 (gbn_apply_kernel) -- so the saved output is not read: 2 of the backward's 7 map passes less
template <int MODE, bool YLESS = false>
__global__ __launch_bounds__(256) void gbn_stats_kernel(GView x, GView g, GView y, int npass, int act, int res_relu,
                                                        const float* __restrict__ mean, const float* __restrict__ inv,
                                                        float* __restrict__ part, const float* __restrict__ fscale = nullptr,
                                                        const float* __restrict__ fshift = nullptr) {
    extern __shared__ float red[];                       // [nsub][2][CT]
    const int C = x.C, CT = C < 256 ? C : 256, nsub = 256 / CT;
    const int lc = threadIdx.x % CT, sub = threadIdx.x / CT;
    const int pass = blockIdx.y;
    const long ppp = (long)(x.B / npass) * x.H * x.W;    // pixels per pass
    const long pix0 = (long)pass * ppp;
    for (int cb = 0; cb < C; cb += CT) {
        const int c = cb + lc;
        float s1 = 0.f, s2 = 0.f;
        if (sub < nsub && c < C) {
            float mu = 0.f, iv = 0.f;
            if (MODE == 1) { mu = mean[pass * C + c]; iv = inv[pass * C + c]; }
            const long pstride = (long)gridDim.x * nsub;
            long p = (long)blockIdx.x * nsub + sub;
            if (MODE == 0) {
                for (; p + 3 * pstride < ppp; p += 4 * pstride) {           // four independent loads in flight
                    const float a0 = x.p[(pix0 + p) * x.ld + c], a1 = x.p[(pix0 + p + pstride) * x.ld + c];
                    const float a2 = x.p[(pix0 + p + 2 * pstride) * x.ld + c], a3 = x.p[(pix0 + p + 3 * pstride) * x.ld + c];
                    s1 += (a0 + a1) + (a2 + a3); s2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                }
            }
            if (MODE == 1) {
                // four pixels' loads (x, g, y: twelve) in flight, accumulated in the pixel order of the plain loop below: same sums bit for
                // bit.  (One pixel per iteration made this pass a chain of dependent load latencies: 33 us per call in the NLSPN profile.)
                const float fs = fscale ? fscale[c] : 0.f, fh = fscale ? fshift[c] : 0.f;
                for (; p + 3 * pstride < ppp; p += 4 * pstride) {
                    float xv[4], gv[4], yv[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const long q = pix0 + p + k * pstride;
                        xv[k] = x.p[q * x.ld + c]; gv[k] = g.p[q * g.ld + c];
                        if constexpr (YLESS) yv[k] = fmaf(xv[k], fs, fh); else yv[k] = y.p[q * y.ld + c];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float gk = gv[k];
                        if constexpr (YLESS) gk = yv[k] > 0.f ? gk : 0.f;
                        else if (res_relu) {
                            gk = yv[k] > 0.f ? gk : 0.f;
                            if (fscale) gk = fmaf(xv[k], fs, fh) > 0.f ? gk : 0.f;
                        } else gk = gbn_act_grad(gk, yv[k], act);
                        s1 += gk; s2 += gk * (xv[k] - mu) * iv;
                    }
                }
            }
            for (; p < ppp; p += pstride) {
                const float xv = x.p[(pix0 + p) * x.ld + c];
                if (MODE == 0) { s1 += xv; s2 += xv * xv; }
                else {
                    float gv = g.p[(pix0 + p) * g.ld + c];
                    const float yv = YLESS ? fmaf(xv, fscale[c], fshift[c]) : y.p[(pix0 + p) * y.ld + c];
                    if constexpr (YLESS) gv = yv > 0.f ? gv : 0.f;
                    else if (res_relu) {
                        gv = yv > 0.f ? gv : 0.f;
                        if (fscale) gv = fmaf(xv, fscale[c], fshift[c]) > 0.f ? gv : 0.f;      // inner ReLU of relu(relu(bn x) + res)
                    } else gv = gbn_act_grad(gv, yv, act);
                    s1 += gv; s2 += gv * (xv - mu) * iv;
                }
            }
        }
        __syncthreads();
        if (sub < nsub && c < C) { red[(sub * 2 + 0) * CT + lc] = s1; red[(sub * 2 + 1) * CT + lc] = s2; }
        __syncthreads();
        if (threadIdx.x < CT && c < C) {
            float a1 = 0.f, a2 = 0.f;
            for (int k = 0; k < nsub; ++k) { a1 += red[(k * 2 + 0) * CT + lc]; a2 += red[(k * 2 + 1) * CT + lc]; }
            float* o = part + (((long)pass * gridDim.x + blockIdx.x) * 2) * C;
            o[c] = a1; o[C + c] = a2;
        }
    }
}

// wave-parallel fixed-order sum of the block partials of one (pass, channel): lane l takes blocks l, l+64, ...
__device__ __forceinline__ void gbn_sum_partials(const float* __restrict__ part, int nblocks, int C, int c, double& s1, double& s2) {
    const int lane = threadIdx.x & 63;
    double a1 = 0.0, a2 = 0.0;
    for (int b = lane; b < nblocks; b += 64) { const float* o = part + ((long)b * 2) * C; a1 += (double)o[c]; a2 += (double)o[C + c]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
    s1 = a1; s2 = a2;
}

// the same with all 256 threads of a block on ONE (pass, channel): thread t takes blocks t, t + 256, ... (four loads in flight), waves
// combine through LDS in a fixed order.  The sums are fp64 of fp32 partials: exact to 2^-53, so the order is immaterial for the fp32
// statistics -- and a full-resolution layer has 3,344 tile partials per channel, 52 dependent iterations for a single wave (22 us).
__device__ __forceinline__ void gbn_sum_partials_block(const float* __restrict__ part, int nblocks, int C, int c, double& s1, double& s2) {
    __shared__ double red[8];
    double a1 = 0.0, a2 = 0.0;
    int b = threadIdx.x;
    for (; b + 768 < nblocks; b += 1024) {
        const float* o0 = part + ((long)b * 2) * C; const float* o1 = o0 + 512L * C; const float* o2 = o1 + 512L * C; const float* o3 = o2 + 512L * C;
        const float x0 = o0[c], y0 = o0[C + c], x1 = o1[c], y1 = o1[C + c], x2 = o2[c], y2 = o2[C + c], x3 = o3[c], y3 = o3[C + c];
        a1 += ((double)x0 + (double)x1) + ((double)x2 + (double)x3); a2 += ((double)y0 + (double)y1) + ((double)y2 + (double)y3);
    }
    for (; b < nblocks; b += 256) { const float* o = part + ((long)b * 2) * C; a1 += (double)o[c]; a2 += (double)o[C + c]; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { a1 += __shfl_xor(a1, m); a2 += __shfl_xor(a2, m); }
    __syncthreads();                                     // red[] may still be read by the previous call's thread 0
    if ((threadIdx.x & 63) == 0) { red[(threadIdx.x >> 6) * 2] = a1; red[(threadIdx.x >> 6) * 2 + 1] = a2; }
    __syncthreads();
    s1 = (red[0] + red[2]) + (red[4] + red[6]); s2 = (red[1] + red[3]) + (red[5] + red[7]);
}

// ---- SyncBatchNorm exchange (ptta_kernels.h: PttaStatSync) ------------------------------------------------------------------
__global__ __launch_bounds__(256) void stat_collapse_kernel(const float* __restrict__ part, int nblocks, int C, int npass, double* __restrict__ out) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);               // (pass, which, c): one wave each
    if (idx >= npass * 2 * C) return;
    const int c = idx % C, which = (idx / C) & 1, pass = idx / (2 * C);
    const int lane = threadIdx.x & 63;
    double a = 0.0;
    for (int b = lane; b < nblocks; b += 64) a += (double)part[(((long)pass * nblocks + b) * 2 + which) * C + c];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) a += __shfl_xor(a, m);
    if (lane == 0) out[idx] = a;
}
__global__ void stat_expand_kernel(const double* __restrict__ in, int nblocks, int C, int npass, float* __restrict__ part) {
    const long total = (long)npass * nblocks * 2 * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); long t_ = i / C;
        const int which = (int)(t_ & 1); t_ >>= 1;
        const int b = (int)(t_ % nblocks); const int pass = (int)(t_ / nblocks);
        const double v = in[((long)pass * 2 + which) * C + c];
        const float hi = (float)v;
        // hi part in block 0, the double -> float correction in block 1; a single-block BatchNorm keeps the rounded sum only
        // (2^-24 relative: below the fp32 statistics it replaces)
        part[i] = b == 0 ? hi : (b == 1 ? (float)(v - (double)hi) : 0.f);
    }
}
int ptta_stat_sync(const PttaStatSync* sy, float* part, int nblocks, int C, int npass, hipStream_t s) {
    if (!sy || !sy->on()) return 0;
    const long n = (long)npass * 2 * C;
    if (n > sy->cap) return -22;
    hipLaunchKernelGGL(stat_collapse_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, part, nblocks, C, npass, sy->buf);
    if (hipGetLastError() != hipSuccess) return -5;
    const int rc = sy->exchange((long long)n, s);                            // SUM over the ranks, in place, ordered on `s`
    if (rc) return rc;
    const long total = (long)npass * nblocks * 2 * C;
    long blocks = (total + 255) / 256; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(stat_expand_kernel, dim3((int)blocks), dim3(256), 0, s, sy->buf, nblocks, C, npass, part);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

// forward finalize: per (pass, c): mean, inv, scale = gamma*inv, shift = beta - mean*scale   (st = [4][npass][C]).  One wave per channel,
// the passes in order.  TRACKED BatchNorm (rm != null; CostDCNet's BatchNorm3d / BatchNorm1d): the momentum update of the running
// statistics -- one per pass in order, `repeats` times each, exactly gbn_running_update_kernel's arithmetic -- and the eval-mode affine
// of the UPDATED statistics (st_eval = [mean, inv, scale, shift][C]) come out of the same launch (two launches per BatchNorm fewer
// per step + eval forward).
struct GbnTrack { float* rm; float* rv; long long* nbt; float momentum; int repeats; float* st_eval; };
__global__ __launch_bounds__(256) void gbn_finalize_kernel(const float* __restrict__ part, int nblocks, int npass, int C, long R, float eps,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ st,
                                                           GbnTrack tr) {
    // untracked: one block per (pass, channel); tracked: one block per channel walks the passes in order (the momentum updates chain)
    const int unit = blockIdx.x;
    const int c = tr.rm ? unit : unit % C;
    const int pfirst = tr.rm ? 0 : unit / C, plast = tr.rm ? npass : pfirst + 1;
    const int n = npass * C;
    float m_run = 0.f, v_run = 0.f;
    const bool lane0 = threadIdx.x == 0;
    if (tr.rm && lane0) { m_run = tr.rm[c]; v_run = tr.rv[c]; }
    for (int pass = pfirst; pass < plast; ++pass) {
        double s1, s2;
        gbn_sum_partials_block(part + ((long)pass * nblocks * 2) * C, nblocks, C, c, s1, s2);
        if (!lane0) continue;
        const double m = s1 / (double)R;
        double var = s2 / (double)R - m * m; if (var < 0.0) var = 0.0;
        const float iv = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * iv;
        const int idx = pass * C + c;
        st[idx] = (float)m; st[n + idx] = iv; st[2 * n + idx] = sc; st[3 * n + idx] = beta[c] - (float)m * sc;
        if (tr.rm) {
            double v2 = 1.0 / ((double)iv * (double)iv) - (double)eps;          // as gbn_running_update_kernel (from the stored inv)
            if (v2 < 0.0) v2 = 0.0;
            const float unb = (float)(R > 1 ? v2 * (double)R / (double)(R - 1) : v2);
            const float mu = (float)m;
            for (int k = 0; k < tr.repeats; ++k) { m_run = (1.f - tr.momentum) * m_run + tr.momentum * mu; v_run = (1.f - tr.momentum) * v_run + tr.momentum * unb; }
        }
    }
    if (tr.rm && lane0) {
        tr.rm[c] = m_run; tr.rv[c] = v_run;
        if (c == 0 && tr.nbt) *tr.nbt += (long long)npass * tr.repeats;
        if (tr.st_eval) {
            const float iv = 1.f / sqrtf(v_run + eps), sc = gamma[c] * iv;
            tr.st_eval[c] = m_run; tr.st_eval[C + c] = iv; tr.st_eval[2 * C + c] = sc; tr.st_eval[3 * C + c] = beta[c] - m_run * sc;
        }
    }
}

// y = act(x*scale+shift) ; with res: y = relu(x*scale+shift + res)   (BasicBlock.forward, nlspnmodel_adapt.py:98-116)
// four channels per thread (every channel count and row stride on this path is a multiple of 4)
__global__ void gbn_apply_kernel(GView x, GView res, GView y, int npass, int act, const float* __restrict__ st, int res_relu = 1,
                                 int act_first = 0) {
    const int C4 = x.C >> 2, n = npass * x.C;
    const long total = (long)x.B * x.H * x.W * C4;
    const long ppp = (long)(x.B / npass) * x.H * x.W;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) << 2; const long pix = idx / C4;
        const int pass = (int)(pix / ppp);
        const float4 xv = *(const float4*)(x.p + pix * x.ld + c);
        const float4 sc = *(const float4*)(st + 2 * n + pass * x.C + c), sh = *(const float4*)(st + 3 * n + pass * x.C + c);
        float v[4] = {fmaf(xv.x, sc.x, sh.x), fmaf(xv.y, sc.y, sh.y), fmaf(xv.z, sc.z, sh.z), fmaf(xv.w, sc.w, sh.w)};
        if (res.p) {
            const float4 r = *(const float4*)(res.p + pix * res.ld + c);
            if (act_first) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = gbn_act(v[k], act);
            }
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
            if (res_relu) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
            }
        } else if (act != GACT_NONE) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = gbn_act(v[k], act);
        }
        *(float4*)(y.p + pix * y.ld + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// fused_blocks > 0: `part` was already filled by the producing convolution's epilogue ([pass][fused_blocks][2][C]), skip the
// statistics pass
int ptta_launch_gbn_forward(const GView& x, const GView& res, const GView& y, int npass, int act, float eps, const float* gamma,
                            const float* beta, float* part, float* st, hipStream_t s, int fused_blocks, int act_first, const PttaStatSync* sync,
                            float* rm, float* rv, long long* nbt, float momentum, int repeats, float* st_eval) {
    const int C = x.C, CT = C < 256 ? C : 256, nsub = 256 / CT;
    const size_t lds = (size_t)nsub * 2 * CT * sizeof(float);
    const long R = (long)(x.B / npass) * x.H * x.W;
    int blocks = (int)((R + nsub - 1) / nsub); if (blocks > GBN_BLOCKS) blocks = GBN_BLOCKS; if (blocks < 1) blocks = 1;
    if (fused_blocks > 0) blocks = fused_blocks;
    else hipLaunchKernelGGL((gbn_stats_kernel<0>), dim3(blocks, npass), dim3(256), lds, s, x, x, x, npass, 0, 0, nullptr, nullptr, part);
    long Rg = R;
    if (sync && sync->on()) { const int rc = ptta_stat_sync(sync, part, blocks, C, npass, s); if (rc) return rc; Rg = R * sync->world; }
    const GbnTrack tr{rm, rv, nbt, momentum, repeats, st_eval};
    hipLaunchKernelGGL(gbn_finalize_kernel, dim3(rm ? C : npass * C), dim3(256), 0, s, part, blocks, npass, C, Rg, eps, gamma, beta, st, tr);
    if ((C & 3) || (x.ld & 3) || (y.ld & 3) || (res.p && (res.ld & 3))) return -22;
    const long total = (long)x.B * x.H * x.W * (C >> 2);
    long ab = (total + 255) / 256; if (ab > 16384) ab = 16384; if (ab < 1) ab = 1;
    hipLaunchKernelGGL(gbn_apply_kernel, dim3((int)ab), dim3(256), 0, s, x, res, y, npass, act, st, 1, act_first);
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_gbn_part_floats(int C, int npass) { return npass * GBN_BLOCKS * 2 * C; }

// normalise (+activation) (+residual, optionally without the BasicBlock's ReLU) from finalized statistics st = [4][npass][C]
int ptta_launch_gbn_apply(const GView& x, const GView& res, const GView& y, int npass, int act, const float* st, int res_relu, hipStream_t s,
                          int act_first) {
    const int C = x.C;
    if ((C & 3) || (x.ld & 3) || (y.ld & 3) || (res.p && (res.ld & 3))) return -22;
    const long total = (long)x.B * x.H * x.W * (C >> 2);
    long ab = (total + 255) / 256; if (ab > 16384) ab = 16384; if (ab < 1) ab = 1;
    hipLaunchKernelGGL(gbn_apply_kernel, dim3((int)ab), dim3(256), 0, s, x, res, y, npass, act, st, res_relu, act_first);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// backward finalize (pass 0 = the grad pass only): dbeta, dgamma, and bw = [gscale, c1, c2][C]
__global__ __launch_bounds__(256) void gbn_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, int C, long R, const float* __restrict__ gamma,
                                                               const float* __restrict__ inv, float* dgamma, float* dbeta, float* __restrict__ bw,
                                                               float grad_scale = 1.f) {
    const int c = blockIdx.x;
    double s1, s2;
    gbn_sum_partials_block(part, nblocks, C, c, s1, s2);
    if (threadIdx.x) return;
    // with SyncBatchNorm the sums are global: 1/world of them = the DDP-averaged local parameter gradients
    if (dbeta) dbeta[c] = (float)s1 * grad_scale;
    if (dgamma) dgamma[c] = (float)s2 * grad_scale;
    bw[c] = gamma[c] * inv[c]; bw[C + c] = (float)(s1 / (double)R); bw[2 * C + c] = (float)(s2 / (double)R);
}

// gx (+)= gscale*(g1 - c1 - xhat*c2) ; gres (+)= g1 (residual branch of a BasicBlock); four channels per thread
// YLESS: as gbn_stats_kernel -- plain ReLU, the mask from fma(x, fscale, fshift) > 0, y not read
template <bool YLESS>
__global__ void gbn_bwd_apply_kernel(GView x, GView g, GView y, GView gx, GView gres, int act, int res_relu, int acc_gx, int acc_gres,
                                     const float* __restrict__ mean, const float* __restrict__ inv, const float* __restrict__ bw,
                                     const float* __restrict__ fscale = nullptr, const float* __restrict__ fshift = nullptr) {
    const int C = x.C, C4 = C >> 2;
    const long total = (long)g.B * g.H * g.W * C4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) << 2; const long pix = idx / C4;
        const float4 g4 = *(const float4*)(g.p + pix * g.ld + c);
        const float4 x4 = *(const float4*)(x.p + pix * x.ld + c);
        float4 y4;
        if constexpr (YLESS) {
            const float4 fs = *(const float4*)(fscale + c), fh = *(const float4*)(fshift + c);
            y4 = make_float4(fmaf(x4.x, fs.x, fh.x), fmaf(x4.y, fs.y, fh.y), fmaf(x4.z, fs.z, fh.z), fmaf(x4.w, fs.w, fh.w));
        } else y4 = *(const float4*)(y.p + pix * y.ld + c);
        const float4 mu = *(const float4*)(mean + c), iv = *(const float4*)(inv + c);
        const float4 b0 = *(const float4*)(bw + c), b1 = *(const float4*)(bw + C + c), b2 = *(const float4*)(bw + 2 * C + c);
        float gv[4] = {g4.x, g4.y, g4.z, g4.w};
        const float yv[4] = {y4.x, y4.y, y4.z, y4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w};
        const float m_[4] = {mu.x, mu.y, mu.z, mu.w}, i_[4] = {iv.x, iv.y, iv.z, iv.w};
        const float s_[4] = {b0.x, b0.y, b0.z, b0.w}, c1[4] = {b1.x, b1.y, b1.z, b1.w}, c2[4] = {b2.x, b2.y, b2.z, b2.w};
        float d[4], gb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (YLESS || res_relu) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;             // gradient of the block's final ReLU (goes to the residual too)
            else gv[k] = gbn_act_grad(gv[k], yv[k], act);
            gb[k] = gv[k];
            if (!YLESS && res_relu && fscale) gb[k] = fmaf(xv[k], fscale[c + k], fshift[c + k]) > 0.f ? gv[k] : 0.f;   // inner ReLU of relu(relu(bn x) + res)
            const float xh = (xv[k] - m_[k]) * i_[k];
            d[k] = s_[k] * (gb[k] - c1[k] - xh * c2[k]);
        }
        float4* o = (float4*)(gx.p + pix * gx.ld + c);
        if (acc_gx) { const float4 t = *o; d[0] += t.x; d[1] += t.y; d[2] += t.z; d[3] += t.w; }
        *o = make_float4(d[0], d[1], d[2], d[3]);
        if (gres.p) {
            float4* r = (float4*)(gres.p + pix * gres.ld + c);
            if (acc_gres) { const float4 t = *r; gv[0] += t.x; gv[1] += t.y; gv[2] += t.z; gv[3] += t.w; }
            *r = make_float4(gv[0], gv[1], gv[2], gv[3]);
        }
    }
}

// x: raw conv output (saved), g: gradient of y over the grad-pass items (g.B = items of pass 0), y: saved output.
// st: forward statistics [4][npass][C] (pass 0 is used).  dgamma/dbeta may be null (frozen affine parameters).
int ptta_launch_gbn_backward(const GView& x, const GView& g, const GView& y, const GView& gx, const GView& gres, int npass, int act,
                             int res_relu, int acc_gx, int acc_gres, const float* gamma, const float* st, float* part, float* bw, float* dgamma,
                             float* dbeta, hipStream_t s, int act_first, const PttaStatSync* sync) {
    const int C = x.C, CT = C < 256 ? C : 256, nsub = 256 / CT;
    const size_t lds = (size_t)nsub * 2 * CT * sizeof(float);
    GView x0 = x; x0.B = g.B;
    GView y0 = y; y0.B = g.B;
    const long R = (long)g.B * g.H * g.W;
    int blocks = (int)((R + nsub - 1) / nsub); if (blocks > GBN_BLOCKS) blocks = GBN_BLOCKS; if (blocks < 1) blocks = 1;
    const float* mean = st; const float* inv = st + (long)npass * C;
    // act_first (CostDCNet's ResBlock, encoder2d.py:44-52): the inner ReLU's mask is recomputed from x and the forward affine
    const float* fscale = (act_first && res_relu) ? st + 2L * npass * C : nullptr;
    const float* fshift = (act_first && res_relu) ? st + 3L * npass * C : nullptr;
    // plain ReLU (no residual): the mask comes from x and the forward affine of pass 0, the saved output is not read
#ifdef GBN_NO_YLESS                   // (comparison builds only: tools/exp/yless_bitwise.sh)
    const bool yless = false;
#else
    const bool yless = !res_relu && act == GACT_RELU && !act_first;
#endif
    const float* ysc = st + 2L * npass * C; const float* ysh = st + 3L * npass * C;
    if (yless) hipLaunchKernelGGL((gbn_stats_kernel<1, true>), dim3(blocks, 1), dim3(256), lds, s, x0, g, y0, 1, act, res_relu, mean, inv, part, ysc, ysh);
    else hipLaunchKernelGGL((gbn_stats_kernel<1>), dim3(blocks, 1), dim3(256), lds, s, x0, g, y0, 1, act, res_relu, mean, inv, part, fscale, fshift);
    long Rg = R; float gsc = 1.f;
    if (sync && sync->on()) { const int rc = ptta_stat_sync(sync, part, blocks, C, 1, s); if (rc) return rc; Rg = R * sync->world; gsc = 1.f / (float)sync->world; }
    hipLaunchKernelGGL(gbn_bwd_finalize_kernel, dim3(C), dim3(256), 0, s, part, blocks, C, Rg, gamma, inv, dgamma, dbeta, bw, gsc);
    if ((C & 3) || (x.ld & 3) || (g.ld & 3) || (y.ld & 3) || (gx.ld & 3) || (gres.p && (gres.ld & 3))) return -22;
    const long total = R * (C >> 2);
    long ab = (total + 255) / 256; if (ab > 16384) ab = 16384; if (ab < 1) ab = 1;
    if (yless) hipLaunchKernelGGL(gbn_bwd_apply_kernel<true>, dim3((int)ab), dim3(256), 0, s, x0, g, y0, gx, gres, act, res_relu, acc_gx, acc_gres, mean, inv, bw, ysc, ysh);
    else hipLaunchKernelGGL(gbn_bwd_apply_kernel<false>, dim3((int)ab), dim3(256), 0, s, x0, g, y0, gx, gres, act, res_relu, acc_gx, acc_gres, mean, inv, bw,
                            fscale, fshift);
    PTTA_CHECK_LAUNCH();
    return 0;
}


// ---- tracked BatchNorm (BatchNorm3d / BatchNorm1d / MinkowskiBatchNorm of CostDCNet keep their running statistics) --------
// train: after the forward finalize, running = (1 - m) * running + m * batch (unbiased variance), one update per pass in
// order (real frames first, then the proxy pass, as the reference's two forward calls), `repeats` times each when the
// reference runs the layer several times on the same input; num_batches_tracked += npass * repeats.
__global__ void gbn_running_update_kernel(const float* __restrict__ st, int npass, int C, long R, float momentum, float eps,
                                          float* rm, float* rv, long long* nbt, int repeats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && nbt) *nbt += (long long)npass * repeats;
    if (c >= C) return;
    float m = rm[c], v = rv[c];
    for (int pass = 0; pass < npass; ++pass) {
        const float mu = st[pass * C + c], iv = st[(long)npass * C + pass * C + c];
        double var = 1.0 / ((double)iv * (double)iv) - (double)eps;
        if (var < 0.0) var = 0.0;
        const float unb = (float)(R > 1 ? var * (double)R / (double)(R - 1) : var);
        for (int k = 0; k < repeats; ++k) { m = (1.f - momentum) * m + momentum * mu; v = (1.f - momentum) * v + momentum * unb; }
    }
    rm[c] = m; rv[c] = v;
}
int ptta_launch_gbn_running_update(const float* st, int npass, int C, long R, float momentum, float eps, float* rm, float* rv, long long* nbt,
                                   int repeats, hipStream_t s) {
    hipLaunchKernelGGL(gbn_running_update_kernel, dim3((C + 255) / 256), dim3(256), 0, s, st, npass, C, R, momentum, eps, rm, rv, nbt, repeats);
    PTTA_CHECK_LAUNCH();
    return 0;
}
// eval: st (npass = 1 layout [mean, inv, scale, shift][C]) from the running statistics
__global__ void gbn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                       const float* __restrict__ rv, float eps, int C, float* __restrict__ st) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float iv = 1.f / sqrtf(rv[c] + eps), sc = gamma[c] * iv;
    st[c] = rm[c]; st[C + c] = iv; st[2 * C + c] = sc; st[3 * C + c] = beta[c] - rm[c] * sc;
}
int ptta_launch_gbn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* st, hipStream_t s) {
    hipLaunchKernelGGL(gbn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, s, gamma, beta, rm, rv, eps, C, st);
    PTTA_CHECK_LAUNCH();
    return 0;
}
