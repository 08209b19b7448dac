// Stage-2 head trainer kernels (src/head_main.py:464-480): the pieces the TTA step does not have because
// there the heads are frozen -- the `prepare` loss and its gradient, WEIGHT gradients of the four Linear layers, BatchNorm1d
// affine gradients (from the backward partials the TTA path already produces), the EMA of the target head.
//
// Weight gradient dW[o][i] = sum_r G[r][o] * X[r][i] is a reduction-GEMM over the R = N*H/4*W/4 embedding rows (26,752 per
// KITTI frame), on the matrix cores with bf16x3 arithmetic like the forward GEMMs (heads.hip).  MFMA operand layout makes
// this one LDS-light: a lane of v_mfma_f32_32x32x16_bf16 holds 8 consecutive k for ONE m (or n); with k = row and m = output
// column the lane's 8 values sit in 8 consecutive ROWS of one COLUMN of G, so each wave loads its own A fragments straight
// from global memory (a half-wave reads 32 consecutive floats of one row: 128-B segments) and only the X fragments, shared by
// the block's four waves, go through LDS, already split and in fragment order (one conflict-free ds_read_b128 each).
// The BatchNorm-backward transform of G (never materialised, same fusion as gemm pro=2) and BatchNorm+ReLU of X (pro=1) are
// applied per lane with per-COLUMN constants, which are loop-invariant registers in this layout.  Row range split over
// blockIdx.y; partials are reduced in a fixed order by a second kernel (deterministic, no atomics), which also finishes
// the bias gradient (column sums of the transformed G, accumulated by the blocks of the first X tile).
#include "ptta_common.h"
#include "ptta_kernels.h"

typedef __bf16 tbf16x2 __attribute__((ext_vector_type(2)));
typedef float tfloat2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void tsplit2(float a, float b, unsigned& hi, unsigned& lo) {
    tfloat2 v = {a, b};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, tbf16x2));
    tfloat2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, tbf16x2));
}
__device__ __forceinline__ void tsplit8(const float* v, uint4& hi, uint4& lo) {
    tsplit2(v[0], v[1], hi.x, lo.x); tsplit2(v[2], v[3], hi.y, lo.y);
    tsplit2(v[4], v[5], hi.z, lo.z); tsplit2(v[6], v[7], hi.w, lo.w);
}

// NT = number of 32-column X tiles per block (4: I = 512 in 128-column tiles; 1: I = 32)
template <int NT, int GPRO, int XPRO>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(LinWgradArgs a) {
    __shared__ __attribute__((aligned(16))) uint4 Bs[2][NT][2][64];          // [stage][tile][hi|lo][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, hg = lane >> 5;
    const int itiles = a.I / (32 * NT);
    const int o0 = (blockIdx.x / itiles) * 128 + 32 * wave;
    const int i0 = (blockIdx.x % itiles) * 32 * NT;
    const long r_begin = (long)blockIdx.y * a.rchunk;
    const long r_end = min((long)a.R, r_begin + a.rchunk);
    const int o = o0 + c;
    float gs = 0.f, c1 = 0.f, c2 = 0.f, mu = 0.f, iv = 0.f;
    if (GPRO) { gs = a.gscale[o]; c1 = a.gc1[o]; c2 = a.gc2[o]; mu = a.gmean[o]; iv = a.ginv[o]; }
    const int xi = i0 + 32 * (wave % NT) + c;                                  // the X column this lane stages
    float xs = 1.f, xsh = 0.f;
    if (XPRO) { xs = a.xscale[xi]; xsh = a.xshift[xi]; }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    // software pipeline: the raw values of slice k+1 are fetched into registers before the MFMAs of slice k are issued
    float gr[8], ghr[8], xr[8];
    auto fetch = [&](long r0) __attribute__((always_inline)) {
        const long rb = r0 + 8 * hg;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const long r = rb + j;
            const bool ok = r < r_end;
            gr[j] = ok ? a.G[r * a.O + o] : 0.f;
            if (GPRO) ghr[j] = ok ? a.Gh[r * a.O + o] : mu;
            if (wave < NT) xr[j] = ok ? a.X[r * a.I + xi] : 0.f;
        }
    };
    fetch(r_begin);
    int stage = 0;
    for (long r0 = r_begin; r0 < r_end; r0 += 16, stage ^= 1) {
        const long rb = r0 + 8 * hg;
        // ---- this wave's A fragment: 8 rows of column o (straight from global), transformed and split ----
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = gr[j];
            if (GPRO) v = (rb + j < r_end) ? gs * (v - c1 - (ghr[j] - mu) * iv * c2) : 0.f;
            g[j] = v; bsum += v;
        }
        uint4 ah, al;
        tsplit8(g, ah, al);
        // ---- X fragment of tile (wave % NT), staged once per block through LDS ------------------------
        if (wave < NT) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = xr[j];
                if (XPRO) v = (rb + j < r_end) ? fmaxf(fmaf(v, xs, xsh), 0.f) : 0.f;
                x[j] = v;
            }
            uint4 xh, xl;
            tsplit8(x, xh, xl);
            Bs[stage][wave][0][lane] = xh; Bs[stage][wave][1][lane] = xl;
        }
        if (r0 + 16 < r_end) fetch(r0 + 16);
        __syncthreads();                 // two LDS stages: the next iteration writes the other one, so one barrier per slice
        const bf16x8 fah = __builtin_bit_cast(bf16x8, ah), fal = __builtin_bit_cast(bf16x8, al);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 bh = __builtin_bit_cast(bf16x8, Bs[stage][t][0][lane]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, Bs[stage][t][1][lane]);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah, bl, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah, bh, acc[t], 0, 0, 0);
        }
    }
    // ---- partial outputs: Wp[chunk][o][i], bp[chunk][o] --------------------------------------------------
    float* Wp = a.Wpart + (long)blockIdx.y * a.O * a.I;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Wp[(long)(o0 + acc_row(r, hg)) * a.I + i0 + 32 * t + c] = acc[t][r];
    if (blockIdx.x % itiles == 0) {
        bsum += __shfl_xor(bsum, 32);
        if (hg == 0) a.bpart[(long)blockIdx.y * a.O + o] = bsum;
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ Wpart, const float* __restrict__ bpart, int chunks,
                                                           long n_w, int n_b, float* __restrict__ dW, float* __restrict__ db) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n_w) {
        float s = 0.f;
        for (int k = 0; k < chunks; ++k) s += Wpart[(long)k * n_w + idx];
        dW[idx] = s;
    } else if (idx < n_w + n_b && db) {
        const long o = idx - n_w;
        float s = 0.f;
        for (int k = 0; k < chunks; ++k) s += bpart[(long)k * n_b + o];
        db[o] = s;
    }
}

int ptta_linear_wgrad_chunks(long R, int* rchunk_out) {
    long rc = 512;
    if ((R + rc - 1) / rc > 64) rc = ((R + 63) / 64 + 15) / 16 * 16;         // at most 64 row chunks
    if (rchunk_out) *rchunk_out = (int)rc;
    return (int)((R + rc - 1) / rc);
}

int ptta_launch_linear_wgrad(LinWgradArgs a, float* dW, float* db, hipStream_t s) {
    if (a.O % 128 || !(a.I == 32 || a.I % 128 == 0) || a.R <= 0) return -22;
    int rchunk = 0;
    const int chunks = ptta_linear_wgrad_chunks(a.R, &rchunk);
    a.rchunk = rchunk;
    const int nt = a.I == 32 ? 1 : 4;
    dim3 grid((a.O / 128) * (a.I / (32 * nt)), chunks);
    const int key = (nt == 4 ? 100 : 0) + (a.Gh ? 10 : 0) + (a.xscale ? 1 : 0);
#define LW_(NT, GP, XP) hipLaunchKernelGGL((linear_wgrad_kernel<NT, GP, XP>), grid, dim3(256), 0, s, a)
    switch (key) {
        case 100: LW_(4, 0, 0); break; case 101: LW_(4, 0, 1); break; case 110: LW_(4, 1, 0); break; case 111: LW_(4, 1, 1); break;
        case 0: LW_(1, 0, 0); break; case 10: LW_(1, 1, 0); break;
        default: return -22;
    }
#undef LW_
    PTTA_CHECK_LAUNCH();
    const long n_w = (long)a.O * a.I;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n_w + a.O + 255) / 256)), dim3(256), 0, s, a.Wpart, a.bpart, chunks, n_w, a.O, dW, db);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// ---- prepare_loss (src/external_model_adapt.py:524-541): L = mean_r (2 - 2 <e_r/|e_r|, f_r/|f_r|>), F.normalize eps 1e-12;
// d L / d e_r = -2/R * (f^ - cos * e^) / |e|.  One wave per row of C columns (C % 64 == 0), block partial sums of the loss.
__global__ __launch_bounds__(256) void prepare_loss_kernel(const float* __restrict__ emb, const float* __restrict__ ref, long R, int C,
                                                           float* __restrict__ g_emb, float* __restrict__ part) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lsum = 0.f;
    for (long r = (long)blockIdx.x * 4 + wave; r < R; r += (long)gridDim.x * 4) {
        const float* e = emb + r * C; const float* f = ref + r * C;
        float dot = 0.f, ne = 0.f, nf = 0.f;
        for (int k = lane; k < C; k += 64) { const float x = e[k], y = f[k]; dot = fmaf(x, y, dot); ne = fmaf(x, x, ne); nf = fmaf(y, y, nf); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { dot += __shfl_xor(dot, o); ne += __shfl_xor(ne, o); nf += __shfl_xor(nf, o); }
        const float le = fmaxf(sqrtf(ne), 1e-12f), lf = fmaxf(sqrtf(nf), 1e-12f);
        const float cs = dot / (le * lf);
        lsum += 2.f - 2.f * cs;
        const float k0 = -2.f / (float)R / le;
        for (int k = lane; k < C; k += 64) g_emb[r * C + k] = k0 * (f[k] / lf - cs * e[k] / le);
    }
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void prepare_loss_finalize_kernel(const float* __restrict__ part, int n, long R, float* __restrict__ loss) {
    double s = 0.0;
    for (int k = threadIdx.x; k < n; k += 64) s += (double)part[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) *loss = (float)(s / (double)R);
}
int ptta_launch_prepare_loss(const float* emb, const float* ref, long R, int C, float* g_emb, float* part /* >= 1024 floats */, float* loss,
                             hipStream_t s) {
    if (C % 64 || R <= 0) return -22;
    long blocks = (R + 3) / 4; if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(prepare_loss_kernel, dim3((int)blocks), dim3(256), 0, s, emb, ref, R, C, g_emb, part);
    PTTA_CHECK_LAUNCH();
    hipLaunchKernelGGL(prepare_loss_finalize_kernel, dim3(1), dim3(64), 0, s, part, (int)blocks, R, loss);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// ---- _update_head (network_exp_msg_chn_adapt.py:701-703): t <- t * tau + s * (1 - tau), all target tensors in one launch
__global__ __launch_bounds__(256) void ema_multi_kernel(const PttaAdamEntry* __restrict__ tab, int nt, long total, const float* __restrict__ tau_dev) {
    const float tau = tau_dev[0], omt = tau_dev[1];      // tau and (1 - tau) as the host rounds them (python doubles -> fp32)
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        int lo = 0, hi = nt - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tab[mid].off <= idx) lo = mid; else hi = mid - 1; }
        const PttaAdamEntry e = tab[lo];
        const long k = idx - e.off;
        e.p[k] = __fadd_rn(__fmul_rn(e.p[k], tau), __fmul_rn(e.g[k], omt));      // p = target, g = online parameter; no fma contraction
    }
}
int ptta_launch_ema_multi(const PttaAdamEntry* tab_dev, int nt, long total, const float* tau_dev, hipStream_t s) {
    long blocks = (total + 255) / 256; if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(ema_multi_kernel, dim3((int)blocks), dim3(256), 0, s, tab_dev, nt, total, tau_dev);
    PTTA_CHECK_LAUNCH();
    return 0;
}
