// 3x3, 32 -> 32 channel convolutions of the MSG_CHN cascade as implicit GEMMs on the matrix cores.
//
// Replaces, for the hot path, every nn.Conv2d(32,32,3) / nn.ConvTranspose2d(32,32,3,s=2) call of
// RGBEncoder / DepthEncoder / DepthDecoder (network_exp_msg_chn_adapt.py:166-311), the meta layer
// conv1_rgb_meta (:1065-1071) and, with re-packed weights, their input gradients (autograd's
// conv backward-data in the reference, src/tta_main.py:632).
//
// Mapping (one wave64 = one 32-pixel x 32-channel output tile, K = 9 taps x 32 input channels):
//   D[pixel][cout] += A[pixel][k] * B[k][cout]
//   fp32 : v_mfma_f32_32x32x2_f32  (exact fp32 FMA chain), 144 MFMAs per tile
//   bf16 : v_mfma_f32_32x32x16_bf16 (fp32 accumulate),      18 MFMAs per tile
// Weights are wave-stationary: the whole 3x3x32x32 filter lives in VGPRs (144 fp32 / 72 bf16
// registers per lane) for the lifetime of a persistent wave that walks over many tiles, so the
// only per-tile traffic is the NHWC input lines (full 128-B / 64-B lines per pixel) and the
// output lines.  ReLU-before-conv, bias, bilinear x2 skip, ReLU-mask (backward) and the decoder's
// skip additions are fused (ptta_common.h Epi).
//
// Three geometries share the body: CONV_S1 (stride 1, pad 1), CONV_S2 (stride 2, pad 1) and
// CONV_T2 (transposed, stride 2, pad 1, output_padding 1; a tile = 32 outputs of one x-parity so
// that the active taps are wave-uniform).
#include "ptta_common.h"
#include "ptta_kernels.h"

template <typename T>
struct Conv32P {
    const T* in; int in_nb;
    const void* wpack;
    Epi<T> epi;
    int B, Hin, Win, Hout, Wout;
};

template <typename T> struct Frag;
template <> struct Frag<float> { typedef float4 A; };
template <> struct Frag<bf16_t> { typedef uint4 A; };

__device__ __forceinline__ float4 relu4(float4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}
__device__ __forceinline__ unsigned relu_bf2(unsigned v) {
    unsigned neg = (v >> 15) & 0x00010001u;        // sign bit of each packed bf16
    return v & ~(neg * 0xffffu);
}
__device__ __forceinline__ uint4 relu8(uint4 v) {
    v.x = relu_bf2(v.x); v.y = relu_bf2(v.y); v.z = relu_bf2(v.z); v.w = relu_bf2(v.w);
    return v;
}

template <typename T, int MODE, bool RELU>
__global__ __launch_bounds__(256) void conv32_mfma_kernel(Conv32P<T> p) {
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int NA = F32 ? 4 : 2;                  // 16-byte A fragments per tap
    typedef typename Frag<T>::A AF;
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    // ---- wave-stationary weights -------------------------------------------------------------
    AF w[9][NA];
    {
        const AF* wp = (const AF*)p.wpack;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int g = 0; g < NA; ++g) w[t][g] = wp[(t * NA + g) * 64 + lane];
    }

    const int Wt = (MODE == CONV_T2) ? p.Win : p.Wout;     // tile domain along x
    const int nseg = (Wt + 31) >> 5;
    const int npar = (MODE == CONV_T2) ? 2 : 1;
    const long nitems = (long)p.B * nseg * npar * p.Hout;
    const float sy = up_scale(p.Hout >> 1, p.Hout), sx = up_scale(p.Wout >> 1, p.Wout);

    for (long item = (long)blockIdx.x * 4 + wave; item < nitems; item += (long)gridDim.x * 4) {
        long t_ = item;
        const int y = (int)(t_ % p.Hout); t_ /= p.Hout;
        int xpar = 0;
        if (MODE == CONV_T2) { xpar = (int)(t_ & 1); t_ >>= 1; }
        const int seg = (int)(t_ % nseg);
        const int b = (int)(t_ / nseg);
        const int x0 = seg << 5;
        const bool lane_in = (x0 + i) < Wt;
        const T* inb = p.in + (size_t)(b % p.in_nb) * p.Hin * p.Win * 32;

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;

#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            bool active = true;
            int yi, xi;
            if (MODE == CONV_S1) { yi = y + ky - 1; xi = x0 + i + kx - 1; }
            else if (MODE == CONV_S2) { yi = 2 * y + ky - 1; xi = 2 * (x0 + i) + kx - 1; }
            else {
                const int ty = y + 1 - ky, tx = xpar + 1 - kx;
                active = ((ty & 1) == 0) && ((tx & 1) == 0);
                yi = ty >> 1; xi = x0 + i + (tx >> 1);
            }
            active = active && (yi >= 0) && (yi < p.Hin);
            if (!active) continue;                                    // wave-uniform
            const bool ok = lane_in && (xi >= 0) && (xi < p.Win);
            const T* src = inb + ((size_t)yi * p.Win + (ok ? xi : 0)) * 32;
            AF a[NA];
#pragma unroll
            for (int g = 0; g < NA; ++g) {
                if (F32) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ok) v = *(const float4*)((const float*)src + 8 * g + 4 * h);
                    if (RELU) v = relu4(v);
                    a[g] = *(AF*)&v;
                } else {
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (ok) v = *(const uint4*)((const bf16_t*)src + 16 * g + 8 * h);
                    if (RELU) v = relu8(v);
                    a[g] = *(AF*)&v;
                }
            }
#pragma unroll
            for (int g = 0; g < NA; ++g) {
                if (F32) {
                    const float4 av = *(float4*)&a[g];
                    const float4 wv = *(float4*)&w[tap][g];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wv.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wv.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wv.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wv.w, acc, 0, 0, 0);
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[g]),
                                                                  __builtin_bit_cast(bf16x8, w[tap][g]),
                                                                  acc, 0, 0, 0);
                }
            }
        }

        // ---- epilogue: lane = channel (lane&31), registers = 16 pixels of the tile --------------
        Lerp ly = {0, 0, 0.f, 0.f};
        if (p.epi.up) ly = lerp_coef(y, p.Hout >> 1, sy);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = x0 + acc_row(r, h);
            if (px >= Wt) continue;
            const int xo = (MODE == CONV_T2) ? 2 * px + xpar : px;
            Lerp lx = {0, 0, 0.f, 0.f};
            if (p.epi.up) lx = lerp_coef(xo, p.Wout >> 1, sx);
            epi_store<T>(p.epi, b, y, xo, p.Hout, p.Wout, i, acc[r], ly, lx);
        }
    }
}

// ---- reference-style direct kernel (one thread per output element), same epilogue ------------
// Selected with PTTA_CONV_IMPL=naive: keeps the whole pipeline testable independently of the MFMA
// fragment packing.  Weights in canonical float layout Wc[tap][cin][cout].
template <typename T, int MODE, bool RELU>
__global__ void conv32_naive_kernel(Conv32P<T> p) {
    const long total = (long)p.B * p.Hout * p.Wout * 32;
    const float* wc = (const float*)p.wpack;
    const float sy = up_scale(p.Hout >> 1, p.Hout), sx = up_scale(p.Wout >> 1, p.Wout);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx & 31);
        long t_ = idx >> 5;
        const int x = (int)(t_ % p.Wout); t_ /= p.Wout;
        const int y = (int)(t_ % p.Hout);
        const int b = (int)(t_ / p.Hout);
        const T* inb = p.in + (size_t)(b % p.in_nb) * p.Hin * p.Win * 32;
        float acc = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            int yi, xi;
            if (MODE == CONV_S1) { yi = y + ky - 1; xi = x + kx - 1; }
            else if (MODE == CONV_S2) { yi = 2 * y + ky - 1; xi = 2 * x + kx - 1; }
            else {
                const int ty = y + 1 - ky, tx = x + 1 - kx;
                if ((ty & 1) || (tx & 1)) continue;
                yi = ty >> 1; xi = tx >> 1;
            }
            if (yi < 0 || yi >= p.Hin || xi < 0 || xi >= p.Win) continue;
            const T* src = inb + ((size_t)yi * p.Win + xi) * 32;
            for (int ci = 0; ci < 32; ++ci) {
                float v = ld(src + ci);
                if (RELU) v = fmaxf(v, 0.f);
                acc = fmaf(v, wc[(tap * 32 + ci) * 32 + co], acc);
            }
        }
        Lerp ly = {0, 0, 0.f, 0.f}, lx = {0, 0, 0.f, 0.f};
        if (p.epi.up) { ly = lerp_coef(y, p.Hout >> 1, sy); lx = lerp_coef(x, p.Wout >> 1, sx); }
        epi_store<T>(p.epi, b, y, x, p.Hout, p.Wout, co, acc, ly, lx);
    }
}

// ---- weight packing (device side, so adapted parameters can be re-packed every step) ----------
// src: a Conv2d weight [out][in][3][3] or ConvTranspose2d weight [in][out][3][3] (fp32, NCHW).
// in_major: src is indexed [cin_eff][cout_eff]; flip: use tap (2-ky, 2-kx).  See DESIGN.md §4 for
// which (in_major, flip) pair each forward/backward use needs.
__global__ void pack_conv32_kernel(const float* __restrict__ src, float* mf32, bf16_t* mbf16, float* canon,
                                   int in_major, int flip) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // over tap*32*32
    if (idx >= 9 * 32 * 32) return;
    const int co = idx & 31, ci = (idx >> 5) & 31, tap = idx >> 10;
    const int st = flip ? 8 - tap : tap;
    const float v = in_major ? src[(ci * 32 + co) * 9 + st] : src[(co * 32 + ci) * 9 + st];
    canon[(tap * 32 + ci) * 32 + co] = v;
    {   // fp32 fragments: [tap][g][lane][s], lane = h*32 + cout, cin = 8g + 4h + s
        const int g = ci >> 3, hh = (ci >> 2) & 1, s = ci & 3;
        mf32[((tap * 4 + g) * 64 + hh * 32 + co) * 4 + s] = v;
    }
    {   // bf16 fragments: [tap][kk][lane][e], cin = 16kk + 8h + e
        const int kk = ci >> 4, hh = (ci >> 3) & 1, e = ci & 7;
        mbf16[((tap * 2 + kk) * 64 + hh * 32 + co) * 8 + e] = f2bf(v);
    }
}

void ptta_pack_conv32(const float* src, float* mf32, bf16_t* mbf16, float* canon, int in_major, int flip,
                      hipStream_t s) {
    hipLaunchKernelGGL(pack_conv32_kernel, dim3(36), dim3(256), 0, s, src, mf32, mbf16, canon, in_major, flip);
}

template <typename T, int MODE>
static int launch_conv32_t(const Conv32Args& a, hipStream_t s) {
    Conv32P<T> p;
    p.in = (const T*)a.in; p.in_nb = a.in_nb;
    p.epi.bias = a.bias;
    p.epi.up = (const T*)a.up; p.epi.up_nb = a.up_nb > 0 ? a.up_nb : 1;
    p.epi.mask = (const T*)a.mask; p.epi.mask_nb = a.mask_nb > 0 ? a.mask_nb : 1;
    p.epi.add1 = (const T*)a.add1; p.epi.add1_nb = a.add1_nb > 0 ? a.add1_nb : 1;
    p.epi.add2 = (const T*)a.add2; p.epi.add2_nb = a.add2_nb > 0 ? a.add2_nb : 1;
    p.epi.out_raw = (T*)a.out_raw; p.epi.out_sum = (T*)a.out_sum;
    p.B = a.B; p.Hin = a.Hin; p.Win = a.Win;
    if (MODE == CONV_S1) { p.Hout = a.Hin; p.Wout = a.Win; }
    else if (MODE == CONV_S2) { p.Hout = a.Hin / 2; p.Wout = a.Win / 2; }
    else { p.Hout = a.Hin * 2; p.Wout = a.Win * 2; }
    if (a.naive) {
        p.wpack = a.w->canon;
        const long total = (long)p.B * p.Hout * p.Wout * 32;
        int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
        if (a.relu_in) hipLaunchKernelGGL((conv32_naive_kernel<T, MODE, true>), dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv32_naive_kernel<T, MODE, false>), dim3(blocks), dim3(256), 0, s, p);
    } else {
        p.wpack = sizeof(T) == 4 ? (const void*)a.w->mf32 : (const void*)a.w->mbf16;
        const int Wt = (MODE == CONV_T2) ? p.Win : p.Wout;
        const long items = (long)p.B * ((Wt + 31) / 32) * (MODE == CONV_T2 ? 2 : 1) * p.Hout;
        long blocks = (items + 3) / 4;
        const long cap = sizeof(T) == 4 ? 512 : 1024;     // persistent waves: 2 (fp32) / 4 (bf16) blocks per CU
        if (blocks > cap) blocks = cap;
        if (a.relu_in) hipLaunchKernelGGL((conv32_mfma_kernel<T, MODE, true>), dim3((int)blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv32_mfma_kernel<T, MODE, false>), dim3((int)blocks), dim3(256), 0, s, p);
    }
    PTTA_CHECK_LAUNCH();
    return 0;
}

int ptta_launch_conv32(const Conv32Args& a, hipStream_t s) {
    if (a.bf16) {
        if (a.mode == CONV_S1) return launch_conv32_t<bf16_t, CONV_S1>(a, s);
        if (a.mode == CONV_S2) return launch_conv32_t<bf16_t, CONV_S2>(a, s);
        return launch_conv32_t<bf16_t, CONV_T2>(a, s);
    }
    if (a.mode == CONV_S1) return launch_conv32_t<float, CONV_S1>(a, s);
    if (a.mode == CONV_S2) return launch_conv32_t<float, CONV_S2>(a, s);
    return launch_conv32_t<float, CONV_T2>(a, s);
}
