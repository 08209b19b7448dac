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
#include <cstdlib>
#include <cstring>
#include "ptta_common.h"
#include "ptta_kernels.h"

template <typename T>
struct Conv32P {
    const T* in; int in_nb;
    const void* wpack;
    const void* wpack2;          // lo fragments (bf16x3 arithmetic)
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

template <typename T, int MODE, bool RELU, bool UP, bool MASK, bool ADD>
__global__ __launch_bounds__(256) void conv32_mfma_kernel(Conv32P<T> p) {
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int NA = F32 ? 4 : 2;                  // 16-byte A fragments per tap
    typedef typename Frag<T>::A AF;
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const float biasv = epi_bias(p.epi, i);               // once per kernel: ptta_common.h epi_tile
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
        epi_tile<T, UP, MASK, ADD>(p.epi, b, y, p.Hout, p.Wout, i, acc, x0, h, Wt, MODE == CONV_T2 ? 2 : 1, xpar, sy, sx, nullptr, 0, 0, &biasv);
    }
}


// ======================================================================================================================
// The tuned kernels.  Every one is a template over the STORAGE type T of its input, output and skip operands:
//   T = float   fp32 maps, "bf16x3" arithmetic: x = xh + xl, w = wh + wl (each half a bf16), x*w ~= xl*wh + xh*wl + xh*wh (the dropped
//               xl*wl term is 2^-16 relative), fp32 accumulate -- the REAL frames' forward, whose output is the scored depth map;
//   T = bf16_t  narrow maps, ONE bf16 MFMA per product ("x1"), fp32 accumulate, one rounding per stored value -- the tensors the reference
//               computes under no_grad / detaches (the zero-image proxy pass, network_exp_msg_chn_adapt.py:509-532) and the data gradients
//               of loss.backward() (src/tta_main.py:632), which reach the scored depth only through an lr-sized Adam move
//               (profiles/r05_precision_budget.txt).  Half the bytes per pixel, a third of the MFMAs, no operand split, no lo planes.
// The two instantiations share every line of tile geometry, prefetch and epilogue structure; they differ in Px<T> (8 channels of a pixel in
// registers), px_stage (ReLU + split while a pixel is written to LDS) and the number of MFMAs per fragment pair.
// ======================================================================================================================
#define X3_TH 8
#define X3_PW 34
#define X3_PH 10

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef short short2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    float2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);          // v_cvt_pk_bf16_f32 (RNE)
    hi = __builtin_bit_cast(unsigned, h);
    float2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}
// ReLU of two packed bf16: sign-magnitude values ordered as int16 -> one v_pk_max_i16 against zero (-0.0 -> +0.0)
__device__ __forceinline__ unsigned relu_pk(unsigned v) {
    const short2_t z = {0, 0};
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2_t, v), z));
}

// 8 consecutive channels of one NHWC pixel in registers
template <typename T> struct Px;
template <> struct Px<float> { float4 a, b; };
template <> struct Px<bf16_t> { uint4 a; };
template <typename T> __device__ __forceinline__ Px<T> px_zero();
template <> __device__ __forceinline__ Px<float> px_zero<float>() { Px<float> r; r.a = make_float4(0.f, 0.f, 0.f, 0.f); r.b = r.a; return r; }
template <> __device__ __forceinline__ Px<bf16_t> px_zero<bf16_t>() { Px<bf16_t> r; r.a = make_uint4(0u, 0u, 0u, 0u); return r; }
__device__ __forceinline__ Px<float> px_load(const float* src) { Px<float> r; r.a = *(const float4*)src; r.b = *(const float4*)(src + 4); return r; }
__device__ __forceinline__ Px<bf16_t> px_load(const bf16_t* src) { Px<bf16_t> r; r.a = *(const uint4*)src; return r; }
// (ReLU,) bf16 hi / lo split, 16 B of hi fragments to `dst` and -- fp32 storage -- 16 B of lo fragments to `dst + lo_off`
template <bool RELU>
__device__ __forceinline__ void px_stage(const Px<float>& v, unsigned char* dst, int lo_off) {
    float4 a0 = v.a, a1 = v.b;
#ifdef X3_TIMING_NOSPLIT
    // timing-only (wrong values): the staged pixel taken as if the producer had stored it split -- the upper bound of a pre-split storage format
    *(float4*)dst = a0; *(float4*)(dst + lo_off) = a1; return;
#endif
    if (RELU) { a0 = relu4(a0); a1 = relu4(a1); }
    uint4 hi, lo;
    split2(a0.x, a0.y, hi.x, lo.x); split2(a0.z, a0.w, hi.y, lo.y);
    split2(a1.x, a1.y, hi.z, lo.z); split2(a1.z, a1.w, hi.w, lo.w);
    *(uint4*)dst = hi;
    *(uint4*)(dst + lo_off) = lo;
}
template <bool RELU>
__device__ __forceinline__ void px_stage(const Px<bf16_t>& v, unsigned char* dst, int) {
    uint4 hi = v.a;
    if (RELU) { hi.x = relu_pk(hi.x); hi.y = relu_pk(hi.y); hi.z = relu_pk(hi.z); hi.w = relu_pk(hi.w); }
    *(uint4*)dst = hi;
}
// LDS pixel of the stride-1 halo tile: fp32 storage [hi 64 B | lo 64 B | pad 16 B], narrow [hi 64 B | pad 16 B]; both strides keep
// ds_read_b128 of 16 consecutive pixels on disjoint banks (144: bank = 4 (9 p mod 16) + const; 80: 20 p mod 64 steps through all 16 quads)
template <typename T> struct Geo;
template <> struct Geo<float> { static constexpr int STR = 144, LO = 64, WL = 18 * 64 * 16, UPPX = 128, UPPC = 8; };
template <> struct Geo<bf16_t> { static constexpr int STR = 80, LO = 0, WL = 0, UPPX = 64, UPPC = 4; };

// staging item -> halo pixel: the 8 lanes of one ds_write_b128 group take pixels p and p+4 (not p, p+1):
// at either pixel stride their footprints then fall on disjoint banks
__device__ __forceinline__ int x3_stage_pix(int idx) { return ((idx >> 5) << 3) + (((idx >> 2) & 1) << 2) + ((idx >> 3) & 3); }

// one (tap, k-step) of the implicit GEMM: a = the lane's A fragment in the LDS tile (hi at a, lo at a + LO); the hi weight fragment is
// wave-stationary in registers (wh), the lo fragment (fp32 storage) comes from LDS (wl).
// (Round 5 measured the narrow kernels with their ONLY weight fragment in LDS as well -- 72 registers less, three resident blocks per CU
// without spills: the full-resolution masked stride-1 launch went from 16.2 to 19.3 us.  At one MFMA per fragment pair the A fragment's
// 1 KB per wave and matrix instruction already takes half of the LDS's 256 B/clk; a second KB for the B fragment saturates it.)
// W2 (narrow storage only): the WEIGHT operand as hi + lo (two MFMAs per product, a . w_lo first) -- the data gradients of the mixed mode.
// A bf16-rounded weight is a SYSTEMATIC 2^-9 error of the gradient's direction, the same for every pixel and every step (a rounded gradient
// value is noise that the sum over the pixels averages out): over a 150-step horizon it was what separated the mixed mode from the
// reference (profiles/r06_drift.txt; CPU simulation tools/precision_sim.py bwd=xw vs bwd=xas).
template <typename T, bool W2 = false>
__device__ __forceinline__ void mma_step(f32x16& acc, const unsigned char* a, const uint4& wh, const unsigned char* wl) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)a);
    const bf16x8 bh = __builtin_bit_cast(bf16x8, wh);
    if constexpr (sizeof(T) == 2 && W2) {
        const bf16x8 bl = __builtin_bit_cast(bf16x8, *(const uint4*)wl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    }
    if constexpr (sizeof(T) == 4) {
        const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(a + Geo<T>::LO));
        const bf16x8 bl = __builtin_bit_cast(bf16x8, *(const uint4*)wl);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);   // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}
// the block's weight fragments: hi in registers (wh); fp32 storage: lo in LDS
template <typename T, bool W2 = false>
__device__ __forceinline__ void load_weights(uint4 (&wh)[9][2], unsigned char* wl_lds, const void* wpack, const void* wpack2, int tid, int lane) {
    const uint4* ph = (const uint4*)wpack;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 2; ++k) wh[t][k] = ph[(t * 2 + k) * 64 + lane];
    if constexpr (sizeof(T) == 4 || W2) {
        const uint4* pl = (const uint4*)wpack2;
        for (int idx = tid; idx < 18 * 64; idx += 256) *(uint4*)(wl_lds + idx * 16) = pl[idx];
    }
}

// the LDS operands of one (tap, k) step in registers and its products -- mma_step split in two so that the reads of a later step can be issued
// in front of the MFMAs of an earlier one (same products in the same order as mma_step)
template <typename T, bool W2> struct FragSet { uint4 ah, al, bl; };
template <typename T, bool W2>
__device__ __forceinline__ void frag_load(FragSet<T, W2>& f, const unsigned char* a, const unsigned char* wl) {
    f.ah = *(const uint4*)a;
    if constexpr (sizeof(T) == 4) { f.al = *(const uint4*)(a + Geo<T>::LO); f.bl = *(const uint4*)wl; }
    else if constexpr (W2) f.bl = *(const uint4*)wl;
}
template <typename T, bool W2>
__device__ __forceinline__ void frag_mma(f32x16& acc, const FragSet<T, W2>& f, const uint4& wh) {
    const bf16x8 ah = __builtin_bit_cast(bf16x8, f.ah), bh = __builtin_bit_cast(bf16x8, wh);
    if constexpr (sizeof(T) == 2 && W2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, f.bl), acc, 0, 0, 0);
    if constexpr (sizeof(T) == 4) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.al), bh, acc, 0, 0, 0);   // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, f.bl), acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
}

// bilinear x2 skip: the half-resolution source window of a tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers,
// one 1-KB piece = 8 (fp32) / 16 (narrow) source pixels per wave instruction, lane-linear image [pixel][32 channels], clamped at the borders)
template <typename T, int UPH, int UPW>
__device__ __forceinline__ void up_window_dma(const T* ub, int Hu, int Wu, int uy0, int ux0, unsigned char* up_lds, int tid, int wave) {
    constexpr int PC = Geo<T>::UPPC, N = UPH * UPW * PC;
#pragma unroll
    for (int k = 0; k < (N + 255) / 256; ++k) {
        const int idx = tid + 256 * k;
        if (idx < N) {
            const int q = idx % PC, pix = idx / PC;
            const int r = pix / UPW, cc = pix - r * UPW;
            const int gy = min(uy0 + r, Hu - 1), gx = min(ux0 + cc, Wu - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ub + ((size_t)gy * Wu + gx) * 32 + (32 / PC) * q),
                                             (__attribute__((address_space(3))) void*)(up_lds + (size_t)(256 * k + 64 * wave) * 16), 16, 0, 0);
        }
    }
}

// ---- stride-1, LDS-staged -------------------------------------------------------------------------------------------------
// A block owns an 8x32 output tile: the (8+2)x(32+2) input halo is read from HBM once (full NHWC lines), ReLU'd and split ONCE while it is
// staged into LDS.  Each wave computes two 32-pixel rows; the hi weight fragments (72 VGPRs) stay in registers across the persistent tile
// loop, the lo fragments (fp32 storage) in LDS.  Without the bilinear epilogue the NEXT tile's global loads are issued before this tile's
// MFMAs and land while the matrix cores work (software prefetch across the persistent tile loop).
// TIMING (make DIAG=1 only): s_memtime stamps around the phases of the tile loop, printed by two blocks of the launch
#define X3STAMP(v) do { if constexpr (TIMING) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); v = t_; } } while (0)
template <typename T, bool RELU, bool UP, bool MASK, bool ADD, bool W2 = false, bool TIMING = false>
__global__ __launch_bounds__(256, 2) void conv32_s1_x3_kernel(Conv32P<T> p) {
    unsigned long long T0 = 0, ta = 0, tb = 0, tc = 0, td = 0, te = 0, tf = 0, tg = 0, th = 0;
    unsigned long long dWait = 0, dSplit = 0, dIssue = 0, dBar1 = 0, dMfma = 0, dEpi = 0, dBar2 = 0, dPro = 0;
    int ntl = 0;
    X3STAMP(T0);
    constexpr int STR = Geo<T>::STR, WL = (sizeof(T) == 4 || W2) ? 18 * 64 * 16 : 0;
    constexpr int UPH = 6, UPW = 18;                      // an 8x32 output tile reads <= 5x17 source pixels
    __shared__ __attribute__((aligned(16))) unsigned char lds[X3_PH * X3_PW * STR + WL + (UP ? UPH * UPW * Geo<T>::UPPX : 0)];
    unsigned char* const wl_lds = lds + X3_PH * X3_PW * STR;
    unsigned char* const up_lds = wl_lds + WL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const float biasv = epi_bias(p.epi, i);               // once per kernel: ptta_common.h epi_tile
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.Hout, W = p.Wout;
    const int ntx = (W + 31) >> 5, nty = (H + X3_TH - 1) / X3_TH;
    const long ntiles = (long)p.B * ntx * nty;
    const float sy = up_scale(H >> 1, H), sx = up_scale(W >> 1, W);

    constexpr int NIT = (((X3_PH * X3_PW + 7) / 8) * 32 + 255) / 256;     // (pixel, 8-channel group) staging items per thread
    constexpr bool PREFETCH = !UP;       // (the bilinear epilogue leaves no registers for it: 13 spills and no gain when forced)
    Px<T> v[NIT];
    auto tile_coords = [&](long tile, int& b, int& y0, int& x0) {
        long t_ = tile;
        const int ty = (int)(t_ % nty); t_ /= nty;          // y fastest: neighbouring blocks share halo rows in L2
        const int tx = (int)(t_ % ntx);
        b = (int)(t_ / ntx); y0 = ty * X3_TH; x0 = tx << 5;
    };
    auto issue_loads = [&](long tile) {
        int b, y0, x0;
        tile_coords(tile, b, y0, x0);
        const T* inb = p.in + (size_t)(b % p.in_nb) * H * W * 32;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int g = idx & 3, pix = x3_stage_pix(idx);
            const int py = pix / X3_PW, px = pix - py * X3_PW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            v[it] = px_zero<T>();
            if (pix < X3_PH * X3_PW && gy >= 0 && gy < H && gx >= 0 && gx < W) v[it] = px_load(inb + ((size_t)gy * W + gx) * 32 + 8 * g);
        }
    };
    if (PREFETCH && blockIdx.x < ntiles) issue_loads(blockIdx.x);
    // weight fragments: loaded AFTER the first tile's loads were issued so that the two L2 round trips overlap
    uint4 wh[9][2];
    load_weights<T, W2>(wh, wl_lds, p.wpack, p.wpack2, tid, lane);
    X3STAMP(ta); dPro = ta - T0;

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b, y0, x0;
        tile_coords(tile, b, y0, x0);
        int uy0 = 0, ux0 = 0;
        X3STAMP(ta);
        if constexpr (TIMING) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        X3STAMP(tb);
        if (UP) {
            // issued FIRST so that it flies while this tile's halo is split and written; drained before the next tile's register prefetch
            // is issued (vmcnt is in order: a later wait for the window would also wait for that prefetch)
            const int Hu = H >> 1, Wu = W >> 1;
            uy0 = lerp_coef(y0, Hu, sy).i0; ux0 = lerp_coef(x0, Wu, sx).i0;
            up_window_dma<T, UPH, UPW>(p.epi.up + (size_t)(b % p.epi.up_nb) * Hu * Wu * 32, Hu, Wu, uy0, ux0, up_lds, tid, wave);
        }
        if (!PREFETCH) issue_loads(tile);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int pix = x3_stage_pix(idx);
            if (pix < X3_PH * X3_PW) px_stage<RELU>(v[it], lds + pix * STR + 16 * (idx & 3), Geo<T>::LO);
        }
        X3STAMP(tc);
        if (UP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the window has landed (and this tile's halo loads before it)
        // narrow masked epilogue: both rows' mask words are requested BEFORE the next tile's prefetch (ptta_common.h epi_mask_words)
        constexpr bool MWPRE = MASK && sizeof(T) == 2;
        uint32_t mwp0[4], mwp1[4];         // (two arrays and a select below: the row loop is not unrolled, an array indexed by the row lives in scratch)
        if constexpr (MWPRE) {
            epi_mask_words<T>(p.epi, b, y0 + 2 * wave, H, W, i, x0, h, mwp0);
            epi_mask_words<T>(p.epi, b, y0 + 2 * wave + 1, H, W, i, x0, h, mwp1);
        }
        if (PREFETCH && tile + gridDim.x < ntiles) issue_loads(tile + gridDim.x);
        X3STAMP(td);
        lds_barrier();          // LDS-only: the next tile's global loads (issued above) stay in flight during the MFMAs
        X3STAMP(te);
        // ---- two output rows per wave ------------------------------------------------------------
        auto do_row = [&](const int rr) __attribute__((always_inline)) {
            const int row = 2 * wave + rr;
            const int y = y0 + row;
            if (y >= H) return;                                     // wave-uniform
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // The fragment reads run AHEAD steps in front of their MFMAs (one step = the (tap, k) pair's reads and products).  Left to hipcc each
            // ds_read_b128 is issued right in front of the MFMA that consumes it, behind s_waitcnt lgkmcnt(0): the LDS latency exposed every two or
            // three matrix instructions, 65 - 75 cycles per MFMA instead of 32 (in-kernel stamps, round 6: the MFMA phase was 7 - 8 k of a tile's
            // 14 k cycles).  The order is fixed by scheduling barriers around each step; the waits are the compiler's own counted lgkmcnt.
            {
                constexpr int AHEAD = sizeof(T) == 4 ? 1 : (W2 ? 2 : 4), RING = AHEAD + 1;
                const unsigned char* const arow = lds + (row * X3_PW + i) * STR + 16 * h;
                const unsigned char* const wrow = wl_lds + lane * 16;
                FragSet<T, W2> fr[RING];
#pragma unroll
                for (int s_ = 0; s_ < AHEAD; ++s_) frag_load<T, W2>(fr[s_], arow + ((s_ / 6) * X3_PW + (s_ / 2) % 3) * STR + 32 * (s_ & 1), wrow + s_ * 1024);
#pragma unroll
                for (int s_ = 0; s_ < 18; ++s_) {
                    if (s_ + AHEAD < 18) {
                        const int n_ = s_ + AHEAD;
                        frag_load<T, W2>(fr[n_ % RING], arow + ((n_ / 6) * X3_PW + (n_ / 2) % 3) * STR + 32 * (n_ & 1), wrow + n_ * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    frag_mma<T, W2>(acc, fr[s_ % RING], wh[s_ >> 1][s_ & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (TIMING) { X3STAMP(tf); dMfma += tf - (rr ? tg : te); }
            // (conv + bias) + bilinear, in the reference's order; the skip comes from the LDS window, added after the quad transpose
            if (UP) epi_tile<T, false, MASK, ADD, true, UPW>(p.epi, b, y, H, W, i, acc, x0, h, W, 1, 0, sy, sx, up_lds, uy0, ux0, &biasv);
            else {
                uint32_t mws[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) mws[g] = MWPRE ? (rr ? mwp1[g] : mwp0[g]) : 0u;
                epi_tile<T, false, MASK, ADD>(p.epi, b, y, H, W, i, acc, x0, h, W, 1, 0, sy, sx, nullptr, 0, 0, &biasv, MWPRE ? mws : nullptr);
            }
            if constexpr (TIMING) { X3STAMP(tg); dEpi += tg - tf; }
        };
        // (narrow masked variants: the two rows as straight-line code -- inside a loop the compiler's wait-count bookkeeping merges with the
        // back edge and waits vmcnt(0) for the preloaded mask words, i.e. for the prefetch issued behind them)
        if constexpr (MWPRE) { do_row(0); do_row(1); }
        else {
#pragma unroll 1
            for (int rr = 0; rr < 2; ++rr) do_row(rr);
        }
        lds_barrier();          // LDS reuse only: do not wait for this tile's stores (nor the prefetch) to drain
        if constexpr (TIMING) { X3STAMP(th); dWait += tb - ta; dSplit += tc - tb; dIssue += td - tc; dBar1 += te - td; dBar2 += th - tg; ++ntl; }
    }
    if constexpr (TIMING) {
        X3STAMP(th);
        if (wave == 0 && lane == 0 && (blockIdx.x < 20 || (blockIdx.x >= 256 && blockIdx.x < 276))) {
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            printf("x3map blk %d xcc %u se %u sh %u cu %u simd %u waveslot %u\n", (int)blockIdx.x, xcc & 15, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 15, (hwid >> 4) & 3, hwid & 15);
        }
        if ((blockIdx.x == 102 || blockIdx.x == 307) && lane == 0 && ntl > 0)
            printf("x3 blk %d wave %d tiles %d: life %llu pro %llu | per tile: vmwait %llu split+write %llu issue %llu barrier1 %llu mfma %llu epilogue %llu barrier2 %llu\n",
                   (int)blockIdx.x, wave, ntl, th - T0, dPro, dWait / ntl, dSplit / ntl, dIssue / ntl, dBar1 / ntl, dMfma / ntl, dEpi / ntl, dBar2 / ntl);
    }
}

// ---- the first two layers of an encoder stage in ONE launch: Conv2d(cin <= 3, 32) - ReLU - Conv2d(32, 32) (RGBEncoder.init / DepthEncoder.init,
// network_exp_msg_chn_adapt.py:172-176, :218-222) ----
// The 32-channel map between the two convolutions never makes a round trip through HBM as the second convolution's INPUT: a block
// stages the (8 + 4) x (32 + 4) window of the cin input planes (<= 5 KB instead of a 43.5-KB 32-channel halo), computes the first
// convolution for its (8 + 2) x (32 + 2) halo on the matrix cores exactly as conv_in_lds_kernel does (v_mfma_f32_32x32x2_f32 over
// k = 9 cin, the same k order: bit-identical values, 1.33 x recomputed), applies the zero padding of the SECOND convolution, ReLU and the
// bf16 split on the accumulators (lane-quad transpose: four channels of one pixel per lane, 8-byte LDS stores) and then runs the
// body of conv32_s1_x3_kernel on that LDS tile.  The pre-activation map is still written for the frames whose backward needs it as a
// ReLU mask (`a_out`, frames b < a_nb: the real frames; not at all for the RGB encoder).  The input-plane window lives in the LDS
// region of the bilinear-skip window (dead until the matrix phase); the next tile's window is prefetched into registers.
// The same construction serves the backward of a prediction head: d v = conv^T_{1 -> 32}(g) * (v > 0) followed by the data gradient of
// prdct.1 (a stride-1 32 -> 32 convolution with a ReLU-mask epilogue) -- FMASK: the first convolution's output is masked by the sign bits
// of a 32-channel map (one word per halo pixel, applied after the lane-quad transpose), RELU_MID = false: no ReLU in between, EMASK: the second convolution's
// epilogue mask.  The 55-MB d v map is neither written nor read.
// T = storage / arithmetic of the SECOND convolution and of the output (the input planes are planar fp32 either way).
struct FirstP {
    Plane pl[3]; int zero_from_b;
    const float* w1;             // [14][64] fp32 MFMA fragments of the first convolution (ptta_pack_conv_in)
    const float* bias1;          // may be null
    float* a_out; int a_nb;
    const uint32_t* fmask_bits; int fmask_nb;        // FMASK: sign bits of the first convolution's ReLU mask (one word per pixel), frames b % fmask_nb
    uint32_t* a_bits;                        // sign bits of the a_out map for the same frames (the backward's mask), or null
};
template <typename T, int CIN, bool UP, bool FMASK = false, bool RELU_MID = true, bool EMASK = false, bool EADD = false, bool W2 = false>
__global__ __launch_bounds__(256, 2) void conv32_s1_first_kernel(Conv32P<T> p, FirstP f) {
    constexpr int STR = Geo<T>::STR, WL = (sizeof(T) == 4 || W2) ? 18 * 64 * 16 : 0;
    constexpr int UPH = 6, UPW = 18;
    constexpr int K1 = 9 * CIN, NS = (K1 + 1) / 2;
    constexpr int PL_W = 36, PL_H = 12, PLANE = PL_H * PL_W;
    constexpr int PLSZ = (CIN * PLANE + 4) * 4, USZ = UP ? UPH * UPW * Geo<T>::UPPX : 0, AUX = USZ > PLSZ ? USZ : PLSZ;
    constexpr int NPJ = (PLANE + 255) / 256, NPL = CIN * NPJ;       // items per thread: NPJ of each plane (the plane index is a compile-time constant
                                                                    // of every item: f.pl[ci] stays in scalar registers -- indexed by a lane-varying
                                                                    // ci it was fetched from memory, a dependent load in front of every data load)
    __shared__ __attribute__((aligned(16))) unsigned char lds[X3_PH * X3_PW * STR + WL + AUX];
    unsigned char* const wl_lds = lds + X3_PH * X3_PW * STR;
    unsigned char* const up_lds = wl_lds + WL;
    float* const planes = (float*)up_lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const float biasv = epi_bias(p.epi, i);               // once per kernel: ptta_common.h epi_tile
    // per-lane index arithmetic is recomputed in each phase from an opaque copy of the lane id: hoisted out of the persistent tile loop it
    // becomes dozens of long-lived registers (the UP variants spilled 5 - 34) and a scratch reload between two global loads serialises them
    auto opaque = [](int v) { asm volatile("" : "+v"(v)); return v; };
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.Hout, W = p.Wout;
    const int ntx = (W + 31) >> 5, nty = (H + X3_TH - 1) / X3_TH;
    const long ntiles = (long)p.B * ntx * nty;
    const float sy = up_scale(H >> 1, H), sx = up_scale(W >> 1, W);
    auto tile_coords = [&](long tile, int& b, int& y0, int& x0) {
        long t_ = tile;
        const int ty = (int)(t_ % nty); t_ /= nty;
        const int tx = (int)(t_ % ntx);
        b = (int)(t_ / ntx); y0 = ty * X3_TH; x0 = tx << 5;
    };
    // the window's loads are UNCONDITIONAL (clamped address) and RAW: zero padding and the optional photometric normalisation are applied when
    // the values go to LDS one tile later.  With `if (inside) { v = load; if (norm) v = ... }` every load was followed by its own
    // s_waitcnt (the normalisation consumes the value inside the branch): four memory round trips in a row in front of every tile's second
    // convolution (round 5, .s)
    float pv[NPL];
    unsigned pvok = 0;
    auto load_planes = [&](long tile) {
        int b, y0, x0;
        tile_coords(tile, b, y0, x0);
        const bool live = b < f.zero_from_b;
        const int tid_ = opaque(tid);
        pvok = 0;
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const Plane& pl = f.pl[ci];
            const float* pb = pl.p + (size_t)(b % pl.nb) * pl.bstride;
#pragma unroll
            for (int j = 0; j < NPJ; ++j) {
                const int r = tid_ + 256 * j;
                const int py = r / PL_W, px = r - py * PL_W;
                const int gy = y0 - 2 + py, gx = x0 - 2 + px;
                const bool ok = r < PLANE && live && gy >= 0 && gy < H && gx >= 0 && gx < W;
                pv[ci * NPJ + j] = pb[(size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];
                pvok |= ok ? (1u << (ci * NPJ + j)) : 0u;
            }
        }
    };
    auto stage_planes = [&]() {                           // the prefetched window -> LDS: 0 outside the image / for the proxy frames, normalised inside
        const int tid_ = opaque(tid);
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            const Plane& pl = f.pl[ci];
#pragma unroll
            for (int j = 0; j < NPJ; ++j) {
                const int r = tid_ + 256 * j, k = ci * NPJ + j;
                float v = pv[k];
                if (pl.norm) v = (v / pl.div - pl.mean) / pl.stdv;
                if (r < PLANE) planes[ci * PLANE + r] = ((pvok >> k) & 1u) ? v : 0.f;
            }
        }
    };
    // FMASK: the first convolution's ReLU mask comes in its sign-bit form only (fmask_bits, one word per pixel): the 12 words of this wave's
    // three halo groups are fetched one tile AHEAD, with the input planes and before the current tile's epilogue stores -- a load issued
    // after those stores cannot complete before they have drained (vmcnt is in order and counts stores), which cost ~20 us per tile round.
    // (Holding all three groups' FLOAT masks across the previous tile's main loop spilled 51 registers and was 20 us slower.)
    constexpr int NGW = FMASK ? 3 : 1;
    uint32_t mkw[NGW][4];
    auto load_masks = [&](long tile) {
        if (!FMASK) return;
        int b, y0, x0;
        tile_coords(tile, b, y0, x0);
        const int tq = (i & 3) + 4 * h;
        const uint32_t* mbb = f.fmask_bits + (size_t)(b % f.fmask_nb) * H * W;
#pragma unroll
        for (int gi = 0; gi < NGW; ++gi)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pp = min(32 * (wave + 4 * gi) + 8 * j + tq, X3_PH * X3_PW - 1);
                const int py = pp / X3_PW, px = pp - py * X3_PW;
                const int gy = min(max(y0 - 1 + py, 0), H - 1), gx = min(max(x0 - 1 + px, 0), W - 1);
                mkw[gi][j] = mbb[(size_t)gy * W + gx];
            }
    };
    if (blockIdx.x < ntiles) { load_planes(blockIdx.x); load_masks(blockIdx.x); }
    uint4 wh[9][2];
    constexpr bool NARF = sizeof(T) == 2;            // narrow: the FIRST convolution on the bf16 matrix cores too (K = 9 cin <= 27 padded to 32: two
                                                     // v_mfma_f32_32x32x16_bf16 instead of fourteen v_mfma_f32_32x32x2_f32 per 32 halo pixels)
    float w1[NARF ? 1 : NS];
    uint4 w1b[2];                                    // NARF: the first convolution's weights as two bf16 B fragments (k = 16 ks + 8 h + e)
    uint4 w1l[W2 ? 2 : 1];                           // W2: their lo halves (w - bf16(w), rounded to bf16)
    load_weights<T, W2>(wh, wl_lds, p.wpack, p.wpack2, tid, lane);
    if constexpr (NARF) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float wv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * ks + 8 * h + e;               // fp32 fragment layout (ptta_pack_conv_in): [k >> 1][lane = (k & 1) * 32 + cout]
                wv[e] = k < K1 ? f.w1[(k >> 1) * 64 + (k & 1) * 32 + i] : 0.f;
            }
            w1b[ks] = make_uint4(pack_bf2(wv[0], wv[1]), pack_bf2(wv[2], wv[3]), pack_bf2(wv[4], wv[5]), pack_bf2(wv[6], wv[7]));
            if constexpr (W2) {
                unsigned hi_[4], lo_[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2(wv[2 * e], wv[2 * e + 1], hi_[e], lo_[e]);
                w1b[ks] = make_uint4(hi_[0], hi_[1], hi_[2], hi_[3]);
                w1l[ks] = make_uint4(lo_[0], lo_[1], lo_[2], lo_[3]);
            }
        }
    } else {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) w1[s_] = f.w1[s_ * 64 + lane];
    }
    const float b1 = f.bias1 ? f.bias1[i] : 0.f;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int b, y0, x0;
        tile_coords(tile, b, y0, x0);
        // ---- the input-plane window (prefetched) -> LDS ----
        stage_planes();
        if (tid == 0) planes[CIN * PLANE] = 0.f;                       // the padded k of an odd 9 cin reads this word
        lds_barrier();
        // ---- first convolution on the halo: pixel p = 34 py + px of the (8 + 2) x (32 + 2) halo, 32 pixels per MFMA group ----
        const bool amap = f.a_out != nullptr && b < f.a_nb;
        const bool abits = f.a_bits != nullptr && b < f.a_nb;
        auto halo_group = [&](const int g, const uint32_t* mk) {
            const int i_ = opaque(i);
            const int tq = (i_ & 3) + 4 * h, c4 = 4 * (i_ >> 2);
            const int pm = min(32 * g + i_, X3_PH * X3_PW - 1);
            const float* base = planes + (pm / X3_PW) * PL_W + (pm % X3_PW);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if constexpr (NARF) {
                // A fragment of lane (pixel i_, half h), k-step ks: input values k = 16 ks + 8 h + e -> (tap, plane) at compile-time offsets per h
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    float av[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int ka = 16 * ks + e, kb = 16 * ks + 8 + e;          // the two candidates (h = 0 / 1)
                        const int oa = ka < K1 ? (ka % CIN) * PLANE + ((ka / CIN) / 3) * PL_W + (ka / CIN) % 3 : -1;
                        const int ob = kb < K1 ? (kb % CIN) * PLANE + ((kb / CIN) / 3) * PL_W + (kb / CIN) % 3 : -1;
                        const float va = oa >= 0 ? base[oa] : 0.f, vb = ob >= 0 ? base[ob] : 0.f;
                        av[e] = h ? vb : va;
                    }
                    const uint4 af = make_uint4(pack_bf2(av[0], av[1]), pack_bf2(av[2], av[3]), pack_bf2(av[4], av[5]), pack_bf2(av[6], av[7]));
                    if constexpr (W2) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, w1l[ks]), acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, w1b[ks]), acc, 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int s_ = 0; s_ < NS; ++s_) {
                const int k0 = 2 * s_, k1 = 2 * s_ + 1;
                const int o0 = (k0 % CIN) * PLANE + ((k0 / CIN) / 3) * PL_W + (k0 / CIN) % 3;
                const int o1 = (k1 < K1) ? (k1 % CIN) * PLANE + ((k1 / CIN) / 3) * PL_W + (k1 / CIN) % 3 : -1;
                const float a = h ? (o1 >= 0 ? base[o1] : planes[CIN * PLANE]) : base[o0];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w1[s_], acc, 0, 0, 0);
            }
            }
            f32x16 t;
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = acc[r] + b1;
            quad_transpose(t, lane);                                   // lane: channels c4 ... c4 + 3 of pixel 8 j + tq of the group
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pp = 32 * g + 8 * j + tq;
                const int py = pp / X3_PW, px = pp - py * X3_PW;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                const bool inimg = pp < X3_PH * X3_PW && gy >= 0 && gy < H && gx >= 0 && gx < W;
                float4 v = make_float4(t[4 * j], t[4 * j + 1], t[4 * j + 2], t[4 * j + 3]);
                if (FMASK) {
                    const uint32_t nib = mk[j] >> c4;
                    v.x = (nib & 1u) ? v.x : 0.f; v.y = (nib & 2u) ? v.y : 0.f; v.z = (nib & 4u) ? v.z : 0.f; v.w = (nib & 8u) ? v.w : 0.f;
                }
                const bool store_a = inimg && py >= 1 && py <= X3_TH && px >= 1 && px <= 32;
                if (amap && store_a) *(float4*)(f.a_out + (((size_t)b * H + gy) * W + gx) * 32 + c4) = v;
                if (abits) {                                                        // (wave-uniform; all lanes reach the ballots)
                    const uint32_t word = mask_word_from_quads(v.x, v.y, v.z, v.w, i, h);
                    if (i < 4 && store_a) f.a_bits[((size_t)b * H + gy) * W + gx] = word;
                }
                if (RELU_MID) v = relu4(v);
                if (!inimg) v = make_float4(0.f, 0.f, 0.f, 0.f);                    // the second convolution's zero padding
                if (pp < X3_PH * X3_PW) {
                    if constexpr (sizeof(T) == 4) {
                        uint2 hi, lo;
                        split2(v.x, v.y, hi.x, lo.x); split2(v.z, v.w, hi.y, lo.y);
                        *(uint2*)(lds + pp * STR + 2 * c4) = hi;
                        *(uint2*)(lds + pp * STR + 64 + 2 * c4) = lo;
                    } else {
                        *(uint2*)(lds + pp * STR + 2 * c4) = f4_to_bf4(v);
                    }
                }
            }
        };
        constexpr int NG = (X3_PH * X3_PW + 31) / 32;
        static_assert(!FMASK || NG <= 12, "three halo groups per wave");
        if (FMASK) {
            uint32_t mk[4];
#pragma unroll 1
            for (int gi = 0; gi < NGW; ++gi) {
                if (wave + 4 * gi >= NG) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) mk[j] = gi == 0 ? mkw[0][j] : (gi == 1 ? mkw[NGW > 1 ? 1 : 0][j] : mkw[NGW > 2 ? 2 : 0][j]);
                halo_group(wave + 4 * gi, mk);
            }
        } else {
#pragma unroll 1
            for (int g = wave; g < NG; g += 4) halo_group(g, mkw[0]);
        }
        lds_barrier();                                                  // halo complete; the plane window is dead
        constexpr bool MWPRE = EMASK && !UP && sizeof(T) == 2;           // (as conv32_s1_x3_kernel: the epilogue's mask words in front of the prefetch)
        uint32_t mwp[2][4];
        if constexpr (MWPRE) {
            epi_mask_words<T>(p.epi, b, y0 + 2 * wave, H, W, i, x0, h, mwp[0]);
            epi_mask_words<T>(p.epi, b, y0 + 2 * wave + 1, H, W, i, x0, h, mwp[1]);
        }
        if (tile + gridDim.x < ntiles) { load_planes(tile + gridDim.x); load_masks(tile + gridDim.x); }
        int uy0 = 0, ux0 = 0;
        if (UP) {
            const int Hu = H >> 1, Wu = W >> 1;
            uy0 = lerp_coef(y0, Hu, sy).i0; ux0 = lerp_coef(x0, Wu, sx).i0;
            up_window_dma<T, UPH, UPW>(p.epi.up + (size_t)(b % p.epi.up_nb) * Hu * Wu * 32, Hu, Wu, uy0, ux0, up_lds, opaque(tid), wave);
        }
        // ---- second convolution, two rows per wave; the bilinear window lands during the first row's MFMAs and is awaited (by every
        // wave: the barrier sits outside the row test) before the first epilogue ----
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr, y = y0 + row;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (y < H) {                                                // wave-uniform
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap % 3;
                    const unsigned char* a = lds + ((row + ky) * X3_PW + i + kx) * STR + 16 * h;
#pragma unroll
                    for (int k = 0; k < 2; ++k) mma_step<T, W2>(acc, a + 32 * k, wh[tap][k], wl_lds + ((tap * 2 + k) * 64 + lane) * 16);
                    if (kx == 2) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (UP && rr == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lds_barrier(); }      // every wave's pieces of the window have landed
            if (y < H) {
                if (UP) epi_tile<T, false, EMASK, false, true, UPW>(p.epi, b, y, H, W, i, acc, x0, h, W, 1, 0, sy, sx, up_lds, uy0, ux0, &biasv);
                else epi_tile<T, false, EMASK, EADD>(p.epi, b, y, H, W, i, acc, x0, h, W, 1, 0, sy, sx, nullptr, 0, 0, &biasv, MWPRE ? mwp[rr] : nullptr);
            }
        }
        lds_barrier();          // LDS reuse by the next tile (plane window over the bilinear window, halo)
    }
}

// ---- the same convolution for SMALL maps (at most ~256 tiles of 8x32: the 1/4 ... 1/16-resolution layers, 18 launches of a step) ----
// Those launches are one tile per block and their duration is a latency chain, not bandwidth (in-kernel stamps of the kernel above on a
// one-tile launch: 26k cycles = weights -> LDS 5.9k + first loads 3.7k + split 2.6k + 108 MFMAs with LDS-fed lo fragments 6.1k + epilogue
// 2.6k).  This form shortens the chain: a 4x32 tile (one output row per wave: twice the blocks -- the chip is mostly idle at these sizes),
// EVERY weight fragment in registers (fp32 storage: hi and lo, 144 VGPRs: no global -> LDS -> barrier hop for the lo half), every global
// load of the block -- halo, weight fragments, bilinear window -- issued before the first wait.  Same products in the same order as
// conv32_s1_x3_kernel: bit-identical outputs.
#define X3S_TH 4
#define X3S_PH 6
template <typename T, bool RELU, bool UP, bool MASK, bool ADD, bool W2 = false>
__global__ __launch_bounds__(256, 1) void conv32_s1_small_kernel(Conv32P<T> p) {
    constexpr int STR = Geo<T>::STR;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int UPH = 4, UPW = 18;                      // a 4x32 output tile reads <= 3x17 source pixels of the half-resolution map
    __shared__ __attribute__((aligned(16))) unsigned char lds[X3S_PH * X3_PW * STR + (UP ? UPH * UPW * Geo<T>::UPPX : 0)];
    unsigned char* const up_lds = lds + X3S_PH * X3_PW * STR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const float biasv = epi_bias(p.epi, i);               // once per kernel: ptta_common.h epi_tile
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.Hout, W = p.Wout;
    const int ntx = (W + 31) >> 5, nty = (H + X3S_TH - 1) / X3S_TH;
    const int ntiles = p.B * ntx * nty;
    const float sy = up_scale(H >> 1, H), sx = up_scale(W >> 1, W);
    constexpr int NPIX = X3S_PH * X3_PW;                  // 204 halo pixels
    constexpr int NIT = (((NPIX + 7) / 8) * 32 + 255) / 256;      // (pixel, 8-channel group) items per thread
    bool wloaded = false;
    uint4 wh[9][2], wl[(F32 || W2) ? 9 : 1][2];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int ty = tile % nty, tx = (tile / nty) % ntx, b = tile / (nty * ntx);
        const int y0 = ty * X3S_TH, x0 = tx << 5;
        const T* inb = p.in + (size_t)(b % p.in_nb) * H * W * 32;
        // UNCONDITIONAL loads from clamped addresses, zero-filled at staging time: behind `if (inside) v = load` hipcc waited vmcnt(0) at the
        // join of every one of the four branches (the select with the zero needs the value) -- four L2 round trips in a row in a kernel whose
        // whole duration is a latency chain (round 5, .s)
        Px<T> v[NIT];
        bool vok[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int g = idx & 3, pix = x3_stage_pix(idx);
            const int py = pix / X3_PW, px = pix - py * X3_PW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            vok[it] = pix < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[it] = px_load(inb + ((size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * 32 + 8 * g);
        }
        if (!wloaded) {                                   // block-uniform; in flight together with the halo
            const uint4* ph = (const uint4*)p.wpack;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int k = 0; k < 2; ++k) wh[t][k] = ph[(t * 2 + k) * 64 + lane];
            if constexpr (F32 || W2) {
                const uint4* pl = (const uint4*)p.wpack2;
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int k = 0; k < 2; ++k) wl[t][k] = pl[(t * 2 + k) * 64 + lane];
            }
            wloaded = true;
        }
        int uy0 = 0, ux0 = 0;
        if (UP) {
            const int Hu = H >> 1, Wu = W >> 1;
            uy0 = lerp_coef(y0, Hu, sy).i0; ux0 = lerp_coef(x0, Wu, sx).i0;
            const T* ub = p.epi.up + (size_t)(b % p.epi.up_nb) * Hu * Wu * 32;
            constexpr int PC = Geo<T>::UPPC;
            for (int idx = tid; idx < UPH * UPW * PC; idx += 256) {
                const int q = idx % PC, pix = idx / PC;
                const int r = pix / UPW, cc = pix - r * UPW;
                const int gy = min(uy0 + r, Hu - 1), gx = min(ux0 + cc, Wu - 1);
                *(uint4*)(up_lds + (size_t)idx * 16) = *(const uint4*)(ub + ((size_t)gy * Wu + gx) * 32 + (32 / PC) * q);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int pix = x3_stage_pix(idx);
            if (!vok[it]) v[it] = px_zero<T>();
            if (pix < NPIX) px_stage<RELU>(v[it], lds + pix * STR + 16 * (idx & 3), Geo<T>::LO);
        }
        lds_barrier();
        const int y = y0 + wave;
        if (y < H) {                                      // wave-uniform
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                const unsigned char* a = lds + ((wave + ky) * X3_PW + i + kx) * STR + 16 * h;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)(a + 32 * k));
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, wh[tap][k]);
                    if constexpr (F32) {
                        const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(a + 32 * k + 64));
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, wl[tap][k]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);   // small terms first (as conv32_s1_x3_kernel)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    } else if constexpr (W2) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, wl[tap][k]), acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                }
            }
            if (UP) epi_tile<T, false, MASK, ADD, true, UPW>(p.epi, b, y, H, W, i, acc, x0, h, W, 1, 0, sy, sx, up_lds, uy0, ux0, &biasv);
            else epi_tile<T, false, MASK, ADD>(p.epi, b, y, H, W, i, acc, x0, h, W, 1, 0, sy, sx, nullptr, 0, 0, &biasv);
        }
        if (tile + (int)gridDim.x < ntiles) lds_barrier();   // LDS reuse by the next tile
    }
}

// ---- the layer loop, measured (tools/bench_chain.py, flag 16; the review's "one bounded trial") --------------------------------------
// `reps` DEPENDENT stride-1 layers on a small map in ONE launch: every block keeps its tiles, layer r reads what layer r - 1 wrote
// (ping-pong buf_a / buf_b), and between two layers the whole grid meets at a device-wide barrier: __syncthreads (every wave's stores
// acknowledged), thread 0 releases at agent scope (buffer_wbl2 sc1: the XCD's dirty L2 lines go to memory -- the 8 L2s are not coherent
// with each other), adds 1 to a counter in memory, polls it with agent-scope loads until all blocks arrived, acquires (buffer_inv sc1).
// The next layer's weight fragments are requested BEFORE the barrier (they fly during the spin), which a launch boundary cannot do.
// Every spin is bounded: after 2^16 polls (~0.1 s) the block raises *err and goes on without the other blocks (wrong values, no hang).
// The grid must be co-resident (<= 256 blocks of 256 threads, one per CU: the launcher clamps), so this form cannot share the chip with
// a second such launch.  Same products in the same order as conv32_s1_small_kernel: bit-identical outputs (tests/test_gpu_parity.py).
// variant 0: acquire loads in the spin (hipcc: global_load sc1 + buffer_inv sc1 PER POLL -- every poll of every waiting block invalidates its
// XCD's L2 under the blocks that still compute); variant 1: one release fence, a relaxed add, relaxed polls (global_load sc1 alone), one
// acquire fence behind the loop.
__device__ __forceinline__ void conv32_grid_barrier(unsigned* ctr, unsigned target, int* err, int variant) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (variant == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 16)) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        } else {
            if (variant == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // (variant 2: no fences -- timing ablation only, values undefined)
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1 << 16)) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (variant == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
}
template <typename T, bool RELU>
__global__ __launch_bounds__(256, 1) void conv32_s1_small_loop_kernel(Conv32P<T> p, T* buf_a, T* buf_b, int reps, unsigned* ctr, unsigned base, int* err, int variant) {
    constexpr int STR = Geo<T>::STR;
    constexpr bool F32 = sizeof(T) == 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[X3S_PH * X3_PW * STR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const float biasv = epi_bias(p.epi, i);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.Hout, W = p.Wout;
    const int ntx = (W + 31) >> 5, nty = (H + X3S_TH - 1) / X3S_TH;
    const int ntiles = p.B * ntx * nty;
    constexpr int NPIX = X3S_PH * X3_PW;
    constexpr int NIT = (((NPIX + 7) / 8) * 32 + 255) / 256;
    uint4 wh[9][2], wl[F32 ? 9 : 1][2];
    auto load_w = [&]() {
        const uint4* ph = (const uint4*)p.wpack;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int k = 0; k < 2; ++k) wh[t][k] = ph[(t * 2 + k) * 64 + lane];
        if constexpr (F32) {
            const uint4* pl = (const uint4*)p.wpack2;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int k = 0; k < 2; ++k) wl[t][k] = pl[(t * 2 + k) * 64 + lane];
        }
    };
    load_w();
    for (int r = 0; r < reps; ++r) {
        const T* lin = r == 0 ? p.in : ((r & 1) ? buf_a : buf_b);
        Epi<T> e = p.epi;
        e.out_raw = (r & 1) ? buf_b : buf_a;
        for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const int ty = tile % nty, tx = (tile / nty) % ntx, b = tile / (nty * ntx);
            const int y0 = ty * X3S_TH, x0 = tx << 5;
            const T* inb = lin + (size_t)(b % p.in_nb) * H * W * 32;
            Px<T> v[NIT];
            bool vok[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + 256 * it;
                const int g = idx & 3, pix = x3_stage_pix(idx);
                const int py = pix / X3_PW, px = pix - py * X3_PW;
                const int gy = y0 - 1 + py, gx = x0 - 1 + px;
                vok[it] = pix < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W;
                v[it] = px_load(inb + ((size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * 32 + 8 * g);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + 256 * it;
                const int pix = x3_stage_pix(idx);
                if (!vok[it]) v[it] = px_zero<T>();
                if (pix < NPIX) px_stage<RELU>(v[it], lds + pix * STR + 16 * (idx & 3), Geo<T>::LO);
            }
            lds_barrier();
            const int y = y0 + wave;
            if (y < H) {
                f32x16 acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap % 3;
                    const unsigned char* a = lds + ((wave + ky) * X3_PW + i + kx) * STR + 16 * h;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)(a + 32 * k));
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, wh[tap][k]);
                        if constexpr (F32) {
                            const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(a + 32 * k + 64));
                            const bf16x8 bl = __builtin_bit_cast(bf16x8, wl[tap][k]);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                        }
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                    }
                }
                epi_tile<T, false, false, false>(e, b, y, H, W, i, acc, x0, h, W, 1, 0, 1.f, 1.f, nullptr, 0, 0, &biasv);
            }
            lds_barrier();
        }
        if (r + 1 < reps) {
            load_w();          // the next layer's fragments (here: the same tensors, fetched again as a chain of different layers would)
            conv32_grid_barrier(ctr, base + (unsigned)(r + 1) * gridDim.x, err, variant);
        }
    }
}
// host side of the trial: returns the counter base the NEXT launch on the same counter must pass
int ptta_launch_conv32_loop(const Conv32Args& a, void* buf_a, void* buf_b, int reps, unsigned* ctr, unsigned* base_io, int* err, int variant, hipStream_t s) {
    if (a.bf16 || a.mode != CONV_S1 || !a.x3 || a.naive || a.up || a.mask || a.add1 || a.out_sum) return -22;
    Conv32P<float> p;
    p.in = (const float*)a.in; p.in_nb = a.in_nb;
    p.epi.bias = a.bias; p.epi.up = nullptr; p.epi.up_nb = 1; p.epi.mask = nullptr; p.epi.mask_nb = 1; p.epi.add1 = nullptr; p.epi.add1_nb = 1;
    p.epi.add2 = nullptr; p.epi.add2_nb = 1; p.epi.out_raw = nullptr; p.epi.out_sum = nullptr;
    p.B = a.B; p.Hin = p.Hout = a.Hin; p.Win = p.Wout = a.Win;
    p.wpack = a.w->mbf16; p.wpack2 = a.w->mlo;
    const long t4 = (long)p.B * ((p.Wout + 31) / 32) * ((p.Hout + X3S_TH - 1) / X3S_TH);
    const int nb = (int)(t4 > 256 ? 256 : t4);                    // co-resident: one block per CU
    if (a.relu_in) hipLaunchKernelGGL((conv32_s1_small_loop_kernel<float, true>), dim3(nb), dim3(256), 0, s, p, (float*)buf_a, (float*)buf_b, reps, ctr, *base_io, err, variant);
    else hipLaunchKernelGGL((conv32_s1_small_loop_kernel<float, false>), dim3(nb), dim3(256), 0, s, p, (float*)buf_a, (float*)buf_b, reps, ctr, *base_io, err, variant);
    *base_io += (unsigned)(reps - 1) * (unsigned)nb;
    PTTA_CHECK_LAUNCH();
    return 0;
}

// ---- stride-2 / transposed geometries, direct loads ----------------------------------------------------------------------
// Same per-wave tiling as conv32_mfma_kernel (32 outputs of one row / one x-parity), but each A fragment (8 channels of one input pixel)
// goes from L1/L2 straight into registers (fp32 storage: split into bf16 hi/lo there and fed to three bf16 MFMAs).  The weight fragments
// sit in LDS, loaded once per block -- the registers go to the activation loads that are in flight ahead of the MFMAs.  Input lines are
// re-read from L1/L2 by neighbouring taps (2.25x for stride 2), which is cheaper here than an 84 KB halo tile in LDS.  Round 2 built the
// LDS-staged, input-stationary transposed form (8x32 input tile, four parity accumulators per input row, 16 instead of 36 ds_reads):
// faster kernel by kernel (forward 35 -> 18 us at full resolution) but SLOWER in the replayed step (2.396 vs 2.384 ms, same
// box): its 79 KB of LDS cannot share a CU with the 120 KB GEMM blocks of the heads that run beside decoder 3, this
// form's 18 KB can -- reverted.
template <typename T, int MODE, bool RELU, bool UP, bool MASK, bool ADD, bool W2 = false>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 3 : 2) void conv32_direct_x3_kernel(Conv32P<T> p) {
    constexpr bool F32 = sizeof(T) == 4;
    __shared__ __attribute__((aligned(16))) unsigned char wl_lds[((F32 || W2) ? 2 : 1) * 18 * 64 * 16];     // [hi | lo][tap][k][lane] weight fragments
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const float biasv = epi_bias(p.epi, i);               // once per kernel: ptta_common.h epi_tile
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Wt = (MODE == CONV_T2) ? p.Win : p.Wout;
    const int nseg = (Wt + 31) >> 5;
    const int npar = (MODE == CONV_T2) ? 2 : 1;
    const long nitems = (long)p.B * nseg * npar * p.Hout;
    const float sy = up_scale(p.Hout >> 1, p.Hout), sx = up_scale(p.Wout >> 1, p.Wout);

    // item state (one item = 32 outputs of one row / x parity), decoded before its first loads
    int y = 0, xpar = 0, b = 0, x0 = 0;
    bool lane_in = false;
    const T* inb = p.in;
    auto decode = [&](long item) __attribute__((always_inline)) {
        long t_ = item;
        y = (int)(t_ % p.Hout); t_ /= p.Hout;
        xpar = 0;
        if (MODE == CONV_T2) { xpar = (int)(t_ & 1); t_ >>= 1; }
        const int seg = (int)(t_ % nseg);
        b = (int)(t_ / nseg);
        x0 = seg << 5;
        lane_in = (x0 + i) < Wt;
        inb = p.in + (size_t)(b % p.in_nb) * p.Hin * p.Win * 32;
    };
    // Loads are UNCONDITIONAL (clamped address + select) and issued three taps ahead of their MFMAs: with a per-load
    // `if (ok)` every one of the 36 loads of an item waited for the previous one (s_waitcnt behind each exec branch), which
    // at the 1/8 and 1/16-resolution layers -- one item per wave -- was the whole kernel: 12.3 us whatever the size.
    Px<T> v[9][2];                                   // [tap][k-step]: channels 16 k + 8 h ... + 7 of the tap's pixel
    bool act[9], okl[9];
    auto fetch = [&](int tap) __attribute__((always_inline)) {
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
        act[tap] = active;                                          // wave-uniform
        okl[tap] = active && lane_in && (xi >= 0) && (xi < p.Win);
        if (!active) return;
        const T* src = inb + ((size_t)yi * p.Win + (okl[tap] ? xi : 0)) * 32 + 8 * h;
        v[tap][0] = px_load(src); v[tap][1] = px_load(src + 16);
    };
    long item = (long)blockIdx.x * 4 + wave;
    // the first item's loads go out BEFORE the weight fragments are staged: one memory round trip for both instead of two in a row
    // (the 1/8 and 1/16-resolution launches are one item per wave: their duration is this latency chain)
    if (item < nitems) { decode(item); fetch(0); fetch(1); fetch(2); fetch(3); fetch(4); fetch(5); }
    {
        const uint4* ph = (const uint4*)p.wpack;
        for (int idx = tid; idx < 18 * 64; idx += 256) *(uint4*)(wl_lds + idx * 16) = ph[idx];
        if constexpr (F32 || W2) {
            const uint4* pl = (const uint4*)p.wpack2;
            for (int idx = tid; idx < 18 * 64; idx += 256) *(uint4*)(wl_lds + (18 * 64 + idx) * 16) = pl[idx];
        }
    }
    lds_barrier();
    while (item < nitems) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        auto compute = [&](int tap) __attribute__((always_inline)) {
            if (!act[tap]) return;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, *(const uint4*)(wl_lds + ((tap * 2 + k) * 64 + lane) * 16));
                if constexpr (F32) {
                    float4 q0 = v[tap][k].a, q1 = v[tap][k].b;
                    if (!okl[tap]) { q0 = make_float4(0.f, 0.f, 0.f, 0.f); q1 = q0; }
                    if (RELU) { q0 = relu4(q0); q1 = relu4(q1); }
                    uint4 hi, lo;
                    split2(q0.x, q0.y, hi.x, lo.x); split2(q0.z, q0.w, hi.y, lo.y);
                    split2(q1.x, q1.y, hi.z, lo.z); split2(q1.z, q1.w, hi.w, lo.w);
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, hi), al = __builtin_bit_cast(bf16x8, lo);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, *(const uint4*)(wl_lds + ((18 + tap * 2 + k) * 64 + lane) * 16));
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                } else {
                    uint4 hi = v[tap][k].a;
                    if (!okl[tap]) hi = make_uint4(0u, 0u, 0u, 0u);
                    if (RELU) { hi.x = relu_pk(hi.x); hi.y = relu_pk(hi.y); hi.z = relu_pk(hi.z); hi.w = relu_pk(hi.w); }
                    if constexpr (W2) {
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, *(const uint4*)(wl_lds + ((18 + tap * 2 + k) * 64 + lane) * 16));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hi), bl, acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, hi), bh, acc, 0, 0, 0);
                }
            }
        };
        compute(0); compute(1); compute(2);
        fetch(6); fetch(7); fetch(8);
        compute(3); compute(4); compute(5);
        compute(6); compute(7); compute(8);
        epi_tile<T, UP, MASK, ADD>(p.epi, b, y, p.Hout, p.Wout, i, acc, x0, h, Wt, MODE == CONV_T2 ? 2 : 1, xpar, sy, sx, nullptr, 0, 0, &biasv);
        item += (long)gridDim.x * 4;
        if (item < nitems) { decode(item); fetch(0); fetch(1); fetch(2); fetch(3); fetch(4); fetch(5); }
    }
}

// ---- stride-2 convolution, LDS-staged: the (2*4+1) x (2*32+1) input window of a 4 x 32 output tile is read
// ONCE as whole pixel lines (every input pixel crosses the vector-memory path 1.14 times) instead of as
// per-tap fragment loads (1.8 times, a quarter of every line touched per instruction).  Same structure as the stride-1
// kernel with two differences: the K loop is split into the two 16-channel k-steps (the window of all 32 channels would not fit
// two blocks per CU in fp32 storage), and a wave owns one output row whose A fragments are read from LDS with a two-pixel stride.
// Accumulation order: k-step outer, tap inner.  The next stage's loads are issued before the current stage's MFMAs (one register set; a
// two-set ring with loads two stages ahead needed 254-256 VGPRs, spilled in the mask / add variants and measured slower:
// 34.4 vs 33.7 us at full resolution, 9.9 vs 8.8 us at 1/8).
#define S2_TH 4
#define S2_PH (2 * S2_TH + 1)
#define S2_PW 65
#define S2_STR 80                                      // per pixel and k-step: hi 32 B | lo 32 B (fp32 storage) | pad 16 B
template <typename T, bool RELU, bool MASK, bool ADD, bool W2 = false>
__global__ __launch_bounds__(256, 2) void conv32_s2_lds_kernel(Conv32P<T> p) {
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int NPIX = S2_PH * S2_PW, NIT = (NPIX * 2 + 255) / 256;
    constexpr int WL = (F32 || W2) ? 18 * 64 * 16 : 0;
    constexpr int STR = F32 ? S2_STR : 48;                          // narrow: hi 32 B | pad 16 B per pixel and k-step (three blocks per CU)
    __shared__ __attribute__((aligned(16))) unsigned char lds[NPIX * STR + WL + 16];
    unsigned char* const wl_lds = lds + NPIX * STR;
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const float biasv = epi_bias(p.epi, i);               // once per kernel: ptta_common.h epi_tile
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hin = p.Hin, Win = p.Win, Ho = p.Hout, Wo = p.Wout;
    const int ntx = (Wo + 31) >> 5, nty = (Ho + S2_TH - 1) / S2_TH;
    const long nstages = 2L * p.B * ntx * nty;                      // (tile, k-step) pairs
    Px<T> v[NIT];
    auto coords = [&](long tile, int& b, int& oy0, int& ox0) {
        long t_ = tile;
        const int ty = (int)(t_ % nty); t_ /= nty;
        const int tx = (int)(t_ % ntx);
        b = (int)(t_ / ntx); oy0 = ty * S2_TH; ox0 = tx << 5;
    };
    auto issue_loads = [&](long stage) {
        int b, oy0, ox0;
        coords(stage >> 1, b, oy0, ox0);
        const T* inb = p.in + (size_t)(b % p.in_nb) * Hin * Win * 32 + 16 * (int)(stage & 1);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int g = idx & 1, pix = idx >> 1;
            const int py = pix / S2_PW, px = pix - py * S2_PW;
            const int gy = 2 * oy0 - 1 + py, gx = 2 * ox0 - 1 + px;
            v[it] = px_zero<T>();
            if (pix < NPIX && gy >= 0 && gy < Hin && gx >= 0 && gx < Win) v[it] = px_load(inb + ((size_t)gy * Win + gx) * 32 + 8 * g);
        }
    };
    const long first = 2L * blockIdx.x, sstride = 2L * gridDim.x;       // this block's stages: 2t, 2t+1 of its tiles
    if (first < nstages) issue_loads(first);
    uint4 wh[9][2];
    load_weights<T, W2>(wh, wl_lds, p.wpack, p.wpack2, tid, lane);
    f32x16 acc;
    bool started = false;
    for (long base = first; base < nstages; base += sstride) {
        int b, oy0, ox0;
        coords(base >> 1, b, oy0, ox0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (started) lds_barrier();                  // the previous stage's MFMAs are done with the window
            started = true;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = tid + 256 * it;
                const int pix = idx >> 1;
                if (pix < NPIX) px_stage<RELU>(v[it], lds + pix * STR + 16 * (idx & 1), 32);
            }
            {   // next stage's loads: the other k-step of this tile, or the first k-step of the block's next tile
                const long nxt = q == 0 ? base + 1 : base + sstride;
                if (nxt < nstages) issue_loads(nxt);
            }
            lds_barrier();                               // LDS-only: the loads just issued stay in flight during the MFMAs
            if (q == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
            if (oy0 + wave < Ho) {                       // wave-uniform
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap % 3;
                    const unsigned char* a = lds + ((2 * wave + ky) * S2_PW + 2 * i + kx) * STR + 16 * h;
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)a);
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, wh[tap][q]);
                    if constexpr (F32) {
                        const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(a + 32));
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, *(const uint4*)(wl_lds + ((tap * 2 + q) * 64 + lane) * 16));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);   // small terms first
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    } else if constexpr (W2) {
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, *(const uint4*)(wl_lds + ((tap * 2 + q) * 64 + lane) * 16));
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                }
                if (q == 1) epi_tile<T, false, MASK, ADD>(p.epi, b, oy0 + wave, Ho, Wo, i, acc, ox0, h, Wo, 1, 0, 0.f, 0.f, nullptr, 0, 0, &biasv);
            }
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
// in_major: src is indexed [cin_eff][cout_eff]; flip: use tap (2-ky, 2-kx).  See DESIGN.md §5 for
// which (in_major, flip) pair each forward/backward use needs.
__global__ void pack_conv32_kernel(const float* __restrict__ src, float* mf32, bf16_t* mbf16, bf16_t* mlo, float* canon,
                                   int in_major, int flip, int row_stride, int col_off) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // over tap*32*32
    if (idx >= 9 * 32 * 32) return;
    const int co = idx & 31, ci = (idx >> 5) & 31, tap = idx >> 10;
    const int st = flip ? 8 - tap : tap;
    // row_stride / col_off select a 32x32 block out of a wider weight (the 32->128->32 meta layer)
    const float v = in_major ? src[((size_t)ci * row_stride + col_off + co) * 9 + st] : src[((size_t)co * row_stride + col_off + ci) * 9 + st];
    canon[(tap * 32 + ci) * 32 + co] = v;
    {   // fp32 fragments: [tap][g][lane][s], lane = h*32 + cout, cin = 8g + 4h + s
        const int g = ci >> 3, hh = (ci >> 2) & 1, s = ci & 3;
        mf32[((tap * 4 + g) * 64 + hh * 32 + co) * 4 + s] = v;
    }
    {   // bf16 fragments: [tap][kk][lane][e], cin = 16kk + 8h + e
        const int kk = ci >> 4, hh = (ci >> 3) & 1, e = ci & 7;
        const bf16_t hi = f2bf(v);
        mbf16[((tap * 2 + kk) * 64 + hh * 32 + co) * 8 + e] = hi;
        mlo[((tap * 2 + kk) * 64 + hh * 32 + co) * 8 + e] = f2bf(v - bf2f(hi));     // bf16x3 split: v ~= hi + lo
    }
}

void ptta_pack_conv32(const float* src, const ConvW& w, int in_major, int flip, hipStream_t s, int row_stride, int col_off) {
    hipLaunchKernelGGL(pack_conv32_kernel, dim3(36), dim3(256), 0, s, src, w.mf32, w.mbf16, w.mlo, w.canon, in_major, flip, row_stride, col_off);
}

// compile-time epilogue flags -> kernel instance
template <typename T, int MODE, bool RELU>
static void launch_mfma(const Conv32P<T>& p, int flags, int blocks, hipStream_t s) {
#define K_(U, M, A) hipLaunchKernelGGL((conv32_mfma_kernel<T, MODE, RELU, U, M, A>), dim3(blocks), dim3(256), 0, s, p)
    switch (flags) {
        case 0: K_(false, false, false); break; case 1: K_(true, false, false); break;
        case 2: K_(false, true, false); break;  case 3: K_(true, true, false); break;
        case 4: K_(false, false, true); break;  case 5: K_(true, false, true); break;
        case 6: K_(false, true, true); break;   default: K_(true, true, true); break;
    }
#undef K_
}
template <typename T, bool RELU, bool W2 = false>
static void launch_x3(const Conv32P<T>& p, int flags, int blocks, hipStream_t s) {
    // small maps: the latency-chain form (conv32_s1_small_kernel), one 4x32 tile per block
    const long tiles8 = (long)p.B * ((p.Wout + 31) / 32) * ((p.Hout + X3_TH - 1) / X3_TH);
    if (tiles8 <= 256) {
        const long t4 = (long)p.B * ((p.Wout + 31) / 32) * ((p.Hout + X3S_TH - 1) / X3S_TH);
        const int nb4 = (int)(t4 > 1024 ? 1024 : t4);
#define KS_(U, M, A) hipLaunchKernelGGL((conv32_s1_small_kernel<T, RELU, U, M, A, W2>), dim3(nb4), dim3(256), 0, s, p)
        switch (flags) {
            case 0: KS_(false, false, false); break; case 1: KS_(true, false, false); break;
            case 2: KS_(false, true, false); break;  case 3: KS_(true, true, false); break;
            case 4: KS_(false, false, true); break;  case 5: KS_(true, false, true); break;
            case 6: KS_(false, true, true); break;   default: KS_(true, true, true); break;
        }
#undef KS_
        return;
    }
#define K_(U, M, A) hipLaunchKernelGGL((conv32_s1_x3_kernel<T, RELU, U, M, A, W2>), dim3(blocks), dim3(256), 0, s, p)
#ifdef PTTA_DIAG_STAMPS      // diagnostic build (make DIAG=1): in-kernel phase stamps of the plain launches (tools/bench_chain.py prints them)
    if (flags == 0 && !W2 && RELU) { hipLaunchKernelGGL((conv32_s1_x3_kernel<T, RELU, false, false, false, false, true>), dim3(blocks), dim3(256), 0, s, p); return; }
#endif
    switch (flags) {
        case 0: K_(false, false, false); break; case 1: K_(true, false, false); break;
        case 2: K_(false, true, false); break;  case 3: K_(true, true, false); break;
        case 4: K_(false, false, true); break;  case 5: K_(true, false, true); break;
        case 6: K_(false, true, true); break;   default: K_(true, true, true); break;
    }
#undef K_
}

template <typename T, int MODE, bool RELU, bool W2 = false>
static void launch_direct_x3(const Conv32P<T>& p, int flags, int blocks, hipStream_t s) {
#define K_(U, M, A) hipLaunchKernelGGL((conv32_direct_x3_kernel<T, MODE, RELU, U, M, A, W2>), dim3(blocks), dim3(256), 0, s, p)
    switch (flags) {
        case 0: K_(false, false, false); break; case 1: K_(true, false, false); break;
        case 2: K_(false, true, false); break;  case 3: K_(true, true, false); break;
        case 4: K_(false, false, true); break;  case 5: K_(true, false, true); break;
        case 6: K_(false, true, true); break;   default: K_(true, true, true); break;
    }
#undef K_
}

// Workgroups of a persistent full-chip launch.  The stride-1 kernel takes 480 of the 512 slots (two 256-VGPR blocks per CU): a launch that
// owns every register of the chip keeps every kernel of the other queues (the tiny BatchNorm finalizes between the heads' GEMMs, the next
// frame's prefix) waiting until it retires, and 32 CUs with a single block break the lock-step of the rest.  Round 4, same box, graph
// replay, pipelined / call by call: 512 1.698 / 1.819, 496 1.699, 480 1.681 / 1.803, 464 1.698, 448 1.73, 416 1.81; the first-layer /
// strided / transposed kernels stay at 512 (480 there: 1.689 vs 1.688).
static constexpr int kFullChipBlocks = 512, kS1Blocks = 480;
// narrow kernels (half the LDS tile, ~160 VGPRs): three resident blocks per CU
static constexpr int kNarrowBlocks = 768;         // (the direct kernel only: <= 160 VGPRs, 18 KB LDS)

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
    p.epi.mask_bits = a.mask_bits; p.epi.bits_out = a.bits_out; p.epi.bits_nb = a.bits_nb; p.epi.bits_sum = a.bits_sum;
    p.B = a.B; p.Hin = a.Hin; p.Win = a.Win;
    if (MODE == CONV_S1) { p.Hout = a.Hin; p.Wout = a.Win; }
    else if (MODE == CONV_S2) { p.Hout = a.Hin / 2; p.Wout = a.Win / 2; }
    else { p.Hout = a.Hin * 2; p.Wout = a.Win * 2; }
    if (a.add2 && !a.add1) return -22;
    const int flags = (a.up ? 1 : 0) | (a.mask ? 2 : 0) | (a.add1 ? 4 : 0);
    p.wpack2 = nullptr;
    // the tuned kernels: fp32 storage with bf16x3 arithmetic, or narrow storage (always one bf16 MFMA per product); a narrow launch with a
    // ReLU mask takes it as sign bits (the mask map itself is an fp32 map of the real frames)
    constexpr bool NAR = sizeof(T) == 2;
    if (NAR && a.mask && !a.mask_bits) return -22;
    if (!a.naive && (NAR || a.x3)) {
        p.wpack = a.w->mbf16; p.wpack2 = (NAR && !a.w2) ? nullptr : a.w->mlo;
        // narrow storage with hi + lo weights (Conv32Args::w2): the data gradients of the mixed mode -- no bilinear epilogue there
        if constexpr (NAR) {
            if (a.w2) {
                if ((flags & 1) || a.relu_in) return -22;          // (data gradients: no ReLU on load, no bilinear skip -- the other forms are not instantiated)
                if (MODE == CONV_S1) {
                    const long tiles = (long)p.B * ((p.Wout + 31) / 32) * ((p.Hout + X3_TH - 1) / X3_TH);
                    const int blocks = (int)(tiles > kS1Blocks ? kS1Blocks : tiles);
                    launch_x3<T, false, true>(p, flags, blocks, s);
                    PTTA_CHECK_LAUNCH();
                    return 0;
                }
                const int Wt = (MODE == CONV_T2) ? p.Win : p.Wout;
                const long items = (long)p.B * ((Wt + 31) / 32) * (MODE == CONV_T2 ? 2 : 1) * p.Hout;
                long blocks = (items + 3) / 4; if (blocks > kNarrowBlocks) blocks = kNarrowBlocks;
                if (MODE == CONV_S2) {
                    const long tiles = (long)p.B * ((p.Wout + 31) / 32) * ((p.Hout + S2_TH - 1) / S2_TH);
                    const int tb = (int)(tiles < kFullChipBlocks ? tiles : kFullChipBlocks);
#define KS2W_(R, M, A) hipLaunchKernelGGL((conv32_s2_lds_kernel<T, R, M, A, true>), dim3(tb), dim3(256), 0, s, p)
                    const bool m_ = flags & 2, a_ = flags & 4;
                    { if (m_) { if (a_) KS2W_(false, true, true); else KS2W_(false, true, false); } else { if (a_) KS2W_(false, false, true); else KS2W_(false, false, false); } }
#undef KS2W_
                    PTTA_CHECK_LAUNCH();
                    return 0;
                }
                if constexpr (MODE != CONV_S1) {
                    launch_direct_x3<T, MODE, false, true>(p, flags, (int)blocks, s);
                }
                PTTA_CHECK_LAUNCH();
                return 0;
            }
        }
        if (MODE == CONV_S1) {
            const long tiles = (long)p.B * ((p.Wout + 31) / 32) * ((p.Hout + X3_TH - 1) / X3_TH);
            const int blocks = (int)(tiles > kS1Blocks ? kS1Blocks : tiles);     // 2 resident blocks per CU, persistent
            if (a.relu_in) launch_x3<T, true>(p, flags, blocks, s); else launch_x3<T, false>(p, flags, blocks, s);
            PTTA_CHECK_LAUNCH();
            return 0;
        }
        const int Wt = (MODE == CONV_T2) ? p.Win : p.Wout;
        const long items = (long)p.B * ((Wt + 31) / 32) * (MODE == CONV_T2 ? 2 : 1) * p.Hout;
        const int capd = NAR ? kNarrowBlocks : kFullChipBlocks;
        long blocks = (items + 3) / 4; if (blocks > capd) blocks = capd;
        if (MODE == CONV_S2 && !(flags & 1)) {
            const long tiles = (long)p.B * ((p.Wout + 31) / 32) * ((p.Hout + S2_TH - 1) / S2_TH);
            const int tb = (int)(tiles < kFullChipBlocks ? tiles : kFullChipBlocks);
#define KS2_(R, M, A) hipLaunchKernelGGL((conv32_s2_lds_kernel<T, R, M, A>), dim3(tb), dim3(256), 0, s, p)
            const bool m_ = flags & 2, a_ = flags & 4;
            if (a.relu_in) { if (m_) { if (a_) KS2_(true, true, true); else KS2_(true, true, false); } else { if (a_) KS2_(true, false, true); else KS2_(true, false, false); } }
            else { if (m_) { if (a_) KS2_(false, true, true); else KS2_(false, true, false); } else { if (a_) KS2_(false, false, true); else KS2_(false, false, false); } }
#undef KS2_
            PTTA_CHECK_LAUNCH();
            return 0;
        }
        if constexpr (MODE != CONV_S1) {
            if (a.relu_in) launch_direct_x3<T, MODE, true>(p, flags, (int)blocks, s); else launch_direct_x3<T, MODE, false>(p, flags, (int)blocks, s);
        }
        PTTA_CHECK_LAUNCH();
        return 0;
    }
    if (a.naive) {
        p.wpack = a.w->canon;
        const long total = (long)p.B * p.Hout * p.Wout * 32;
        int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
        if (a.relu_in) hipLaunchKernelGGL((conv32_naive_kernel<T, MODE, true>), dim3(blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv32_naive_kernel<T, MODE, false>), dim3(blocks), dim3(256), 0, s, p);
    } else {
        // exact arithmetic (validation): v_mfma_f32_32x32x2_f32, fp32 storage only
        if (NAR) return -22;
        p.wpack = (const void*)a.w->mf32;
        const int Wt = (MODE == CONV_T2) ? p.Win : p.Wout;
        const long items = (long)p.B * ((Wt + 31) / 32) * (MODE == CONV_T2 ? 2 : 1) * p.Hout;
        long blocks = (items + 3) / 4;
        if (blocks > 512) blocks = 512;
        if constexpr (!NAR) { if (a.relu_in) launch_mfma<T, MODE, true>(p, flags, (int)blocks, s); else launch_mfma<T, MODE, false>(p, flags, (int)blocks, s); }
    }
    PTTA_CHECK_LAUNCH();
    return 0;
}

// Conv2d(cin, 32) - ReLU - Conv2d(32, 32) in one launch (conv32_s1_first_kernel): `a` describes the SECOND convolution as for
// ptta_launch_conv32 (its `in` is ignored; a.bf16: narrow output and skip operands, one MFMA per product), `f` the first one as for
// ptta_launch_conv_in (its outputs are ignored); a_out: where the first convolution's pre-activation map is still written (frames
// b < a_nb), or null.  Returns 1 when this form does not apply (the caller then launches the two kernels): exact / naive arithmetic,
// epilogues other than the bilinear skip, small maps.
template <typename T>
static int launch_first_t(const Conv32Args& a, const ConvInArgs& f, void* a_out, int a_nb, bool bwd_form, long tiles, hipStream_t s) {
    Conv32P<T> p;
    p.in = nullptr; p.in_nb = 1; p.wpack = a.w->mbf16; p.wpack2 = (sizeof(T) == 4 || a.w2) ? a.w->mlo : nullptr;
    p.epi.bias = a.bias; p.epi.up = (const T*)a.up; p.epi.up_nb = a.up_nb > 0 ? a.up_nb : 1;
    p.epi.mask = (const T*)a.mask; p.epi.mask_nb = a.mask_nb > 0 ? a.mask_nb : 1; p.epi.add1 = (const T*)a.add1; p.epi.add1_nb = a.add1_nb > 0 ? a.add1_nb : 1; p.epi.add2 = nullptr; p.epi.add2_nb = 1;
    p.epi.out_raw = (T*)a.out_raw; p.epi.out_sum = (T*)a.out_sum;          // (add1 / out_sum: the backward form only, see ptta_launch_conv32_first)
    p.epi.mask_bits = a.mask_bits; p.epi.bits_out = a.bits_out; p.epi.bits_nb = a.bits_nb; p.epi.bits_sum = 0;
    p.B = a.B; p.Hin = p.Hout = a.Hin; p.Win = p.Wout = a.Win;
    const int blocks = (int)(tiles > kFullChipBlocks ? kFullChipBlocks : tiles);
    FirstP q;
    for (int c = 0; c < 3; ++c) { q.pl[c] = f.pl[c]; if (q.pl[c].nb < 1) q.pl[c].nb = 1; }
    q.zero_from_b = f.zero_from_b; q.w1 = f.wfrag; q.bias1 = f.bias; q.a_out = (float*)a_out; q.a_nb = a_nb;
    q.fmask_nb = f.mask_nb > 0 ? f.mask_nb : 1; q.fmask_bits = f.mask_bits; q.a_bits = f.a_bits;
    if (bwd_form) {
        if constexpr (sizeof(T) == 2) {
            if (a.w2) {
                if (a.out_sum) hipLaunchKernelGGL((conv32_s1_first_kernel<T, 1, false, true, false, true, true, true>), dim3(blocks), dim3(256), 0, s, p, q);
                else hipLaunchKernelGGL((conv32_s1_first_kernel<T, 1, false, true, false, true, false, true>), dim3(blocks), dim3(256), 0, s, p, q);
                PTTA_CHECK_LAUNCH();
                return 0;
            }
        }
        if (a.out_sum) hipLaunchKernelGGL((conv32_s1_first_kernel<T, 1, false, true, false, true, true>), dim3(blocks), dim3(256), 0, s, p, q);
        else hipLaunchKernelGGL((conv32_s1_first_kernel<T, 1, false, true, false, true>), dim3(blocks), dim3(256), 0, s, p, q);
        PTTA_CHECK_LAUNCH();
        return 0;
    }
#define KF_(C) do { if (a.up) hipLaunchKernelGGL((conv32_s1_first_kernel<T, C, true>), dim3(blocks), dim3(256), 0, s, p, q); \
                    else hipLaunchKernelGGL((conv32_s1_first_kernel<T, C, false>), dim3(blocks), dim3(256), 0, s, p, q); } while (0)
    if (f.cin == 1) KF_(1); else if (f.cin == 2) KF_(2); else KF_(3);
#undef KF_
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_launch_conv32_first(const Conv32Args& a, const ConvInArgs& f, void* a_out, int a_nb, hipStream_t s) {
    // two shapes: forward (ReLU between the convolutions, optional bilinear skip, no masks) and the prediction head's backward (cin = 1, the
    // first convolution masked, no ReLU in between, the second convolution's epilogue masked, no bilinear skip)
    const bool bwd_form = f.mask != nullptr;
    if (bwd_form && (!f.mask_bits || !a.mask_bits)) return 1;
    if (a.naive || (!a.bf16 && !a.x3) || a.mode != CONV_S1 || a.add2 || f.naive) return 1;
    // (a second output = masked result + one addend: the backward form only -- d z4 = d s0_2 + up2^T(d e3_0), ptta_api.hip backbone_backward)
    if ((a.add1 != nullptr) != (a.out_sum != nullptr) || (a.out_sum && !bwd_form)) return 1;
    if (bwd_form ? (a.relu_in || !a.mask || a.up || f.cin != 1) : (!a.relu_in || a.mask != nullptr)) return 1;
    if (f.cin < 1 || f.cin > 3 || f.up || f.add1 || f.B != a.B || f.H != a.Hin || f.W != a.Win) return 1;
    if (a.bf16 && (a_out || f.a_bits)) return 1;             // (a narrow launch has no backward of its own: nothing to keep of the first map)
    const long tiles = (long)a.B * ((a.Win + 31) / 32) * ((a.Hin + X3_TH - 1) / X3_TH);
    if (tiles <= 256) return 1;
    return a.bf16 ? launch_first_t<bf16_t>(a, f, a_out, a_nb, bwd_form, tiles, s) : launch_first_t<float>(a, f, a_out, a_nb, bwd_form, tiles, s);
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
