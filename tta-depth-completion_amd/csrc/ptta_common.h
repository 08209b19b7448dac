// Shared device/host definitions for libptta_hip (gfx950 only).
//
// Data layout in HBM:
//   * 32-channel feature maps: NHWC, element type T = float (fp32 mode) or bf16 (bf16 mode);
//     one pixel = one 128-B (fp32) / 64-B (bf16) line, so every access to a pixel is a full line.
//   * 1- and 3-channel maps (image, sparse depth, predictions, depth gradients): planar NCHW fp32,
//     exactly the layout the reference hands over, so the C-ABI needs no transposes.
//   * head activations: row-major [rows][512], rows in NHWC pixel order (= the reference's
//     permute(0,2,3,1).reshape(-1,C), network_exp_msg_chn_adapt.py:551-554).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;   // raw bfloat16 bits
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define PTTA_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
    // round-to-nearest-even; NaN kept a NaN (MI355X_MICROARCH.md "Correctness boundaries")
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    return (bf16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// four bf16 <-> four floats (8-byte pieces of a narrow NHWC pixel); v_cvt_pk_bf16_f32 rounds to nearest even and keeps a NaN a NaN
typedef __bf16 cbf16x2 __attribute__((ext_vector_type(2)));
typedef float cfloat2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
    cfloat2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, cbf16x2));
}
__device__ __forceinline__ float4 bf4_to_f4(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 f4_to_bf4(float4 v) { return make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w)); }

// ---- bilinear x2, align_corners=True (F.interpolate, network_exp_msg_chn_adapt.py:200-209) ----
// PyTorch's upsample_bilinear2d: src = dst * (in-1)/(out-1); i0 = (int)src; i1 = i0 + (i0 < in-1);
// l1 = src - i0; l0 = 1 - l1.
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_coef(int dst, int in_size, float scale) {
    float s = scale * (float)dst;
    int i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    Lerp r;
    r.i0 = i0;
    r.i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
    r.l1 = s - (float)i0;
    r.l0 = 1.0f - r.l1;
    return r;
}
__host__ __device__ __forceinline__ float up_scale(int in_size, int out_size) {
    return out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.0f;
}

// ---- universal producer epilogue -------------------------------------------------------------
// v = acc (+bias[ch]) (+ bilinear_up2(up)[b,y,x,ch]) ; v *= (mask[b,y,x,ch] > 0)
// out_raw[b,y,x,ch] = v ; out_sum[b,y,x,ch] = v + add1 + add2
// Every auxiliary tensor has its own batch count nb and is indexed with b % nb, which lets the
// proxy (zero-image) pass share the depth-only tensors of the grad pass.
template <typename T>
struct Epi {
    const float* bias;
    const T* up;   int up_nb;
    const T* mask; int mask_nb;
    const T* add1; int add1_nb;
    const T* add2; int add2_nb;
    T* out_raw;
    T* out_sum;
    // Sign-bit form of the ReLU masks (fp32 storage, MFMA kernels): one 32-bit word per pixel, bit c = (value of channel c > 0).
    // A forward epilogue whose output the backward uses as a mask also writes `bits_out` for frames b < bits_nb (of out_sum when
    // bits_sum, else of out_raw); a backward epilogue then reads `mask_bits` (4 B per pixel) instead of the 128-B fp32 pixel of `mask`.
    // Same predicate (> 0) on the same stored values: results are bit-identical to the float-mask form.
    const uint32_t* mask_bits = nullptr;
    uint32_t* bits_out = nullptr; int bits_nb = 0; int bits_sum = 0;
};

// One mask word from the transposed epilogue layout: lane (ch, h) holds channels 4 (ch >> 2) ... + 3 of one pixel and the eight lanes
// ch = a, a + 4, ..., a + 28 (same h) hold that pixel's 32 channels.  Four ballots (one per channel of the lane's quad); the lane with
// ch = a < 4 assembles the pixel's word: bit 4 k + q = ballot_q bit (32 h + a + 4 k).  Must be called with all 64 lanes active.
__device__ __forceinline__ uint32_t mask_word_from_quads(float x, float y, float z, float w, int ch, int h) {
    const unsigned long long b0 = __builtin_amdgcn_ballot_w64(x > 0.0f), b1 = __builtin_amdgcn_ballot_w64(y > 0.0f);
    const unsigned long long b2 = __builtin_amdgcn_ballot_w64(z > 0.0f), b3 = __builtin_amdgcn_ballot_w64(w > 0.0f);
    const int sh = 32 * h + (ch & 3);
    return ((uint32_t)(b0 >> sh) & 0x11111111u) | (((uint32_t)(b1 >> sh) & 0x11111111u) << 1) |
           (((uint32_t)(b2 >> sh) & 0x11111111u) << 2) | (((uint32_t)(b3 >> sh) & 0x11111111u) << 3);
}

template <typename T>
__device__ __forceinline__ void epi_store(const Epi<T>& e, int b, int y, int x, int H, int W, int ch,
                                          float v, const Lerp& ly, const Lerp& lx) {
    if (e.bias) v += e.bias[ch];
    if (e.up) {
        const int Hu = H >> 1, Wu = W >> 1;
        const T* u = e.up + (size_t)(b % e.up_nb) * Hu * Wu * 32 + ch;
        float v00 = ld(u + ((size_t)ly.i0 * Wu + lx.i0) * 32), v01 = ld(u + ((size_t)ly.i0 * Wu + lx.i1) * 32);
        float v10 = ld(u + ((size_t)ly.i1 * Wu + lx.i0) * 32), v11 = ld(u + ((size_t)ly.i1 * Wu + lx.i1) * 32);
        v += ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
    }
    const size_t pix = (size_t)y * W + x;
    if (e.mask) {
        float m = ld(e.mask + ((size_t)(b % e.mask_nb) * H * W + pix) * 32 + ch);
        v = m > 0.0f ? v : 0.0f;
    }
    const size_t o = ((size_t)b * H * W + pix) * 32 + ch;
    if (e.out_raw) st(e.out_raw + o, v);
    if (e.out_sum) {
        if (e.add1) v += ld(e.add1 + ((size_t)(b % e.add1_nb) * H * W + pix) * 32 + ch);
        if (e.add2) v += ld(e.add2 + ((size_t)(b % e.add2_nb) * H * W + pix) * 32 + ch);
        st(e.out_sum + o, v);
    }
}

__device__ const float kZeroBias[32] = {0.f};

// 4x4 transposes inside the lane quads of a 32x32 MFMA accumulator (DPP quad_perm, no LDS): on entry lane (i, h) holds in t[4g + a]
// row (a + 8g + 4h), column i; on return it holds in t[4g + c] row ((i & 3) + 8g + 4h), column 4 (i >> 2) + c -- four consecutive
// columns per lane, one float4 per g, and the 64 lanes of one g cover eight consecutive rows x 32 columns (1 KB when the row is
// a 32-channel fp32 pixel).  The accumulator layout itself would store one dword per lane: four times the vector-memory
// instructions, which is what the conv / GEMM epilogues are bound by.
__device__ __forceinline__ float quad_xchg(float v, bool xor2) {
    const int s = __float_as_int(v);
    return __int_as_float(xor2 ? __builtin_amdgcn_update_dpp(s, s, 0x4E, 0xF, 0xF, false)      // quad_perm [2,3,0,1]
                               : __builtin_amdgcn_update_dpp(s, s, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
}
__device__ __forceinline__ void quad_transpose(f32x16& t, int lane) {
    const bool odd = lane & 1, up = lane & 2;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float x0 = t[4 * g], x1 = t[4 * g + 1], x2 = t[4 * g + 2], x3 = t[4 * g + 3];
        float r;
        r = quad_xchg(odd ? x0 : x1, false); x0 = odd ? r : x0; x1 = odd ? x1 : r;
        r = quad_xchg(odd ? x2 : x3, false); x2 = odd ? r : x2; x3 = odd ? x3 : r;
        r = quad_xchg(up ? x0 : x2, true); x0 = up ? r : x0; x2 = up ? x2 : r;
        r = quad_xchg(up ? x1 : x3, true); x1 = up ? r : x1; x3 = up ? x3 : r;
        t[4 * g] = x0; t[4 * g + 1] = x1; t[4 * g + 2] = x2; t[4 * g + 3] = x3;
    }
}

// Tile form of the epilogue for the MFMA kernels: lane = channel `ch`, the 16 accumulator registers
// are 16 pixels of one output row.  UP / MASK / ADD are compile-time so that every auxiliary load
// of the tile is unconditional and can be issued back to back (a runtime "pointer or nothing" test
// per element makes hipcc branch around each load and wait vmcnt(0) 100+ times per tile:
// cdna_hip_programming.md §5 "Three .s-level traps" (c)).  Out-of-range pixels load from a
// clamped address and are simply not stored.
// UPL (fp32 storage only): the bilinear x2 skip comes from an LDS window `upw` = [row][UPLW columns][32 floats] of the half-resolution
// source whose origin is (uy0, ux0), and is added AFTER the quad transpose: a lane then holds four consecutive channels of one pixel,
// so one row costs it 4 coefficient pairs and 16 ds_read_b128 instead of 16 pairs and 64 ds_read_b32 -- the same expression per
// element as the accumulator-layout form, (conv + bias) + (ly.l0 (lx.l0 v00 + lx.l1 v01) + ly.l1 (lx.l0 v10 + lx.l1 v11)).
template <typename T>
__device__ __forceinline__ float epi_bias(const Epi<T>& e, int ch) { return (e.bias ? e.bias : kZeroBias)[ch]; }
// The four mask words epi_tile's narrow branch reads for output row y (stride-1 geometry: xstep 1, xoff 0), loaded AHEAD by the caller and
// passed back as `mw_pre`: vmcnt is in order, so a load issued in the epilogue -- behind the next tile's halo prefetch -- cannot be waited
// for without waiting for that prefetch too; issued in front of it, the wait leaves the prefetch in flight (round 5, .s: every row's epilogue
// of the masked kernels stalled until the next tile had landed).
template <typename T>
__device__ __forceinline__ void epi_mask_words(const Epi<T>& e, int b, int y, int H, int W, int ch, int x0, int h, uint32_t (&mw)[4]) {
    const uint32_t* mb = e.mask_bits + ((size_t)(b % e.mask_nb) * H + min(y, H - 1)) * W;
#pragma unroll
    for (int g = 0; g < 4; ++g) mw[g] = mb[min(x0 + (ch & 3) + 8 * g + 4 * h, W - 1)];
}
template <typename T, bool UP, bool MASK, bool ADD, bool UPL = false, int UPLW = 18>
__device__ __forceinline__ void epi_tile(const Epi<T>& e, int b, int y, int H, int W, int ch, const f32x16& acc,
                                         int x0, int h, int Wt, int xstep, int xoff, float sy, float sx,
                                         const void* upw_ = nullptr, int uy0 = 0, int ux0 = 0, const float* bias_lane = nullptr,
                                         const uint32_t* mw_pre = nullptr) {
    const float* upw = (const float*)upw_;                // fp32 storage: the bilinear window holds floats; narrow storage casts upw_ to bf16 below
    const float* bp = e.bias ? e.bias : kZeroBias;       // pointer select, not a branch around the load
    // bias_lane: the lane's bias, loaded ONCE by the kernel (epi_bias).  Loaded here -- once per output row -- it cannot be hoisted by hipcc
    // (the epilogue's stores may alias it), and the wait for this youngest load is s_waitcnt vmcnt(0): every row then drained the next
    // tile's halo prefetch and the previous row's stores before its own epilogue began (round 5, .s of the stride-1 kernel)
    const float bias = bias_lane ? *bias_lane : bp[ch];
    float v[16];
    int xo[16];
    bool ok[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int px = x0 + ((r & 3) + 8 * (r >> 2) + 4 * h);
        ok[r] = px < Wt;
        xo[r] = (ok[r] ? px : Wt - 1) * xstep + xoff;
        v[r] = acc[r] + bias;
    }
    if (UP) {
        const int Hu = H >> 1, Wu = W >> 1;
        const Lerp ly = lerp_coef(y, Hu, sy);
        const T* u0 = e.up + ((size_t)(b % e.up_nb) * Hu + ly.i0) * Wu * 32 + ch;
        const T* u1 = e.up + ((size_t)(b % e.up_nb) * Hu + ly.i1) * Wu * 32 + ch;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const Lerp lx = lerp_coef(xo[r], Wu, sx);
            const float v00 = ld(u0 + (size_t)lx.i0 * 32), v01 = ld(u0 + (size_t)lx.i1 * 32);
            const float v10 = ld(u1 + (size_t)lx.i0 * 32), v11 = ld(u1 + (size_t)lx.i1 * 32);
            v[r] += ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // 16 gathers in flight at a time (VGPR budget)
        }
    }
    const size_t rowoff = (size_t)y * W;
    if constexpr (sizeof(T) == 4) {
        // fp32 storage: mask / store / add / store as float4 in the transposed layout (same arithmetic per element, in the same order)
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = v[r];
        quad_transpose(t, ch + 32 * h);
        const int j4 = (ch >> 2) * 4;
        bool okg[4];
        size_t xq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int px = x0 + (ch & 3) + 8 * g + 4 * h;
            okg[g] = px < Wt;
            xq[g] = ((size_t)((okg[g] ? px : Wt - 1) * xstep + xoff) + rowoff) * 32 + j4;
        }
        if constexpr (UPL) {
            const Lerp ly = lerp_coef(y, H >> 1, sy);
            const float* r0 = upw + (ly.i0 - uy0) * UPLW * 32 + j4;
            const float* r1 = upw + (ly.i1 - uy0) * UPLW * 32 + j4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int px = min(x0 + (ch & 3) + 8 * g + 4 * h, W - 1);
                const Lerp lx = lerp_coef(px, W >> 1, sx);
                const int c0 = (lx.i0 - ux0) * 32, c1 = (lx.i1 - ux0) * 32;
                const float4 v00 = *(const float4*)(r0 + c0), v01 = *(const float4*)(r0 + c1);
                const float4 v10 = *(const float4*)(r1 + c0), v11 = *(const float4*)(r1 + c1);
                t[4 * g] = t[4 * g] + (ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x));
                t[4 * g + 1] = t[4 * g + 1] + (ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y));
                t[4 * g + 2] = t[4 * g + 2] + (ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z));
                t[4 * g + 3] = t[4 * g + 3] + (ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w));
                __builtin_amdgcn_sched_barrier(0);                   // four ds_read_b128 in flight at a time (VGPR budget)
            }
        }
        // every auxiliary load of the row is issued before the first store (a store between them would order them)
        float4 mk[4], t1[4], t2[4];
        uint32_t mw[4];
        const bool mbits = MASK && e.mask_bits != nullptr;
        if (MASK) {
            if (mbits) {
                const uint32_t* mb = e.mask_bits + (size_t)(b % e.mask_nb) * H * W;
#pragma unroll
                for (int g = 0; g < 4; ++g) { mw[g] = mb[xq[g] >> 5]; mk[g] = make_float4(0.f, 0.f, 0.f, 0.f); }
            } else {
                const float* mb = (const float*)e.mask + (size_t)(b % e.mask_nb) * H * W * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) { mk[g] = *(const float4*)(mb + xq[g]); mw[g] = 0u; }
            }
        }
        if (ADD) {
            const float* a1 = (const float*)e.add1 + (size_t)(b % e.add1_nb) * H * W * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) t1[g] = *(const float4*)(a1 + xq[g]);
            if (e.add2) {
                const float* a2 = (const float*)e.add2 + (size_t)(b % e.add2_nb) * H * W * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) t2[g] = *(const float4*)(a2 + xq[g]);
#pragma unroll
                for (int g = 0; g < 4; ++g) { t1[g].x += t2[g].x; t1[g].y += t2[g].y; t1[g].z += t2[g].z; t1[g].w += t2[g].w; }
            }
        }
        float* const oraw = (float*)e.out_raw + (size_t)b * H * W * 32;
        float* const osum = (float*)e.out_sum + (size_t)b * H * W * 32;
        const bool wbits = e.bits_out != nullptr && b < e.bits_nb;              // wave-uniform
        uint32_t* const obits = e.bits_out + (size_t)b * H * W;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 val = make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
            if (MASK) {
                if (mbits) {
                    const uint32_t nib = mw[g] >> j4;
                    val.x = (nib & 1u) ? val.x : 0.0f; val.y = (nib & 2u) ? val.y : 0.0f;
                    val.z = (nib & 4u) ? val.z : 0.0f; val.w = (nib & 8u) ? val.w : 0.0f;
                } else {
                    val.x = mk[g].x > 0.0f ? val.x : 0.0f; val.y = mk[g].y > 0.0f ? val.y : 0.0f;
                    val.z = mk[g].z > 0.0f ? val.z : 0.0f; val.w = mk[g].w > 0.0f ? val.w : 0.0f;
                }
            }
            if (e.out_raw && okg[g]) *(float4*)(oraw + xq[g]) = val;
            const float4 vraw = val;
            if (ADD) { val.x += t1[g].x; val.y += t1[g].y; val.z += t1[g].z; val.w += t1[g].w; }
            if (e.out_sum && okg[g]) *(float4*)(osum + xq[g]) = val;
            if (wbits) {
                const float4 bs = e.bits_sum ? val : vraw;
                const uint32_t word = mask_word_from_quads(bs.x, bs.y, bs.z, bs.w, ch, h);
                if (ch < 4 && okg[g]) obits[xq[g] >> 5] = word;
            }
        }
        return;
    }
    if constexpr (sizeof(T) == 2) {
        // narrow (bf16) storage, the mixed mode's proxy-pass and gradient maps: the same transposed layout, 8 B per lane (four consecutive
        // channels of one pixel), eight pixels x 64 contiguous bytes per instruction.  ReLU masks come as sign-bit words only (they are
        // planes of the REAL frames' fp32 maps: the launchers refuse a narrow launch whose mask has no bit plane); skip additions are narrow
        // maps; arithmetic in fp32, one rounding per stored value.
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = v[r];
        quad_transpose(t, ch + 32 * h);
        const int j4 = (ch >> 2) * 4;
        bool okg[4];
        size_t xq[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int px = x0 + (ch & 3) + 8 * g + 4 * h;
            okg[g] = px < Wt;
            xq[g] = ((size_t)((okg[g] ? px : Wt - 1) * xstep + xoff) + rowoff) * 32 + j4;
        }
        if constexpr (UPL) {
            const bf16_t* upb = (const bf16_t*)upw_;
            const Lerp ly = lerp_coef(y, H >> 1, sy);
            const bf16_t* r0 = upb + (ly.i0 - uy0) * UPLW * 32 + j4;
            const bf16_t* r1 = upb + (ly.i1 - uy0) * UPLW * 32 + j4;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int px = min(x0 + (ch & 3) + 8 * g + 4 * h, W - 1);
                const Lerp lx = lerp_coef(px, W >> 1, sx);
                const int c0 = (lx.i0 - ux0) * 32, c1 = (lx.i1 - ux0) * 32;
                const float4 v00 = bf4_to_f4(*(const uint2*)(r0 + c0)), v01 = bf4_to_f4(*(const uint2*)(r0 + c1));
                const float4 v10 = bf4_to_f4(*(const uint2*)(r1 + c0)), v11 = bf4_to_f4(*(const uint2*)(r1 + c1));
                t[4 * g] = t[4 * g] + (ly.l0 * (lx.l0 * v00.x + lx.l1 * v01.x) + ly.l1 * (lx.l0 * v10.x + lx.l1 * v11.x));
                t[4 * g + 1] = t[4 * g + 1] + (ly.l0 * (lx.l0 * v00.y + lx.l1 * v01.y) + ly.l1 * (lx.l0 * v10.y + lx.l1 * v11.y));
                t[4 * g + 2] = t[4 * g + 2] + (ly.l0 * (lx.l0 * v00.z + lx.l1 * v01.z) + ly.l1 * (lx.l0 * v10.z + lx.l1 * v11.z));
                t[4 * g + 3] = t[4 * g + 3] + (ly.l0 * (lx.l0 * v00.w + lx.l1 * v01.w) + ly.l1 * (lx.l0 * v10.w + lx.l1 * v11.w));
            }
        }
        uint32_t mw[4];
        uint2 t1[4], t2[4];
        if (MASK) {
            if (mw_pre) {
#pragma unroll
                for (int g = 0; g < 4; ++g) mw[g] = mw_pre[g];
            } else {
                const uint32_t* mb = e.mask_bits + (size_t)(b % e.mask_nb) * H * W;
#pragma unroll
                for (int g = 0; g < 4; ++g) mw[g] = mb[xq[g] >> 5];
            }
        }
        const bool two = ADD && e.add2 != nullptr;
        if (ADD) {
            const bf16_t* a1 = (const bf16_t*)e.add1 + (size_t)(b % e.add1_nb) * H * W * 32;
#pragma unroll
            for (int g = 0; g < 4; ++g) t1[g] = *(const uint2*)(a1 + xq[g]);
            if (two) {
                const bf16_t* a2 = (const bf16_t*)e.add2 + (size_t)(b % e.add2_nb) * H * W * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) t2[g] = *(const uint2*)(a2 + xq[g]);
            }
        }
        bf16_t* const oraw = (bf16_t*)e.out_raw + (size_t)b * H * W * 32;
        bf16_t* const osum = (bf16_t*)e.out_sum + (size_t)b * H * W * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 val = make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
            if (MASK) {
                const uint32_t nib = mw[g] >> j4;
                val.x = (nib & 1u) ? val.x : 0.0f; val.y = (nib & 2u) ? val.y : 0.0f;
                val.z = (nib & 4u) ? val.z : 0.0f; val.w = (nib & 8u) ? val.w : 0.0f;
            }
            if (e.out_raw && okg[g]) *(uint2*)(oraw + xq[g]) = f4_to_bf4(val);
            if (ADD) {
                float4 a = bf4_to_f4(t1[g]);
                if (two) { const float4 a2 = bf4_to_f4(t2[g]); a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w; }
                val.x += a.x; val.y += a.y; val.z += a.z; val.w += a.w;
            }
            if (e.out_sum && okg[g]) *(uint2*)(osum + xq[g]) = f4_to_bf4(val);
        }
        return;
    }
    if (MASK) {
        const T* mrow = e.mask + ((size_t)(b % e.mask_nb) * H * W + rowoff) * 32 + ch;
        float m[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = ld(mrow + (size_t)xo[r] * 32);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = m[r] > 0.0f ? v[r] : 0.0f;
    }
    const size_t obase = ((size_t)b * H * W + rowoff) * 32 + ch;
    if (e.out_raw) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (ok[r]) st(e.out_raw + obase + (size_t)xo[r] * 32, v[r]);
    }
    if (ADD) {
        const T* a1 = e.add1 + ((size_t)(b % e.add1_nb) * H * W + rowoff) * 32 + ch;
        float t1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) t1[r] = ld(a1 + (size_t)xo[r] * 32);
        if (e.add2) {
            const T* a2 = e.add2 + ((size_t)(b % e.add2_nb) * H * W + rowoff) * 32 + ch;
            float t2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t2[r] = ld(a2 + (size_t)xo[r] * 32);
#pragma unroll
            for (int r = 0; r < 16; ++r) t1[r] += t2[r];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += t1[r];
    }
    if (e.out_sum) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (ok[r]) st(e.out_sum + obase + (size_t)xo[r] * 32, v[r]);
    }
}

// MFMA 32x32 accumulator row of register r for lane half h (cdna_hip_programming.md §3):
// row = (r&3) + 8*(r>>2) + 4*h, col = lane&31.
// d loss_cos / d ref, one element: coef (e / |e| - c ref / |ref|) / |ref| with ie = 1 / |e|, ir = 1 / |ref| (loss.hip cos_grad_body; heads.hip PRO 4:
// the same expression in both, but the compiler is free to contract it differently in the two kernels -- last-bit differences)
__device__ __forceinline__ float cos_grad_elem(float coef, float e, float r, float ie, float ir, float proj) {
    return coef * (e * ie - proj * r * ir) * ir;
}
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// conv geometry
enum { CONV_S1 = 0, CONV_S2 = 1, CONV_T2 = 2 };

// ---- bilinear sampling with the deformable-convolution boundary rule (modulated_deform_im2col_cuda.cuh:25-54,:180):
// a sample contributes iff -1 < h < H and -1 < w < W; corners outside the image read 0
struct Corner { int h0, w0; float lh, lw; bool inside; };
__device__ __forceinline__ Corner corner_of(float h, float w, int H, int W) {
    Corner c;
    c.inside = (h > -1.f) && (w > -1.f) && (h < (float)H) && (w < (float)W);
    const float fh = floorf(h), fw = floorf(w);
    c.h0 = (int)fh; c.w0 = (int)fw; c.lh = h - fh; c.lw = w - fw;
    return c;
}
__device__ __forceinline__ float bilinear_at(const float* __restrict__ im, int H, int W, const Corner& c) {
    if (!c.inside) return 0.f;
    const int h1 = c.h0 + 1, w1 = c.w0 + 1;
    const float v1 = (c.h0 >= 0 && c.w0 >= 0) ? im[c.h0 * W + c.w0] : 0.f;
    const float v2 = (c.h0 >= 0 && w1 <= W - 1) ? im[c.h0 * W + w1] : 0.f;
    const float v3 = (h1 <= H - 1 && c.w0 >= 0) ? im[h1 * W + c.w0] : 0.f;
    const float v4 = (h1 <= H - 1 && w1 <= W - 1) ? im[h1 * W + w1] : 0.f;
    const float hh = 1.f - c.lh, hw = 1.f - c.lw;
    return hh * hw * v1 + hh * c.lw * v2 + c.lh * hw * v3 + c.lh * c.lw * v4;
}



// Workgroups are dealt to the 8 XCDs round-robin by their linear id, and every XCD has its own L2.  This bijective
// remap gives each XCD a CONTIGUOUS range of logical ids, so neighbours in the logical order (the output-channel tiles
// of one pixel tile, adjacent pixel tiles) share an L2 instead of each fetching the same input tile from HBM / the
// Infinity Cache through a different L2 (cdna_hip_programming.md, "XCD swizzle must be bijective").
__device__ __forceinline__ unsigned xcd_swizzle(unsigned orig, unsigned nwg) {
    const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding GLOBAL load
// (s_waitcnt vmcnt(0)), which turns a register prefetch issued before the barrier into a synchronous load; this one
// waits for the wave's LDS operations and leaves vector-memory loads in flight across the barrier
// (cdna_hip_programming.md, "Pipelining across barriers").
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

#define PTTA_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
