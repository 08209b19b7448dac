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
};

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

// MFMA 32x32 accumulator row of register r for lane half h (cdna_hip_programming.md §3):
// row = (r&3) + 8*(r>>2) + 4*h, col = lane&31.
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// conv geometry
enum { CONV_S1 = 0, CONV_S2 = 1, CONV_T2 = 2 };

#define PTTA_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
