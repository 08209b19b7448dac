// First-layer and prediction convolutions of the MSG_CHN cascade: the ones that are NOT dense
// 32x32 contractions (network_exp_msg_chn_adapt.py: RGBEncoder.init[0] 3->32 :218, DepthEncoder
// .init[0] {1,2}->32 :172, DepthDecoder.prdct[3] 32->1 :291) and their input gradients.
//
//  * conv_in  : planar fp32 inputs with 1..3 channels -> 32-channel NHWC.  K = 9*cin <= 27 is
//               padded to an even number and run on the fp32 matrix cores (v_mfma_f32_32x32x2_f32)
//               so that the output fragment, and therefore the fused epilogue and the full-line
//               NHWC stores, are identical to conv32.  Inputs stay fp32 in both precision modes.
//               Also used as the input gradient of the 32->1 prediction conv (1 -> 32, flipped).
//  * conv_out1: 32-channel NHWC -> 1 planar fp32 channel, one lane per output pixel, HBM-bound
//               (reads each 128-B pixel line once from HBM, 8 more times from L1/L2).
//               Also the input gradient of conv_in w.r.t. one of its input planes.
#include "ptta_common.h"
#include "ptta_kernels.h"

template <typename T>
struct ConvInP {
    Plane pl[3]; int zero_from_b;
    const float* w;
    Epi<T> epi;
    int B, H, W;
};

template <typename T, int CIN>
__global__ __launch_bounds__(256) void conv_in_mfma_kernel(ConvInP<T> p) {
    constexpr int K = 9 * CIN;
    constexpr int NS = (K + 1) / 2;
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    float w[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) w[s] = p.w[s * 64 + lane];

    const int nseg = (p.W + 31) >> 5;
    const long nitems = (long)p.B * nseg * p.H;
    const float sy = up_scale(p.H >> 1, p.H), sx = up_scale(p.W >> 1, p.W);
    for (long item = (long)blockIdx.x * 4 + wave; item < nitems; item += (long)gridDim.x * 4) {
        long t_ = item;
        const int y = (int)(t_ % p.H); t_ /= p.H;
        const int seg = (int)(t_ % nseg);
        const int b = (int)(t_ / nseg);
        const int x0 = seg << 5;
        const bool live = (x0 + i) < p.W && b < p.zero_from_b;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            // lane half h contributes k = 2s + h -> (tap, ci); both candidates are compile-time
            const int k0 = 2 * s, k1 = 2 * s + 1;
            const int c0 = k0 % CIN, c1 = (k1 < K) ? (k1 % CIN) : 0;
            const int tap = h ? (k1 / CIN) : (k0 / CIN);
            const bool kval = h ? (k1 < K) : true;
            const int yi = y + tap / 3 - 1, xi = x0 + i + tap % 3 - 1;
            const float* pp = h ? p.pl[c1].p : p.pl[c0].p;          // static plane indices
            const int pnb = h ? p.pl[c1].nb : p.pl[c0].nb;
            const long pbs = h ? p.pl[c1].bstride : p.pl[c0].bstride;
            float a = 0.f;
            if (live && kval && yi >= 0 && yi < p.H && xi >= 0 && xi < p.W)
                a = pp[(size_t)(b % pnb) * pbs + (size_t)yi * p.W + xi];
            {
                const Plane& pq = h ? p.pl[c1] : p.pl[c0];
                if (pq.norm && live && kval && yi >= 0 && yi < p.H && xi >= 0 && xi < p.W) a = (a / pq.div - pq.mean) / pq.stdv;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[s], acc, 0, 0, 0);
        }
        Lerp ly = {0, 0, 0.f, 0.f};
        if (p.epi.up) ly = lerp_coef(y, p.H >> 1, sy);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = x0 + acc_row(r, h);
            if (px >= p.W) continue;
            Lerp lx = {0, 0, 0.f, 0.f};
            if (p.epi.up) lx = lerp_coef(px, p.W >> 1, sx);
            epi_store<T>(p.epi, b, y, px, p.H, p.W, i, acc[r], ly, lx);
        }
    }
}

// LDS-staged variant (the shipped one): a block owns an 8x32 output tile, the (8+2)x(32+2) window of
// each input plane is staged once (zero-filled outside the image / for the proxy pass's zero image), so the
// 9*cin A operands of a lane are ds_read_b32 at compile-time offsets instead of per-tap address arithmetic
// and bounds tests (the direct form spent ~36 VALU instructions per MFMA on them).
#define CI_TH 8
#define CI_PW 36            // padded row: 34 used
template <typename T, int CIN, bool MASK>
__global__ __launch_bounds__(256) void conv_in_lds_kernel(ConvInP<T> p) {
    constexpr int K = 9 * CIN, NS = (K + 1) / 2, PLANE = (CI_TH + 2) * CI_PW;
    __shared__ float tile[CIN * PLANE + 4];                      // + a zero word for the padded k
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const float biasv = epi_bias(p.epi, i);               // once per kernel: ptta_common.h epi_tile
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float w[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) w[s] = p.w[s * 64 + lane];
    if (tid == 0) tile[CIN * PLANE] = 0.f;
    const int H = p.H, W = p.W;
    const int ntx = (W + 31) >> 5, nty = (H + CI_TH - 1) / CI_TH;
    const long ntiles = (long)p.B * ntx * nty;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        long t_ = t;
        const int ty = (int)(t_ % nty); t_ /= nty;
        const int tx = (int)(t_ % ntx);
        const int b = (int)(t_ / ntx);
        const int y0 = ty * CI_TH, x0 = tx << 5;
        const bool live = b < p.zero_from_b;
        for (int idx = tid; idx < CIN * (CI_TH + 2) * 34; idx += 256) {
            const int px = idx % 34; int r_ = idx / 34;
            const int py = r_ % (CI_TH + 2); const int ci = r_ / (CI_TH + 2);
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            float v = 0.f;
            if (live && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const Plane& pl = p.pl[ci];
                v = pl.p[(size_t)(b % pl.nb) * pl.bstride + (size_t)gy * W + gx];
                if (pl.norm) v = (v / pl.div - pl.mean) / pl.stdv;
            }
            tile[ci * PLANE + py * CI_PW + px] = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr, y = y0 + row;
            if (y >= H) break;
            const float* base = tile + row * CI_PW + i;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k0 = 2 * s, k1 = 2 * s + 1;
                const int o0 = (k0 % CIN) * PLANE + ((k0 / CIN) / 3) * CI_PW + (k0 / CIN) % 3;
                const int o1 = (k1 < K) ? (k1 % CIN) * PLANE + ((k1 / CIN) / 3) * CI_PW + (k1 / CIN) % 3 : -1;
                // lane half h supplies k = 2s + h; the padded k reads the zero word
                const float a = h ? (o1 >= 0 ? base[o1] : tile[CIN * PLANE]) : base[o0];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w[s], acc, 0, 0, 0);
            }
            epi_tile<T, false, MASK, false>(p.epi, b, y, H, W, i, acc, x0, h, W, 1, 0, 0.f, 0.f, nullptr, 0, 0, &biasv);
        }
        __syncthreads();
    }
}

template <typename T>
__global__ void conv_in_naive_kernel(ConvInP<T> p, int cin) {
    const long total = (long)p.B * p.H * p.W * 32;
    const float sy = up_scale(p.H >> 1, p.H), sx = up_scale(p.W >> 1, p.W);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx & 31);
        long t_ = idx >> 5;
        const int x = (int)(t_ % p.W); t_ /= p.W;
        const int y = (int)(t_ % p.H);
        const int b = (int)(t_ / p.H);
        float acc = 0.f;
        if (b < p.zero_from_b)
            for (int tap = 0; tap < 9; ++tap) {
                const int yi = y + tap / 3 - 1, xi = x + tap % 3 - 1;
                if (yi < 0 || yi >= p.H || xi < 0 || xi >= p.W) continue;
                for (int ci = 0; ci < cin; ++ci) {
                    const Plane& pl = p.pl[ci];
                    float v = pl.p[(size_t)(b % pl.nb) * pl.bstride + (size_t)yi * p.W + xi];
                    if (pl.norm) v = (v / pl.div - pl.mean) / pl.stdv;
                    acc = fmaf(v, p.w[(tap * cin + ci) * 32 + co], acc);
                }
            }
        Lerp ly = {0, 0, 0.f, 0.f}, lx = {0, 0, 0.f, 0.f};
        if (p.epi.up) { ly = lerp_coef(y, p.H >> 1, sy); lx = lerp_coef(x, p.W >> 1, sx); }
        epi_store<T>(p.epi, b, y, x, p.H, p.W, co, acc, ly, lx);
    }
}

__global__ void pack_conv_in_kernel(const float* __restrict__ src, int cin_total, int cin_first, int cin,
                                    int transpose_flip, float* wfrag, float* wcanon) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;       // over 14*64 fragment slots
    if (idx >= 14 * 64) return;
    const int lane = idx & 63, s = idx >> 6;
    const int co = lane & 31, h = lane >> 5;
    const int k = 2 * s + h;
    float v = 0.f;
    if (k < 9 * cin) {
        const int tap = k / cin, ci = k % cin;
        v = transpose_flip ? src[(size_t)co * 9 + (8 - tap)]
                           : src[((size_t)co * cin_total + cin_first + ci) * 9 + tap];
        wcanon[(tap * cin + ci) * 32 + co] = v;
    }
    wfrag[idx] = v;
}

void ptta_pack_conv_in(const float* src, int cin_total, int cin_first, int cin, int transpose_flip,
                       float* wfrag, float* wcanon, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_in_kernel, dim3(4), dim3(256), 0, s, src, cin_total, cin_first, cin,
                       transpose_flip, wfrag, wcanon);
}

template <typename T>
static int launch_conv_in_t(const ConvInArgs& a, hipStream_t s) {
    ConvInP<T> p;
    for (int c = 0; c < 3; ++c) { p.pl[c] = a.pl[c]; if (p.pl[c].nb < 1) p.pl[c].nb = 1; }
    p.zero_from_b = a.zero_from_b;
    p.epi.bias = a.bias;
    p.epi.up = (const T*)a.up; p.epi.up_nb = a.up_nb > 0 ? a.up_nb : 1;
    p.epi.mask = (const T*)a.mask; p.epi.mask_nb = a.mask_nb > 0 ? a.mask_nb : 1;
    p.epi.add1 = (const T*)a.add1; p.epi.add1_nb = a.add1_nb > 0 ? a.add1_nb : 1;
    p.epi.add2 = nullptr; p.epi.add2_nb = 1;
    p.epi.out_raw = (T*)a.out_raw; p.epi.out_sum = (T*)a.out_sum;
    p.epi.mask_bits = a.mask_bits;              // fp32 output: optional; narrow output: the only form a ReLU mask can take
    if (sizeof(T) == 4 && a.a_bits && a.a_bits_nb > 0 && !a.naive) { p.epi.bits_out = a.a_bits; p.epi.bits_nb = a.a_bits_nb; p.epi.bits_sum = 0; }
    if (sizeof(T) == 2 && a.mask && !a.mask_bits) return -22;
    p.B = a.B; p.H = a.H; p.W = a.W;
    if (a.naive) {
        p.w = a.wcanon;
        const long total = (long)p.B * p.H * p.W * 32;
        int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL((conv_in_naive_kernel<T>), dim3(blocks), dim3(256), 0, s, p, a.cin);
    } else {
        p.w = a.wfrag;
        if (!a.up && !a.add1) {          // every use on the path: bias (+ ReLU mask in the backward) only
            const long tiles = (long)p.B * ((p.W + 31) / 32) * ((p.H + CI_TH - 1) / CI_TH);
            const int blocks = (int)(tiles > 2048 ? 2048 : tiles);
#define CL_(C) do { if (a.mask) hipLaunchKernelGGL((conv_in_lds_kernel<T, C, true>), dim3(blocks), dim3(256), 0, s, p); \
                    else hipLaunchKernelGGL((conv_in_lds_kernel<T, C, false>), dim3(blocks), dim3(256), 0, s, p); } while (0)
            if (a.cin == 1) CL_(1); else if (a.cin == 2) CL_(2); else CL_(3);
#undef CL_
            PTTA_CHECK_LAUNCH();
            return 0;
        }
        const long items = (long)p.B * ((p.W + 31) / 32) * p.H;
        long blocks = (items + 3) / 4; if (blocks > 2048) blocks = 2048;
        if (a.cin == 1) hipLaunchKernelGGL((conv_in_mfma_kernel<T, 1>), dim3((int)blocks), dim3(256), 0, s, p);
        else if (a.cin == 2) hipLaunchKernelGGL((conv_in_mfma_kernel<T, 2>), dim3((int)blocks), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv_in_mfma_kernel<T, 3>), dim3((int)blocks), dim3(256), 0, s, p);
    }
    PTTA_CHECK_LAUNCH();
    return 0;
}

int ptta_launch_conv_in(const ConvInArgs& a, hipStream_t s) {
    if (a.cin < 1 || a.cin > 3) return -22;
    return a.bf16 ? launch_conv_in_t<bf16_t>(a, s) : launch_conv_in_t<float>(a, s);
}

// ---- 32 -> 1 ------------------------------------------------------------------------------------
// 8 lanes per pixel (each lane 4 channels = 16 B of the pixel's 128-B line, 8 B in narrow storage); the 288 weights live in 36 VGPRs; the
// 8 partial sums of a pixel are combined with three xor-shuffles.  (The round-1 strip form -- a wave walking down 8 columns with 18 loads
// per lane straight from L1/L2 -- moved every input pixel across the vector-memory path 4.5 times; removed in round 5.)
// LDS-staged: a block owns an 8 x 32 output tile and stages its (8+2) x (32+2) x 32-channel
// window ONCE as whole 128-B pixel lines (zero padding and the input ReLU applied while staging), so every input pixel crosses the
// vector-memory path 1.33 times instead of the 4.5 times of the strip kernel above (18 float4 loads per lane and 4 x 8 outputs).
// Same lane roles (8 lanes per pixel, 4 channels each), same tap order and the same xor-shuffle reduction: bit-identical sums.
// The 32 results of an output row are collected into one half-wave (lane quad index = pass) and leave as ONE 128-B store.
template <typename T, bool RELU>
__global__ __launch_bounds__(256) void conv_out1_lds_kernel(const T* __restrict__ in, int in_nb, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const float* __restrict__ add,
                                                            int add_nb, float* __restrict__ out, int B, int H, int W) {
    constexpr int TH = 8, PW = 34, PH = TH + 2, NV = PH * PW * 8, NIT = (NV + 255) / 256;
    __shared__ float4 tile[NV];
    const int tid = threadIdx.x, lane = tid & 63, pl = lane >> 3, cq = lane & 7;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    float4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = *(const float4*)(w + t * 32 + 4 * cq);
    const float b0 = bias ? bias[0] : 0.f;
    const int ntx = (W + 31) >> 5, nty = (H + TH - 1) / TH;
    const long ntiles = (long)B * ntx * nty;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        long t_ = t;
        const int ty = (int)(t_ % nty); t_ /= nty;
        const int tx = (int)(t_ % ntx);
        const int b = (int)(t_ / ntx);
        const int y0 = ty * TH, x0 = tx << 5;
        const T* inb = in + (size_t)(b % in_nb) * H * W * 32;
        // all loads of the window first (unconditional, clamped), zero padding / ReLU afterwards (narrow storage: 8 B per piece, widened here)
        float4 v[NIT];
        unsigned okm = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = min(tid + 256 * it, NV - 1);
            const int q = idx & 7, pix = idx >> 3;
            const int py = pix / PW, px = pix - py * PW;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            okm |= (gy >= 0 && gy < H && gx >= 0 && gx < W) ? (1u << it) : 0u;
            const T* src = inb + ((size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * 32 + 4 * q;
            if constexpr (sizeof(T) == 4) v[it] = *(const float4*)src; else v[it] = bf4_to_f4(*(const uint2*)src);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            float4 a = v[it];
            if (!((okm >> it) & 1u)) a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
            if (idx < NV) tile[idx] = a;
        }
        lds_barrier();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int row = 2 * wave + rr, y = y0 + row;
            const int xs = x0 + cq * 8 + pl;                  // the pixel this lane stores (lanes with cq < 4)
            const bool sok = cq < 4 && y < H && xs < W;
            const size_t o = (size_t)min(y, H - 1) * W + min(xs, W - 1);
            float addv = 0.f;
            if (add) addv = add[(size_t)(b % add_nb) * H * W + o];
            float res = 0.f;
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const float4* base = tile + (row * PW + pass * 8 + pl) * 8 + cq;
                float acc = 0.f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const float4 a = base[((tap / 3) * PW + tap % 3) * 8];
                    acc = fmaf(a.x, wt[tap].x, acc); acc = fmaf(a.y, wt[tap].y, acc);
                    acc = fmaf(a.z, wt[tap].z, acc); acc = fmaf(a.w, wt[tap].w, acc);
                }
                acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
                res = cq == pass ? acc : res;
            }
            if (sok) {
                float r = res + b0;
                if (add) r += addv;
                out[(size_t)b * H * W + o] = r;
            }
        }
        lds_barrier();
    }
}

__global__ void pack_conv_out1_kernel(const float* __restrict__ src, int cin_total, int cin_index, int from_conv_in, float* w) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 288) return;
    const int c = idx & 31, tap = idx >> 5;
    w[idx] = from_conv_in ? src[((size_t)c * cin_total + cin_index) * 9 + (8 - tap)] : src[(size_t)c * 9 + tap];
}

void ptta_pack_conv_out1(const float* src, int src_cin_total, int src_cin_index, int from_conv_in, float* w, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_out1_kernel, dim3(2), dim3(256), 0, s, src, src_cin_total, src_cin_index, from_conv_in, w);
}

int ptta_launch_conv_out1(const ConvOut1Args& a, hipStream_t s) {
    const int add_nb = a.add_nb > 0 ? a.add_nb : 1;
    const long tiles = (long)a.B * ((a.W + 31) / 32) * ((a.H + 7) / 8);
    const int tb = (int)(tiles < 768 ? tiles : 768);                                            // three 43.5-KB blocks per CU
#define LDS_(T, R) hipLaunchKernelGGL((conv_out1_lds_kernel<T, R>), dim3(tb), dim3(256), 0, s, (const T*)a.in, a.in_nb, a.w, a.bias, a.add, add_nb, a.out, a.B, a.H, a.W)
    if (a.bf16) { if (a.relu_in) LDS_(bf16_t, true); else LDS_(bf16_t, false); }
    else { if (a.relu_in) LDS_(float, true); else LDS_(float, false); }
#undef LDS_
    PTTA_CHECK_LAUNCH();
    return 0;
}
