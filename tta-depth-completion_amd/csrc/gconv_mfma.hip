// Matrix-core path of the generic NHWC convolution (NLSPN backbone): stride-1 3x3 (and 1x1) convolutions with any
// channel counts that are multiples of 8, one or two channel-concatenated sources, fp32 storage, bf16x3 arithmetic
// (x = xh + xl, w = wh + wl; xl*wh + xh*wl + xh*wh on v_mfma_f32_32x32x16_bf16, fp32 accumulate) -- the same
// arithmetic as the MSG_CHN hot path (conv32.hip), generalised over input-channel chunks and output-channel tiles.
//
// Block = 256 threads = 4 waves, output tile 8 rows x 32 pixels x 32 output channels (one MFMA column fragment).
// K loop over 16-channel sub-chunks of the source(s): the (8+2)x(32+2) halo of the sub-chunk is split into bf16 hi/lo
// while it is staged into LDS ([pixel][hi 32 B | lo 32 B | pad 16 B]: conflict-free ds_read_b128) and ALL weight
// fragments of the sub-chunk (hi and lo, 18 KB for 3x3, pre-split in global memory) are staged cooperatively next to
// it -- 45 KB of LDS and ~160 VGPRs, three blocks per CU.  Each wave owns two rows of the tile: 2 x f32x16
// accumulators live across the whole K loop; the next sub-chunk's halo and weights are loaded into registers before
// the current sub-chunk's MFMAs.  Output channel tiles of the same pixel tile are adjacent in the grid (L2 reuse).
// (Measured alternatives, all slower: hi fragments in registers per wave -- 2.5x the weight traffic through L1, the
// bottleneck by ablation; 64-channel output tiles -- re-measured in round 2 as a template variant with two accumulator sets
// per wave (8 instead of 12 ds_read_b128 per 12 MFMAs, halo staged once per 64 channels): 256 VGPRs, 64 KB of LDS, two
// blocks per CU instead of three and the NLSPN step went 26.87 -> 28.35 ms on one box, i.e. the kernel is bound by how many
// blocks overlap each other's stage -> barrier -> MFMA phases, not by LDS read bandwidth; double-buffered LDS with hi
// fragments in registers.  The ISA of THIS form drains its own register prefetch: the last weight load is conditional and the
// compiler merges its result right behind it (s_waitcnt vmcnt(0) + v_mov in front of the barrier).  A version with two explicit
// register sets and unconditional loads keeps 11 loads in flight across the MFMAs, needs 234 VGPRs (two blocks per CU) and
// measures the same: 25.73 vs 25.66 ms per NLSPN step, A/B on one box -- three non-overlapping blocks hide as much as two
// overlapping ones, and neither is bound by the loads.)
#include <cstdlib>
#include "ptta_common.h"
#include "ptta_kernels.h"

namespace {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void gsplit2(float a, float b, unsigned& hi, unsigned& lo) {
    float2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    float2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
}

// three-way split a = h + m + l (8 + 8 + 8 significant bits): h and m are the two-way split's (hi, lo); l2 is what that split drops
__device__ __forceinline__ void gsplit3(float a, float b, unsigned& hi, unsigned& lo, unsigned& l2) {
    float2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    hi = __builtin_bit_cast(unsigned, h);
    float2_t r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    const bf16x2_t m = __builtin_convertvector(r, bf16x2_t);
    lo = __builtin_bit_cast(unsigned, m);
    float2_t r2 = {r[0] - __uint_as_float(lo << 16), r[1] - __uint_as_float(lo & 0xffff0000u)};
    l2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
}

#define GX_TH 8

// fragment (nf, chunk, tap, kk): lane l holds column co = 32 nf + (l & 31), rows ci = 32 chunk + 16 kk + 8 (l >> 5) + e
__global__ void gfrag_pack_kernel(const float* __restrict__ canon, long wld, long wts, int KK, int C0, int C1, int c0_0, int c0_1,
                                  int Co, int nchunks, int nf_total, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, bf16_t* __restrict__ l2) {
    const long total = (long)nf_total * nchunks * KK * 2 * 64 * 8;
    const int nch0 = (C0 + 31) / 32;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 7); long t_ = idx >> 3;
        const int lane = (int)(t_ & 63); t_ >>= 6;
        const int kk = (int)(t_ & 1); t_ >>= 1;
        const int t = (int)(t_ % KK); t_ /= KK;
        const int c = (int)(t_ % nchunks); const int nf = (int)(t_ / nchunks);
        const int s = c < nch0 ? 0 : 1;
        const int cl = (s == 0 ? c : c - nch0) * 32 + kk * 16 + (lane >> 5) * 8 + e;
        const int Cs = s == 0 ? C0 : C1;
        const int co = nf * 32 + (lane & 31);
        float v = 0.f;
        if (cl < Cs && co < Co) v = canon[(long)t * wts + (long)((s == 0 ? c0_0 : c0_1) + cl) * wld + co];
        const bf16_t h = f2bf(v);
        const float r = v - bf2f(h);
        const bf16_t m = f2bf(r);
        hi[idx] = h; lo[idx] = m;
        if (l2) l2[idx] = f2bf(r - bf2f(m));
    }
}

#define GX_STR16 80                                      // LDS bytes per pixel: hi 32 | lo 32 | pad 16
// VERT: only the middle column of the 3x3 window (a 3x1x1 Conv3d viewed as a vertical 3-tap convolution over [D][H*W] images,
// gnet.h Op::rH/rW): three taps, no horizontal halo -- a third of the MFMAs of the zero-padded 3x3 form it replaces.
// TIMING: s_memtime stamps around the phases of one block's K loop, printed by two blocks of the launch (diagnostic builds of the
// launcher are compiled with -DPTTA_DIAG_STAMPS; the shipped instantiation has no stamps)
#define STAMP(v) do { if constexpr (TIMING) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); v = t_; } } while (0)
// SIX: three operand planes (GX3Args::six_B): 112 B of LDS per pixel, three weight planes, two blocks per CU; blocks of images
// >= six_B (the proxy frames) skip the three extra products
template <int KS, bool VERT, bool TIMING = false, bool SIX = false>
__global__ __launch_bounds__(256, SIX ? 2 : 3) void gconv_x3_s1_kernel(GX3Args p) {
    unsigned long long T0 = 0, ta = 0, tb = 0, tc = 0, td = 0, te = 0, tf = 0, tg = 0;
    unsigned long long dA = 0, dW = 0, dS = 0, dL = 0, dB = 0, dM = 0;
    STAMP(T0);
    constexpr int PAD = KS / 2, PADX = VERT ? 0 : PAD, KKX = VERT ? 1 : KS, PH = GX_TH + 2 * PAD, PW = 32 + 2 * PADX, KK = KS * KKX;
    constexpr int NPIX = PH * PW;
    constexpr int NIT = (NPIX * 2 + 255) / 256;          // (pixel, 8-channel group) items per thread
    constexpr int NPL = SIX ? 3 : 2;                     // operand planes
    constexpr int STR = SIX ? 112 : GX_STR16;            // LDS bytes per pixel: hi 32 | lo 32 [| l2 32] | pad 16 (an odd number of 16-B slots)
    constexpr int NWF = KK * 64 * NPL;                   // weight uint4 per sub-chunk: [hi | lo | l2][tap][lane]
    constexpr int NW = (NWF + 255) / 256;
    constexpr int ACT = NPIX * STR;
    constexpr int LDSB = ACT + NWF * 16 > 32768 ? ACT + NWF * 16 : 32768;     // the epilogue stages 4 x 8 KB through it
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB + 1024];
    uint4* const wlds = (uint4*)(lds + ACT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    const int ntx = (W + 31) >> 5, nty = (H + GX_TH - 1) / GX_TH;
    const unsigned bid = xcd_swizzle(blockIdx.x, gridDim.x);      // the channel tiles of one pixel tile stay on one XCD / L2
    const int nfl = (int)(bid % p.nnf);
    long t_ = bid / p.nnf;
    const int ty = (int)(t_ % nty); t_ /= nty;
    const int tx = (int)(t_ % ntx);
    const int b = (int)(t_ / ntx);
    const int y0 = ty * GX_TH, x0 = tx << 5;
    const int nf = p.nf0 + nfl;
    const int nch0 = (p.C0 + 31) >> 5;
    const int nq = 2 * p.nchunks;
    const bool six = SIX && b < p.six_B;                 // block-uniform
    const bool x1 = b >= p.x1_from_B;                    // block-uniform: images of the single-MFMA class (mixed mode: proxy frames, data gradients)

    // (weight planes: one base pointer + a per-lane byte offset to the lane's plane -- see issue_loads)
    const char* const wbase_p = (const char*)p.whi;
    const long wd_lo = (const char*)p.wlo - wbase_p, wd_l2 = SIX ? (const char*)p.wl2 - wbase_p : wd_lo;
    float4 v0[NIT], v1[NIT];
    unsigned vok = 0;                                    // bit `it`: item it of the sub-chunk in flight is inside the image and the channel range
    uint4 wr[NW];
    auto issue_loads = [&](int q) __attribute__((always_inline)) {
        const int c = q >> 1, kk = q & 1;
        const bool s1 = c >= nch0;
        const float* src = s1 ? p.x1 : p.x0;
        const int ld = s1 ? p.ld1 : p.ld0, Cs = s1 ? p.C1 : p.C0, cb = ((s1 ? c - nch0 : c) << 5) + 16 * kk;
        const float* inb = src + (size_t)b * H * W * ld;
        // UNCONDITIONAL loads from clamped addresses; zero padding is applied when the values are split into LDS (vok).  Behind
        // `if (inside) { v0 = load; v1 = load; }` hipcc waited vmcnt(0) between the items of one sub-chunk (round 5, .s): the wave
        // that issues the prefetch stalled for a memory round trip per item instead of going on to its matrix instructions.
        vok = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int g = idx & 1, pix = idx >> 1;
            const int py = pix / PW, px = pix - py * PW;
            const int gy = y0 - PAD + py, gx = x0 - PADX + px;
            const bool ok = pix < NPIX && gy >= 0 && gy < H && gx >= 0 && gx < W && cb + 8 * g < Cs;
            const float* s_ = inb + ((size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)) * ld + min(cb + 8 * g, Cs - 8);
            v0[it] = *(const float4*)s_; v1[it] = *(const float4*)(s_ + 4);
            vok |= ok ? (1u << it) : 0u;
        }
        const size_t wbase = ((size_t)nf * p.nchunks + c) * (KK * 2 * 64) + kk * 64;
        // The three weight planes are three allocations and a lane's plane index varies within one j.  Written as
        // (hl == 0 ? p.whi : hl == 1 ? p.wlo : p.wl2)[...] hipcc selects the ADDRESS of the kernel argument and loads the pointer from
        // memory -- a dependent load whose s_waitcnt vmcnt(0) also waited for the six activation loads issued just above: every wave
        // stalled for a full memory round trip per sub-chunk, twice, and the prefetch never overlapped the matrix phase (round 5, .s).
        // The plane is therefore chosen by a byte OFFSET from the hi plane's pointer (wd_lo, wd_l2: scalar differences formed once): two
        // v_cndmask on registers, the load unconditional from a clamped index.
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int idx = min(tid + 256 * j, NWF - 1);  // [hl][tap][lane]  (items beyond NWF are not written to LDS)
            const int hl = idx / (KK * 64), r = idx - hl * (KK * 64);
            const uint4* wp = (const uint4*)(wbase_p + (hl == 0 ? 0L : (hl == 1 ? wd_lo : wd_l2)));
            const uint4 t = wp[wbase + (r >> 6) * 128 + (r & 63)];
            wr[j].x = t.x; wr[j].y = t.y; wr[j].z = t.z; wr[j].w = t.w;       // (member-wise: `wr[j] = wp[...]` is a memcpy into the array, which then lives in scratch)
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rr][r] = 0.f;

    issue_loads(0);
    // fetched here, not in the epilogue: there its L2 round trip sat between the last MFMA and the first store (in-kernel stamps)
    const int co = nf * 32 + i - p.nf0 * 32;          // channel inside the output view
    const bool cok = co < p.Cy;
    const float bias = (cok && p.bias) ? p.bias[co] : 0.f;
    unsigned long long Tpro = 0; STAMP(Tpro);
    for (int q = 0; q < nq; ++q) {
        STAMP(ta);
        if (q) lds_barrier();                            // previous sub-chunk's MFMAs are done with the LDS tile
        STAMP(tb);
        if constexpr (TIMING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(tc);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + 256 * it;
            const int pix = idx >> 1;
            if (pix < NPIX) {
                const bool ok = (vok >> it) & 1u;
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 a0 = ok ? v0[it] : z4, a1 = ok ? v1[it] : z4;
                uint4 hi, lo;
                unsigned char* dst = lds + pix * STR + 16 * (idx & 1);
                if constexpr (SIX) {
                    uint4 l2;
                    gsplit3(a0.x, a0.y, hi.x, lo.x, l2.x); gsplit3(a0.z, a0.w, hi.y, lo.y, l2.y);
                    gsplit3(a1.x, a1.y, hi.z, lo.z, l2.z); gsplit3(a1.z, a1.w, hi.w, lo.w, l2.w);
                    *(uint4*)(dst + 64) = l2;
                } else {
                    gsplit2(a0.x, a0.y, hi.x, lo.x); gsplit2(a0.z, a0.w, hi.y, lo.y);
                    gsplit2(a1.x, a1.y, hi.z, lo.z); gsplit2(a1.z, a1.w, hi.w, lo.w);
                }
                *(uint4*)dst = hi;
                *(uint4*)(dst + 32) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int idx = tid + 256 * j;
            if (idx < NWF) wlds[idx] = wr[j];
        }
        STAMP(td);
        if (q + 1 < nq) issue_loads(q + 1);
        STAMP(te);
        lds_barrier();                                   // LDS-only: the loads just issued stay in flight during the MFMAs
        STAMP(tf);
        __builtin_amdgcn_s_setprio(1);                    // MFMA phase outranks the other blocks' staging code at issue
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int ky = tap / KKX, kx = tap % KKX;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, wlds[tap * 64 + lane]);
            const bf16x8 bl = __builtin_bit_cast(bf16x8, wlds[KK * 64 + tap * 64 + lane]);
            // the two rows' MFMAs alternate: consecutive matrix instructions never accumulate into the same registers
            bf16x8 ah[2], al[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const unsigned char* a = lds + ((2 * wave + rr + ky) * PW + i + kx) * STR + 16 * h;
                ah[rr] = __builtin_bit_cast(bf16x8, *(const uint4*)a);
                al[rr] = __builtin_bit_cast(bf16x8, *(const uint4*)(a + 32));
            }
            if constexpr (SIX) if (six) {                 // the 2^-16 terms first: l.h, h.l, m.m
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, wlds[2 * KK * 64 + tap * 64 + lane]);
                bf16x8 a2[2];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
                    a2[rr] = __builtin_bit_cast(bf16x8, *(const uint4*)(lds + ((2 * wave + rr + ky) * PW + i + kx) * STR + 16 * h + 64));
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[0], bh, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[1], bh, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], b2, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], b2, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0], bl, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[1], bl, acc[1], 0, 0, 0);
            }
            if (!x1) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[0], bh, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[1], bh, acc[1], 0, 0, 0);
            }
            if (!x1 || p.x1_w2) {                          // (x1 with hi + lo WEIGHTS: the data gradients of the mixed mode, GX3Args::x1_w2)
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bl, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bl, acc[1], 0, 0, 0);
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[0], bh, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[1], bh, acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        STAMP(tg);
        dA += tb - ta; dW += tc - tb; dS += td - tc; dL += te - td; dB += tf - te; dM += tg - tf;
    }
    unsigned long long Tloop = 0; STAMP(Tloop);
    float s1 = 0.f, s2 = 0.f;
    auto activate = [&](float v) {
        if (p.act == GACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == GACT_LRELU) v = v > 0.f ? v : 0.2f * v;
        else if (p.act == GACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        return v;
    };
    // selects only (the compiler if-converts `activate` and then evaluates expf and the division for every element of every launch:
    // 6k of a wave's 36k cycles by the in-kernel stamps); the sigmoid (one layer of the network) runs in its own uniform branch
    auto activate_cheap = [&](float v) {
        const float neg = p.act == GACT_LRELU ? 0.2f * v : (p.act == GACT_RELU ? 0.f : v);
        return v > 0.f ? v : neg;
    };
    // The accumulator layout (lane = channel, register = pixel) would store one dword per lane, 32 store instructions per wave, and
    // the vector-memory instruction path is what this kernel is bound by (in-kernel stamps: the epilogue was 35-40 % of a wave's
    // life).  Transposed through the wave's own 8 KB of the (now idle) tile buffers -- [row 2][pixel 32][channel 32] floats, written
    // conflict-free one pixel per half-wave, read back as float4 -- every lane stores 16 B: 8 store instructions per wave, each
    // covering eight pixels x 128 contiguous bytes.  Bias / activation / statistics stay in the register layout (per-lane channel);
    // a launch that accumulates adds the old values as float4 after the transpose and activates there.
    const bool sig = p.act == GACT_SIGMOID;
    const bool wide = !(p.accumulate && (p.stat_part || sig)) && !(p.ldy & 3) && !(p.Cy & 3) && !(((size_t)p.y) & 15);
    unsigned long long E1 = 0, E2 = 0, E3 = 0;
    if (wide) {
        lds_barrier();                                    // every wave is done with the tile buffers
        STAMP(E1);
        float* const tw = (float*)lds + wave * 2048;
        float bias_w = bias;
        if (sig) {                                        // uniform
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rr][r] = 1.f / (1.f + expf(-(acc[rr][r] + bias)));
            bias_w = 0.f;
        }
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const bool yok = y0 + 2 * wave + rr < H;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int px = acc_row(r, h);
                float v = sig ? acc[rr][r] : acc[rr][r] + bias_w;
                if (!p.accumulate) {
                    v = sig ? v : activate_cheap(v);
                    if (yok && cok && x0 + px < W) { s1 += v; s2 += v * v; }
                }
                tw[(rr * 32 + px) * 32 + i] = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // wave-local: a wave's LDS operations execute in order
        STAMP(E2);
        const int c4 = (lane & 7) * 4, pl = lane >> 3;
        const int cbase = nf * 32 - p.nf0 * 32 + c4;
        if (cbase < p.Cy) {
            // an accumulating launch reads the eight old values FIRST, unconditionally (clamped pixel): one load + s_waitcnt + store per
            // pixel group was eight memory round trips in a row per wave
            float4 old[8];
            if (p.accumulate) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int rr = k >> 2, px = (k & 3) * 8 + pl;
                    const int y = min(y0 + 2 * wave + rr, H - 1), x = min(x0 + px, W - 1);
                    old[k] = *(const float4*)(p.y + (((size_t)b * H + y) * W + x) * p.ldy + cbase);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int rr = k >> 2, px = (k & 3) * 8 + pl;
                const int y = y0 + 2 * wave + rr, x = x0 + px;
                if (y >= H || x >= W) continue;
                float4 v = *(const float4*)(tw + (rr * 32 + px) * 32 + c4);
                float* dst = p.y + (((size_t)b * H + y) * W + x) * p.ldy + cbase;
                if (p.accumulate) {
                    const float4 o = old[k];
                    v.x = activate_cheap(v.x + o.x); v.y = activate_cheap(v.y + o.y); v.z = activate_cheap(v.z + o.z); v.w = activate_cheap(v.w + o.w);
                }
                *(float4*)dst = v;
            }
        }
    } else {
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int y = y0 + 2 * wave + rr;
        if (y >= H || !cok) continue;
        float* yrow = p.y + ((size_t)b * H + y) * W * p.ldy + co;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int x = x0 + acc_row(r, h);
            if (x >= W) continue;
            float* dst = yrow + (size_t)x * p.ldy;
            float v = acc[rr][r] + bias;
            if (p.accumulate) v += *dst;
            v = activate(v);
            *dst = v;
            s1 += v; s2 += v * v;
        }
    }
    }
    STAMP(E3);
    if (p.stat_part) {
        // fused BatchNorm statistics of this tile: lane halves, then the four waves through LDS (fixed order)
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        float* red = (float*)(lds + LDSB);                // [wave 4][2][32], its own 1 KB behind the tile buffers
        if (h == 0) { red[(wave * 2 + 0) * 32 + i] = s1; red[(wave * 2 + 1) * 32 + i] = s2; }
        lds_barrier();
        if (tid < 64 && (tid & 31) + nf * 32 - p.nf0 * 32 < p.Cy) {
            const int which = tid >> 5, ch = tid & 31;
            const float v = (red[(0 * 2 + which) * 32 + ch] + red[(1 * 2 + which) * 32 + ch]) + (red[(2 * 2 + which) * 32 + ch] + red[(3 * 2 + which) * 32 + ch]);
            const int bpp = p.B / p.stat_npass, pass = b / bpp;
            const long tile = ((long)(b - pass * bpp) * ntx + tx) * nty + ty;
            const long tiles_pp = (long)bpp * ntx * nty;
            p.stat_part[((pass * tiles_pp + tile) * 2 + which) * p.stat_C + (nf - p.nf0) * 32 + ch] = v;
        }
    }
    if constexpr (TIMING) {
        unsigned long long Tend = 0; STAMP(Tend);
        if ((blockIdx.x == gridDim.x / 2 || blockIdx.x == gridDim.x / 3) && lane == 0)
            printf("blk %u/%u wave %d nq %d: life %llu pro %llu loop %llu epi %llu (barrier %llu transpose-write %llu read+store %llu stats %llu) | barA %llu vmwait %llu split %llu issue %llu barB %llu mfma %llu\n", blockIdx.x, gridDim.x, wave, nq,
                   Tend - T0, Tpro - T0, Tloop - Tpro, Tend - Tloop, E1 - Tloop, E2 - E1, E3 - E2, Tend - E3, dA, dW, dS, dL, dB, dM);
    }
}

// ---- stride-2 and transposed (fractionally strided) geometries: direct A loads, no LDS ----------------------------
// One wave = 32 outputs of one output row (transposed: of one x parity, so that all lanes use the same taps) x 32
// output channels.  A fragments (8 fp32 channels of one input pixel per lane) are loaded straight from global
// memory, split into bf16 hi/lo in registers and fed to three MFMAs; weight fragments (hi and lo) stream from L1/L2.
// These layers are ~5 % of the network's multiply-accumulates (ResNet stage entries, conv6, the four decoder
// transposed convolutions and their data gradients).
//   MODE 1 (S2): y[oy][ox] = sum x[2oy+ky-pad][2ox+kx-pad] w[tap]
//   MODE 2 (T2): y[oy][ox] = sum over taps with (oy+pad-ky), (ox+pad-kx) even of x[(oy+pad-ky)/2][(ox+pad-kx)/2] w[tap]
// NCO = output-channel tiles (of 32) per wave: with two, every A fragment (uncoalesced 32-B pieces of strided pixels, split to
// bf16 in registers) feeds two weight fragments
// SIX: bf16x6 for the images below p.six_B (GX3Args), the third operand planes split in registers / streamed like the other two
template <int MODE, int KS, int KSPLIT, int NCO, bool SIX = false>
__global__ __launch_bounds__(256) void gconv_x3_direct_kernel(GX3Args p, int Hin, int Win) {
    constexpr int PAD = KS / 2, KK = KS * KS;
    // stride 2: two output rows per wave, tap-outer / row-inner, so the A loads of both rows are in flight before the
    // first MFMA and the weight fragments are fetched once per two rows; transposed: one row (its active taps depend on
    // the row parity)
    constexpr int R = MODE == 1 ? 2 : 1;
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int Hout = p.H, Wout = p.W;
    const int Wt = MODE == 2 ? Win : Wout;
    const int nseg = (Wt + 31) >> 5, npar = MODE == 2 ? 2 : 1, nyg = (Hout + R - 1) / R;
    const long nitems = (long)p.B * nseg * npar * nyg;
    const int nch0 = (p.C0 + 31) >> 5;
    // block -> (4 consecutive items, channel tile); channel tile fastest.  KSPLIT = 4 (few pixels, many channels: the
    // 1/8 and 1/16-resolution layers): the block's four waves share ONE item and split its K loop (chunk c goes to wave
    // c mod 4), partial accumulators are summed through LDS in a fixed order -- 4x more blocks, 4x shorter latency chains
    const int ngrp = p.nnf / NCO;
    const int nfl = (int)(blockIdx.x % ngrp) * NCO;
    const long item = KSPLIT == 4 ? (long)(blockIdx.x / ngrp) : (long)(blockIdx.x / ngrp) * 4 + wave;
    if (item >= nitems) return;
    const int nf = p.nf0 + nfl;
    long t_ = item;
    const int yg = (int)(t_ % nyg); t_ /= nyg;
    int xpar = 0;
    if (MODE == 2) { xpar = (int)(t_ & 1); t_ >>= 1; }
    const int seg = (int)(t_ % nseg);
    const int b = (int)(t_ / nseg);
    const int x0 = seg << 5, ybase = yg * R;
    const bool lane_in = (x0 + i) < Wt;
    const bool six = SIX && b < p.six_B;                 // wave-uniform
    const bool x1 = b >= p.x1_from_B;                    // wave-uniform: single-MFMA class
    f32x16 acc[NCO][R];
#pragma unroll
    for (int t = 0; t < NCO; ++t)
#pragma unroll
    for (int rr = 0; rr < R; ++rr)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][rr][r] = 0.f;
    for (int c = KSPLIT == 4 ? wave : 0; c < p.nchunks; c += KSPLIT == 4 ? 4 : 1) {
        const bool s1 = c >= nch0;
        const float* src = s1 ? p.x1 : p.x0;
        const int ld = s1 ? p.ld1 : p.ld0, Cs = s1 ? p.C1 : p.C0, cb = (s1 ? c - nch0 : c) << 5;
        const float* inb = src + (size_t)b * Hin * Win * ld + cb + 8 * h;
        const uint4* ph = p.whi + ((size_t)nf * p.nchunks + c) * (KK * 2 * 64) + lane;
        const uint4* pl = p.wlo + ((size_t)nf * p.nchunks + c) * (KK * 2 * 64) + lane;
        const uint4* p2 = SIX ? p.wl2 + ((size_t)nf * p.nchunks + c) * (KK * 2 * 64) + lane : nullptr;
        const size_t tstride = (size_t)p.nchunks * (KK * 2 * 64);                 // next output-channel tile
#pragma unroll
        for (int tap = 0; tap < KK; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            if (MODE == 2) {
                const int ty = ybase + PAD - ky, tx = xpar + PAD - kx;
                const int yi = ty >> 1;
                if ((ty & 1) || (tx & 1) || yi < 0 || yi >= Hin) continue;          // wave-uniform: 5 to 8 of 9 taps are inactive by parity
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                bf16x8 bh[NCO], bl[NCO], b2[NCO];
#pragma unroll
                for (int t = 0; t < NCO; ++t) {
                    bh[t] = __builtin_bit_cast(bf16x8, ph[t * tstride + (tap * 2 + k) * 64]);
                    bl[t] = __builtin_bit_cast(bf16x8, pl[t * tstride + (tap * 2 + k) * 64]);
                    if constexpr (SIX) b2[t] = __builtin_bit_cast(bf16x8, p2[t * tstride + (tap * 2 + k) * 64]);
                }
                bf16x8 ah[R], al[R], a2[R];
#pragma unroll
                for (int rr = 0; rr < R; ++rr) {
                    const int y = ybase + rr;
                    int yi, xi;
                    if (MODE == 1) { yi = 2 * y + ky - PAD; xi = 2 * (x0 + i) + kx - PAD; }
                    else { yi = (y + PAD - ky) >> 1; xi = x0 + i + ((xpar + PAD - kx) >> 1); }
                    // out-of-range rows / columns / channels are folded into the load predicate (zero operands): branch-free
                    const bool ok = lane_in && y < Hout && xi >= 0 && xi < Win && yi >= 0 && yi < Hin && cb + 16 * k + 8 * h < Cs;
                    // unconditional loads from a clamped (always valid) address + select: no exec-masked branch per load, so
                    // the loads of all taps can be in flight together
                    const float* q = ok ? inb + ((size_t)yi * Win + xi) * ld + 16 * k : src;
                    float4 a0 = *(const float4*)q, a1 = *(const float4*)(q + 4);
                    if (!ok) { a0 = make_float4(0.f, 0.f, 0.f, 0.f); a1 = a0; }
                    uint4 hi, lo;
                    if constexpr (SIX) {
                        uint4 l2;
                        gsplit3(a0.x, a0.y, hi.x, lo.x, l2.x); gsplit3(a0.z, a0.w, hi.y, lo.y, l2.y);
                        gsplit3(a1.x, a1.y, hi.z, lo.z, l2.z); gsplit3(a1.z, a1.w, hi.w, lo.w, l2.w);
                        a2[rr] = __builtin_bit_cast(bf16x8, l2);
                    } else {
                        gsplit2(a0.x, a0.y, hi.x, lo.x); gsplit2(a0.z, a0.w, hi.y, lo.y);
                        gsplit2(a1.x, a1.y, hi.z, lo.z); gsplit2(a1.z, a1.w, hi.w, lo.w);
                    }
                    ah[rr] = __builtin_bit_cast(bf16x8, hi); al[rr] = __builtin_bit_cast(bf16x8, lo);
                }
                if constexpr (SIX) if (six) {
#pragma unroll
                    for (int rr = 0; rr < R; ++rr)
#pragma unroll
                        for (int t = 0; t < NCO; ++t) {
                            acc[t][rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[rr], bh[t], acc[t][rr], 0, 0, 0);
                            acc[t][rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rr], b2[t], acc[t][rr], 0, 0, 0);
                            acc[t][rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rr], bl[t], acc[t][rr], 0, 0, 0);
                        }
                }
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
#pragma unroll
                    for (int t = 0; t < NCO; ++t) {
                        if (!x1) acc[t][rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rr], bh[t], acc[t][rr], 0, 0, 0);
                        if (!x1 || p.x1_w2) acc[t][rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rr], bl[t], acc[t][rr], 0, 0, 0);
                        acc[t][rr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rr], bh[t], acc[t][rr], 0, 0, 0);
                    }
            }
        }
    }
    if constexpr (KSPLIT == 4) {
        __shared__ float red[3][NCO][R][16][64];
        if (wave) {
#pragma unroll
            for (int t = 0; t < NCO; ++t)
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - 1][t][rr][r][lane] = acc[t][rr][r];
        }
        __syncthreads();
        if (wave) return;
#pragma unroll
        for (int t = 0; t < NCO; ++t)
#pragma unroll
        for (int rr = 0; rr < R; ++rr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][rr][r] = (acc[t][rr][r] + red[0][t][rr][r][lane]) + (red[1][t][rr][r][lane] + red[2][t][rr][r][lane]);
    }
    auto activate = [&](float v) {
        if (p.act == GACT_RELU) v = v > 0.f ? v : 0.f;
        else if (p.act == GACT_LRELU) v = v > 0.f ? v : 0.2f * v;
        else if (p.act == GACT_SIGMOID) v = 1.f / (1.f + expf(-v));
        return v;
    };
    auto activate_cheap = [&](float v) {                 // selects only; the sigmoid takes the scalar path below
        const float neg = p.act == GACT_LRELU ? 0.2f * v : (p.act == GACT_RELU ? 0.f : v);
        return v > 0.f ? v : neg;
    };
    const bool wide = p.act != GACT_SIGMOID && !(p.ldy & 3) && !(p.Cy & 3) && !(((size_t)p.y) & 15);
#pragma unroll
    for (int t = 0; t < NCO; ++t) {
    const int co = (nf + t) * 32 + i - p.nf0 * 32;
    const float bias = (p.bias && co < p.Cy) ? p.bias[co] : 0.f;
    if (wide) {
        // 16 B per lane in the quad-transposed layout (ptta_common.h quad_transpose): a quarter of the store instructions
        const int c4 = (nf + t) * 32 - p.nf0 * 32 + 4 * (i >> 2);
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int y = ybase + rr;
            if (y >= Hout) break;
            f32x16 tt;
#pragma unroll
            for (int r = 0; r < 16; ++r) tt[r] = acc[t][rr][r] + bias;
            quad_transpose(tt, lane);
            float* yrow = p.y + ((size_t)b * Hout + y) * Wout * p.ldy + c4;
            bool ok[4]; size_t xo[4]; float4 old[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int xl = x0 + (i & 3) + 8 * g + 4 * h;
                const int x = MODE == 2 ? 2 * xl + xpar : xl;
                ok[g] = c4 < p.Cy && xl < Wt && x < Wout;
                xo[g] = (size_t)(ok[g] ? x : 0) * p.ldy;
            }
            if (p.accumulate) {
#pragma unroll
                for (int g = 0; g < 4; ++g) old[g] = ok[g] ? *(const float4*)(yrow + xo[g]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = make_float4(tt[4 * g], tt[4 * g + 1], tt[4 * g + 2], tt[4 * g + 3]);
                if (p.accumulate) { v.x += old[g].x; v.y += old[g].y; v.z += old[g].z; v.w += old[g].w; }
                v.x = activate_cheap(v.x); v.y = activate_cheap(v.y); v.z = activate_cheap(v.z); v.w = activate_cheap(v.w);
                if (ok[g]) *(float4*)(yrow + xo[g]) = v;
            }
        }
        continue;
    }
    if (co >= p.Cy) continue;
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int y = ybase + rr;
        if (y >= Hout) break;
        float* yrow = p.y + ((size_t)b * Hout + y) * Wout * p.ldy + co;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int xl = x0 + acc_row(r, h);
            if (xl >= Wt) continue;
            const int x = MODE == 2 ? 2 * xl + xpar : xl;
            if (MODE == 2 && x >= Wout) continue;         // odd output width: the gradient of a stride-2 conv over an odd-sized map
            float* dst = yrow + (size_t)x * p.ldy;
            float v = acc[t][rr][r] + bias;
            if (p.accumulate) v += *dst;
            *dst = activate(v);
        }
    }
    }
}

// ---- weight gradient of a stride-1 3x3 convolution with up to 64 x 64 channels (the adapted conv1_rgb_meta,
// Conv2d(48,48,3,1,1), nlspnmodel_adapt.py:1371) on the fp32 matrix cores: dW[co][ci][tap] = sum_p gy[p][co] x[p+tap][ci].
// One wave per (pixel chunk, 32x32 channel-block pair): nine 32x32 accumulators (one per tap) + one for the bias,
// K = pixels, two pixels per v_mfma_f32_32x32x2_f32; a second fixed-order pass sums the chunk partials.
#define GWG_PART (10 * 1024)
__global__ __launch_bounds__(64) void gwgrad_mfma_kernel(GView x, GView gy, int nchunks, int ncib, float* __restrict__ part) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int H = x.H, W = x.W;
    const long P = (long)x.B * H * W;
    const long per = ((P + nchunks - 1) / nchunks + 7) & ~7L;
    const long p0 = (long)blockIdx.x * per;
    long p1 = p0 + per; if (p1 > P) p1 = P;
    const int cob = blockIdx.y / ncib, cib = blockIdx.y % ncib;
    const int co = cob * 32 + i, ci = cib * 32 + i;
    const bool cov = co < gy.C, civ = ci < x.C;
    f32x16 acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (long pp = p0; pp < p1; pp += 8) {
        float a[4], bv[4][9];
        bool pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long pix = pp + 2 * u + h;
            pv[u] = pix < p1;
            int px = 0, py = 0;
            a[u] = 0.f;
            if (pv[u]) {
                px = (int)(pix % W); py = (int)((pix / W) % H);
                if (cov) a[u] = gy.p[pix * gy.ld + co];
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
                bv[u][tap] = 0.f;
                if (pv[u] && civ && yy >= 0 && yy < H && xx >= 0 && xx < W) bv[u][tap] = x.p[(pix + (long)(tap / 3 - 1) * W + (tap % 3 - 1)) * x.ld + ci];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u][tap], acc[tap], 0, 0, 0);
            acc[9] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], pv[u] ? 1.f : 0.f, acc[9], 0, 0, 0);
        }
    }
    float* out = part + ((long)blockIdx.x * gridDim.y + blockIdx.y) * GWG_PART;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(tap * 32 + acc_row(r, h)) * 32 + i] = acc[tap][r];
    if (i == 0)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[9 * 1024 + acc_row(r, h)] = acc[9][r];
}

// ---- the same weight gradient on the bf16 matrix cores (bf16x3), for Ci, Co <= 64: dW[co][ci][tap] = sum_p gy[p][co] x[p+tap][ci] as a
// reduction-GEMM over pixels, like head_train.hip's Linear gradient: an MFMA lane holds 8 consecutive k for one m, and with
// k = pixel, m = channel those are 8 consecutive PIXELS of one channel -- 8 dword loads per fragment, straight from global.
// Block = 4 waves = the (co tile, ci tile) pairs of a 64 x 64 problem; a K step is 16 consecutive pixels of one image row.
// Per row offset dy the lane loads x[row + dy][p - 1 .. p + 8] ONCE (10 loads) and the three horizontal taps are register
// renames of it (v[0..7], v[1..8], v[2..9]): 38 loads per step instead of 80.  Nine 32 x 32 accumulators per wave; the next
// step's loads are issued before the current step's 27 MFMAs.  Partials in the layout of gwgrad_mfma_kernel (same reducer).
// 352x1216, Conv2d(48,48,3): 876 -> see DESIGN.md (the fp32 form alternated a 40-load phase and a 40-MFMA phase per 8 pixels).
__device__ __forceinline__ void wsplit8(const float* v, uint4& hi, uint4& lo) {
    gsplit2(v[0], v[1], hi.x, lo.x); gsplit2(v[2], v[3], hi.y, lo.y);
    gsplit2(v[4], v[5], hi.z, lo.z); gsplit2(v[6], v[7], hi.w, lo.w);
}
// single = 1 (Ci, Co <= 32): one tile pair; the four waves take different items instead and write a partial each.
// BLKRED (single only): the four waves' partial tiles are summed inside the block in a fixed tree ((w0 + w2) + (w1 + w3)) through LDS and the
// block writes ONE partial -- four times the waves for the same partial traffic: the launch is a latency chain of ~2 us per item and wave
// (38 strided loads, then 27 MFMAs), so its duration is the items per wave.
template <bool GY16, bool BLKRED = false>          // GY16: gy.p points at bf16 values (the narrow gradient maps of the MSG_CHN mixed mode): widened on load, the same products
__global__ __launch_bounds__(256) void gwgrad_x3_kernel(GView x, GView gy, int nitems_x, int single, int ncib, float* __restrict__ part) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, hg = lane >> 5;
    // the four waves of a block = the four (ci tile, co tile) pairs: 2 x 2 (<= 64 x 64 channels), 1 x 4 (32 -> 128) or 4 x 1 (128 -> 32)
    const int cob = single ? 0 : wave / ncib, cib = single ? 0 : wave % ncib;
    const int H = x.H, W = x.W;
    const int co = cob * 32 + c, ci = cib * 32 + c;
    const bool cov = co < gy.C, civ = ci < x.C;
    // Addresses = uniform base + 32-bit BYTE offset (maps < 4 GB, checked by the launcher).  The item is wave-uniform (scalar row / image
    // arithmetic), a lane adds its channel and its half's eight pixels; clamps are selects between precomputed offsets.  The first form --
    // 64-bit element indices from a lane-varying item, clamp, multiply by the row pitch per load -- spent ~190 quarter-rate integer
    // multiplies per item: 3 k of the ~4.5 k cycles an item took (round 5, .s).
    constexpr unsigned GE = GY16 ? 2u : 4u;
    const unsigned gld = (unsigned)gy.ld * GE, xld = (unsigned)x.ld * 4u;
    const unsigned glane = (unsigned)(cov ? co : 0) * GE + (hg ? 8u * gld : 0u), xlane = (unsigned)(civ ? ci : 0) * 4u + (hg ? 8u * xld : 0u);
    const char* const gbase = (const char*)gy.p;
    const char* const xbase = (const char*)x.p;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const int nitems = x.B * H * nitems_x;                            // item = 16 consecutive pixels of one row
    struct Regs { float g[8]; float v[3][10]; unsigned gm, vm[3]; };      // raw loads + validity bits (zero-fill happens at use: no wait at the load)
    auto fetch = [&](int item, Regs& r) __attribute__((always_inline)) {
        const int xs = item % nitems_x; const int t_ = item / nitems_x;
        const int y = t_ % H; const int b = t_ / H;
        const int p0 = xs * 16 + 8 * hg;                               // this lane's first pixel of the step
        const unsigned rowpix = (unsigned)((b * H + y) * W);
        const unsigned g0 = (rowpix + (unsigned)(xs * 16)) * gld + glane;
        const unsigned glast = (rowpix + (unsigned)(W - 1)) * gld + (unsigned)(cov ? co : 0) * GE;
        r.gm = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = p0 + j < W;
            const unsigned off = ok ? g0 + (unsigned)j * gld : glast;
            if constexpr (GY16) r.g[j] = __uint_as_float((unsigned)*(const bf16_t*)(gbase + off) << 16);
            else r.g[j] = *(const float*)(gbase + off);
            r.gm |= (ok && cov) ? (1u << j) : 0u;
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const bool rok = yy >= 0 && yy < H;
            const unsigned rp = (unsigned)((b * H + min(max(yy, 0), H - 1)) * W);
            const unsigned x0 = (rp + (unsigned)(xs * 16)) * xld + xlane;          // pixel p0 of the row
            const unsigned xfirst = rp * xld + (unsigned)(civ ? ci : 0) * 4u, xlast = (rp + (unsigned)(W - 1)) * xld + (unsigned)(civ ? ci : 0) * 4u;
            r.vm[dy] = 0;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const int px = p0 + j - 1;
                const bool lo_ok = px >= 0, hi_ok = px < W;
                const unsigned off = !lo_ok ? xfirst : (!hi_ok ? xlast : x0 + (unsigned)(j - 1) * xld);
                r.v[dy][j] = *(const float*)(xbase + off);
                r.vm[dy] |= (rok && civ && lo_ok && hi_ok) ? (1u << j) : 0u;
            }
        }
    };
    auto mma = [&](const Regs& rr_) __attribute__((always_inline)) {
        Regs r = rr_;
#pragma unroll
        for (int j = 0; j < 8; ++j) if (!((r.gm >> j) & 1u)) r.g[j] = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int j = 0; j < 10; ++j) if (!((r.vm[dy] >> j) & 1u)) r.v[dy][j] = 0.f;
        uint4 gh, gl;
        wsplit8(r.g, gh, gl);
        const bf16x8 ah = __builtin_bit_cast(bf16x8, gh), al = __builtin_bit_cast(bf16x8, gl);
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum += r.g[j];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            // the row's ten values are split ONCE: packed hi / lo of the five even-aligned pairs (0,1) .. (8,9) and the four odd-aligned pairs
            // (1,2) .. (7,8); tap dx = 0 / 2 takes even pairs 0..3 / 1..4, dx = 1 the odd ones.  Same conversions as wsplit8 of the three
            // shifted windows (hi = RNE(v), lo = RNE(v - hi)): identical fragments, 38 instead of 72 conversions.
            const float* v = r.v[dy];
            unsigned eh[5], el[5], oh[4], ol[4];
            float res[10];
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                float2_t pr = {v[2 * k], v[2 * k + 1]};
                eh[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(pr, bf16x2_t));
                res[2 * k] = v[2 * k] - __uint_as_float(eh[k] << 16);
                res[2 * k + 1] = v[2 * k + 1] - __uint_as_float(eh[k] & 0xffff0000u);
                float2_t rr2 = {res[2 * k], res[2 * k + 1]};
                el[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr2, bf16x2_t));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // hi halves of values 2k+1, 2k+2 are already in eh[k] (upper half) and eh[k+1] (lower half)
                oh[k] = (eh[k] >> 16) | (eh[k + 1] << 16);
                float2_t rr2 = {res[2 * k + 1], res[2 * k + 2]};
                ol[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(rr2, bf16x2_t));
            }
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const uint4 xh = dx == 1 ? make_uint4(oh[0], oh[1], oh[2], oh[3]) : (dx == 0 ? make_uint4(eh[0], eh[1], eh[2], eh[3]) : make_uint4(eh[1], eh[2], eh[3], eh[4]));
                const uint4 xl = dx == 1 ? make_uint4(ol[0], ol[1], ol[2], ol[3]) : (dx == 0 ? make_uint4(el[0], el[1], el[2], el[3]) : make_uint4(el[1], el[2], el[3], el[4]));
                const bf16x8 bh = __builtin_bit_cast(bf16x8, xh), bl = __builtin_bit_cast(bf16x8, xl);
                const int t = dy * 3 + dx;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    };
    Regs r0, r1;
    const int stride = single ? (int)gridDim.x * 4 : (int)gridDim.x;
    int item = single ? (int)blockIdx.x * 4 + wave : (int)blockIdx.x;
    if (item < nitems) fetch(item, r0);
    while (item < nitems) {
        const int n1 = item + stride;
        if (n1 < nitems) fetch(n1, r1);
        mma(r0);
        if (n1 >= nitems) break;
        const int n2 = n1 + stride;
        if (n2 < nitems) fetch(n2, r0);
        mma(r1);
        item = n2;
    }
    bsum += __shfl_xor(bsum, 32);
    if constexpr (BLKRED) {
        __shared__ float red[2][9 * 1024 + 32];
        auto put = [&](float* dst) __attribute__((always_inline)) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(tap * 32 + acc_row(r, hg)) * 32 + c] = acc[tap][r];
            if (hg == 0) dst[9 * 1024 + c] = bsum;
        };
        auto add = [&](const float* src) __attribute__((always_inline)) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tap][r] += src[(tap * 32 + acc_row(r, hg)) * 32 + c];
            bsum += src[9 * 1024 + c];
        };
        if (wave >= 2) put(red[wave - 2]);
        __syncthreads();
        if (wave < 2) add(red[wave]);                                  // w0 += w2, w1 += w3
        __syncthreads();
        if (wave == 1) put(red[0]);
        __syncthreads();
        if (wave != 0) return;
        add(red[0]);
        put(part + (long)blockIdx.x * GWG_PART);
        return;
    }
    float* out = part + ((long)blockIdx.x * 4 + wave) * GWG_PART;      // pair index = cob * ncib + cib = wave; single: chunk index
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[(tap * 32 + acc_row(r, hg)) * 32 + c] = acc[tap][r];
    if (hg == 0) out[9 * 1024 + c] = bsum;                            // bias partial of channel co (read from the pairs with cib == 0)
}

// Fixed-order second stage.  A block owns 64 consecutive elements; thread (e = tid & 63, g = tid >> 6) sums the chunks g, g + 16, ... of
// its element in fp64 -- the 64 lanes of a wave read 64 consecutive floats of one partial tile (ci is the fastest index of both the
// element order and the tile) -- eight loads in flight per thread; the sixteen group sums are combined in a fixed order through LDS.
// (Four groups with four loads in flight: 16 dependent rounds over 256 partials, 9 us; this form: two.)
#define GWR_G 16
// `ad` (ad.pw != null; MSG_CHN 1layer fused step without a gradient exchange): the thread that finishes an element's sum applies Adam to it at
// once -- the arithmetic of adam_multi_kernel (wgrad_adam.hip) on the value just stored in gw / gb, t = *step + 1 read by every block, the step
// count written by the last block to finish: the same parameters and moments bit for bit, one dependent launch less at the very end of the step.
__global__ __launch_bounds__(64 * GWR_G) void gwgrad_mfma_reduce_kernel(const float* __restrict__ part, int nchunks, int npairs, int ncib, int Ci, int Co,
                                                                       float* __restrict__ gw, float* __restrict__ gb, GwAdam ad) {
    __shared__ double red[GWR_G][64];
    const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + el;
    const long nw = 9L * Co * Ci;
    const bool live = e < nw + Co;
    int tap = 0, co = 0, ci = 0;
    if (live) {
        if (e < nw) { tap = (int)(e / ((long)Co * Ci)); const int r = (int)(e % ((long)Co * Ci)); co = r / Ci; ci = r % Ci; }
        else co = (int)(e - nw);
    }
    const int pair = (co >> 5) * ncib + (e < nw ? (ci >> 5) : 0);
    const long off = e < nw ? ((long)tap * 32 + (co & 31)) * 32 + (ci & 31) : 9 * 1024 + (co & 31);
    const float* p0 = part + (long)pair * GWG_PART + off;
    const long cs = (long)npairs * GWG_PART;                   // floats between consecutive chunks of one element
    double s = 0.0;
    if (live) {
        int k = g;
        for (; k + 7 * GWR_G < nchunks; k += 8 * GWR_G) {
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = p0[(long)(k + j * GWR_G) * cs];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (double)a[j];
        }
        for (; k < nchunks; k += GWR_G) s += (double)p0[(long)k * cs];
    }
    red[g][el] = s;
    __syncthreads();
    if (g == 0 && live) {
        s = 0.0;
#pragma unroll
        for (int j = 0; j < GWR_G; ++j) s += red[j][el];
        const float gv = (float)s;
        const bool isw = e < nw;
        if (isw) gw[((long)co * Ci + ci) * 9 + tap] = gv;
        else if (gb) gb[co] = gv;
        if (ad.pw && (isw || ad.pb)) {
            const float lr = ad.hyper[0], b1 = ad.hyper[1], b2 = ad.hyper[2], eps = ad.hyper[3], wd = ad.hyper[4];
            const int t = *ad.step + 1;
            const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
            const float step_size = (float)((double)lr / bc1);
            const float bc2s = (float)sqrt(bc2);
            const long k = isw ? ((long)co * Ci + ci) * 9 + tap : co;
            float* P = isw ? ad.pw : ad.pb; float* M = isw ? ad.mw : ad.mb; float* V = isw ? ad.vw : ad.vb;
            float pk = P[k], mk = M[k], vk = V[k];
            ptta_adam_update(pk, mk, vk, gv, wd, b1, b2, eps, step_size, bc2s);
            M[k] = mk; V[k] = vk; P[k] = pk;
        }
    }
    if (ad.pw) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned done = atomicAdd(ad.ticket, 1u);
            if (done == gridDim.x - 1) { *ad.step = *ad.step + 1; *ad.ticket = 0u; }
        }
    }
}

}  // namespace

void ptta_gfrag_pack(const float* canon, long wld, long wts, int KK, int C0, int C1, int c0_0, int c0_1, int Co, bf16_t* hi, bf16_t* lo,
                     hipStream_t s, bf16_t* l2) {
    const int nchunks = (C0 + 31) / 32 + (C1 + 31) / 32, nf_total = (Co + 31) / 32;
    const long total = (long)nf_total * nchunks * KK * 2 * 64 * 8;
    long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gfrag_pack_kernel, dim3((int)blocks), dim3(256), 0, s, canon, wld, wts, KK, C0, C1, c0_0, c0_1, Co, nchunks, nf_total, hi, lo, l2);
}
long ptta_gfrag_elems(int KK, int C0, int C1, int Co) {
    return (long)((Co + 31) / 32) * ((C0 + 31) / 32 + (C1 + 31) / 32) * KK * 2 * 64 * 8;
}

int ptta_gconv_x3_tiles(int B, int H, int W) { return B * ((W + 31) / 32) * ((H + GX_TH - 1) / GX_TH); }

int ptta_launch_gconv_x3(const GX3Args& a, int ks, hipStream_t s) {
    if ((a.C0 & 7) || (a.C1 & 7) || (a.ld0 & 3) || (a.ld1 & 3)) return -22;        // channel counts: multiples of 8
    if (a.nchunks != (a.C0 + 31) / 32 + (a.C1 + 31) / 32) return -22;
    const long tiles = (long)a.B * ((a.W + 31) / 32) * ((a.H + GX_TH - 1) / GX_TH);
    const long blocks = tiles * a.nnf;
    if (blocks < 1 || blocks > 0x7fffffffL) return -22;
    if (a.six_B > 0) {
        if (!a.wl2) return -22;
        if (ks == 3 && a.vert) hipLaunchKernelGGL((gconv_x3_s1_kernel<3, true, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        else if (ks == 3) hipLaunchKernelGGL((gconv_x3_s1_kernel<3, false, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        else if (ks == 1) hipLaunchKernelGGL((gconv_x3_s1_kernel<1, false, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
        else return -22;
        PTTA_CHECK_LAUNCH();
        return 0;
    }
    if (ks == 3 && a.vert) hipLaunchKernelGGL((gconv_x3_s1_kernel<3, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
#ifdef PTTA_DIAG_STAMPS      // diagnostic build (make DIAG=1): in-kernel phase stamps of the 3x3 launches
    else if (ks == 3) hipLaunchKernelGGL((gconv_x3_s1_kernel<3, false, true>), dim3((unsigned)blocks), dim3(256), 0, s, a);
#endif
    else if (ks == 3) hipLaunchKernelGGL((gconv_x3_s1_kernel<3, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else if (ks == 1) hipLaunchKernelGGL((gconv_x3_s1_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else return -22;
    PTTA_CHECK_LAUNCH();
    return 0;
}

// mode 1: stride-2 convolution (a.H/a.W = output size, input = hin x win); mode 2: stride-2 transposed convolution
int ptta_launch_gconv_x3_strided(const GX3Args& a, int ks, int mode, int hin, int win, hipStream_t s) {
    if ((a.C0 & 15) || (a.C1 & 15) || (a.ld0 & 3) || (a.ld1 & 3) || (ks != 1 && ks != 3) || (mode != 1 && mode != 2)) return -22;
    if (a.nchunks != (a.C0 + 31) / 32 + (a.C1 + 31) / 32) return -22;
    const int Wt = mode == 2 ? win : a.W;
    const long nitems = (long)a.B * ((Wt + 31) / 32) * (mode == 2 ? 2 : 1) * (mode == 1 ? (a.H + 1) / 2 : a.H);   // stride 2: two rows per wave
    // few pixels and a long K loop: split K over the block's waves
    // few pixels and a long K loop (the 1/8 and 1/16-resolution layers): one item per block, K split over its waves, one channel
    // tile per wave; otherwise two channel tiles per wave
    const bool split = a.nchunks >= 4 && ((nitems + 3) / 4) * a.nnf < 1536;
    // (measured on the NLSPN step, same box: one tile 24.83 ms, two 24.36, four 24.47)
    const int nco = (split || ks != 3 || (a.nnf & 1)) ? 1 : 2;
    const int ngrp = a.nnf / nco;
    const long blocks = (split ? nitems : (nitems + 3) / 4) * ngrp;
    if (blocks < 1 || blocks > 0x7fffffffL) return -22;
    if (a.six_B > 0 && !a.wl2) return -22;
#define L__(M, K, S) do { if (nco == 2) hipLaunchKernelGGL((gconv_x3_direct_kernel<M, K, 1, 2, S>), dim3((unsigned)blocks), dim3(256), 0, s, a, hin, win); \
                      else if (split) hipLaunchKernelGGL((gconv_x3_direct_kernel<M, K, 4, 1, S>), dim3((unsigned)blocks), dim3(256), 0, s, a, hin, win); \
                      else hipLaunchKernelGGL((gconv_x3_direct_kernel<M, K, 1, 1, S>), dim3((unsigned)blocks), dim3(256), 0, s, a, hin, win); } while (0)
#define L_(M, K) do { if (a.six_B > 0) L__(M, K, true); else L__(M, K, false); } while (0)
    if (mode == 1) { if (ks == 3) L_(1, 3); else L_(1, 1); }
    else { if (ks == 3) L_(2, 3); else L_(2, 1); }
#undef L_
#undef L__
    PTTA_CHECK_LAUNCH();
    return 0;
}

int ptta_gwgrad_mfma_chunks(long pixels) { long n = (pixels + 127) / 128; return (int)(n > 1024 ? 1024 : (n < 1 ? 1 : n)); }
long ptta_gwgrad_mfma_part_floats(long pixels, int Ci, int Co) {
    const long a = (long)ptta_gwgrad_mfma_chunks(pixels) * ((Ci + 31) / 32) * ((Co + 31) / 32) * GWG_PART;
    const long b = 256L * 4 * GWG_PART;                                  // gwgrad_x3_kernel: 256 blocks x 4 tile pairs
    return a > b ? a : b;
}
int ptta_launch_gwgrad_mfma(const GView& x, const GView& gy, float* part, float* gw, float* gb, hipStream_t s, int gy_bf16, const GwAdam* adam) {
    const GwAdam noad{};
    const bool sq = x.C <= 64 && gy.C <= 64, wide_out = x.C <= 32 && gy.C <= 128, wide_in = x.C <= 128 && gy.C <= 32;
    // (the bf16x3 kernel addresses its maps with 32-bit byte offsets)
    const bool small_maps = (double)x.B * x.H * x.W * x.ld * 4.0 < 4294967296.0 && (double)gy.B * gy.H * gy.W * gy.ld * 4.0 < 4294967296.0;
    if ((sq || wide_out || wide_in) && x.C > 16 && small_maps) {
        // bf16x3 form: blocks of four waves = four 32x32 channel-tile pairs (2 x 2, or 1 x 4 / 4 x 1 for the 32 <-> 128 layers of the
        // 2layers meta block), partials [block][4 pairs]; <= 32 x 32 channels: one pair, the waves split the items, partials
        // [block * 4 + wave]
        const int nx = (x.W + 15) / 16;
        const long nitems = (long)x.B * x.H * nx;
        const int single = (x.C <= 32 && gy.C <= 32) ? 1 : 0;
        const int kcib = sq ? 2 : (wide_out ? 1 : 4);
        const long cap = 256;       // (single: measured 64 / 128 / 256 / 336 / 418 blocks on the MSG_CHN step: 1.213 / 1.207 / 1.201 / 1.208 / 1.209 ms)
        const long want = single ? (nitems + 3) / 4 : nitems;
        const int nblk = (int)(want < cap ? want : cap);
        const long n = 9L * x.C * gy.C + gy.C;
        if (single) {
            // one partial per BLOCK (summed across its waves in LDS)
            if (gy_bf16) hipLaunchKernelGGL((gwgrad_x3_kernel<true, true>), dim3(nblk), dim3(256), 0, s, x, gy, nx, single, kcib, part);
            else hipLaunchKernelGGL((gwgrad_x3_kernel<false, true>), dim3(nblk), dim3(256), 0, s, x, gy, nx, single, kcib, part);
            hipLaunchKernelGGL(gwgrad_mfma_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * GWR_G), 0, s, part, nblk, 1, 1, x.C, gy.C, gw, gb, adam ? *adam : noad);
            PTTA_CHECK_LAUNCH();
            return 0;
        }
        if (gy_bf16) hipLaunchKernelGGL(gwgrad_x3_kernel<true>, dim3(nblk), dim3(256), 0, s, x, gy, nx, single, kcib, part);
        else hipLaunchKernelGGL(gwgrad_x3_kernel<false>, dim3(nblk), dim3(256), 0, s, x, gy, nx, single, kcib, part);
        if (single) hipLaunchKernelGGL(gwgrad_mfma_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * GWR_G), 0, s, part, nblk * 4, 1, 1, x.C, gy.C, gw, gb, noad);
        else hipLaunchKernelGGL(gwgrad_mfma_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * GWR_G), 0, s, part, nblk, 4, kcib, x.C, gy.C, gw, gb, noad);
        PTTA_CHECK_LAUNCH();
        return 0;
    }
    if (gy_bf16) return -22;          // (the fp32-MFMA form takes fp32 gradients)
    const int nchunks = ptta_gwgrad_mfma_chunks((long)x.B * x.H * x.W);
    const int ncib = (x.C + 31) / 32, ncob = (gy.C + 31) / 32, npairs = ncib * ncob;
    hipLaunchKernelGGL(gwgrad_mfma_kernel, dim3(nchunks, npairs), dim3(64), 0, s, x, gy, nchunks, ncib, part);
    const long n = 9L * x.C * gy.C + gy.C;
    hipLaunchKernelGGL(gwgrad_mfma_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * GWR_G), 0, s, part, nchunks, npairs, ncib, x.C, gy.C, gw, gb, noad);
    PTTA_CHECK_LAUNCH();
    return 0;
}
