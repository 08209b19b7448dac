// The projection / prediction heads of the MIXED mode (include/ptta.h PTTA_DTYPE_MIXED): the embedding branch of ProxyTTA is computed under
// no_grad / detached in the reference (network_exp_msg_chn_adapt.py:551-554) and the reference branch enters the scored depth only through
// the cosine term's lr-sized Adam move, so every tensor here is NARROW -- bf16 storage, ONE bf16 MFMA per product, fp32 accumulate -- and
// BatchNorm statistics, biases, masks and the row / column reductions stay fp32.  (PTTA_DTYPE_F32 keeps heads.hip: fp32 storage, bf16x3.)
//
// One kernel shape for the four 26752 x 512 x 512 GEMMs of a step: block = 128 FULL rows x 512 columns (209 blocks at 352x1216: one
// round on 256 CUs), 8 waves, each 128 rows x 64 columns (8 accumulator tiles), K = 512 in 16 slices of 32 through two LDS stages
// (A 128 x 64 B, B 512 x 64 B, 80-B row stride: conflict-free ds_read_b128), next slice's global loads in flight during the MFMAs.
// Full rows per block mean the prologues / epilogues need no second pass:
//   PRO 1  A = relu(h * scale + shift)                   (BatchNorm1d + ReLU of the previous layer, applied while staged)
//   PRO 3  A = relu(bn(x W0^T + b0))                      (proj's 512-wide hidden computed from the 32-channel feature row on the matrix
//                                                          cores by waves 0-3, never materialised: as heads.hip PRO 3)
//   PRO 4  A = d L_cos / d ref                            (from emb, ref and the loss kernel's per-row statistics: as heads.hip PRO 4)
//   PRO 5  A = an fp32 [R][512] tensor, rounded while staged (ptta_backward with a caller's d L / d ref)
// Layouts chosen for this kernel (everything it touches is private to the heads):
//   * weights SLICE-major, Wsl[k / 32][n][k % 32] bf16: the 32-KB B operand of a K slice is one contiguous block (row-major made every
//     lane group fetch 64-B pieces at a 1-KB stride -- the same few L2 channels from every CU at once: 60 us per GEMM instead of ~15);
//   * activations TILED, T[r / 128][k / 32][r % 128][k % 32] bf16: the 8-KB A operand of a (block, slice) is contiguous for the consumer and
//     the producer's transposed epilogue writes 8 rows x 64 B per instruction.  hn_untile_kernel gives the row-major fp32 view to callers.
//   EPI 0  C = acc + bias -> narrow store
//   EPI 1  the same + per-block column sums (sum, sum of squares of the fp32 values) for the next BatchNorm1d's batch statistics
//   EPI 2  the same as EPI 0 + per-row sum of squares of the stored values -> rs[r]              (emb: |e|^2)
//   EPI 5  the same + per-row |c|^2 and c . e against the tensor E of EPI 2 -> rowstats[r] = (|e|, |c|, cos) and the block's partial of
//          sum_r (2 - 2 cos) -- the cosine term's row pass (loss.hip cos_rows_kernel) without a second read of emb / ref
//   EPI 3  data gradient through proj.3 and proj.0: hidden recomputed in the accumulator layout for the ReLU mask and the two
//          BatchNorm-backward column sums; the masked gradient x gamma x invstd goes to an LDS plane [128][512] and is contracted with
//          W0 inside the block -> P [R][32] (heads.hip EPI 3 with both column halves in one block)
#include "ptta_common.h"
#include "ptta_kernels.h"

#define HN_BM 128
#define HN_ROW 80            // bytes per LDS row of a K slice: 32 bf16 + 16 pad
#define HN_DY 1040           // bytes per row of the EPI 3 gradient plane: 512 bf16 + 16 pad (conflict-free ds_read_b128: 260 dwords = 4 mod 64 ... x4)

struct HnP {
    const float* Af;
    const bf16_t* A; const bf16_t* Bref; const float* rowstats; const float* coef;
    const void* X; int x_bf16;
    const bf16_t* W0; const float* b0; const float* pscale; const float* pshift;
    const bf16_t* W; const float* bias;
    bf16_t* C; float* part;
    const bf16_t* E; float* rs; float* rowstats_out; float* cpart; int cpart_n;
    const float *escale, *eshift, *emean, *einv; const bf16_t* W0t; float* P;
    long R;
};

__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u); v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u); v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    return make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}
// 8 consecutive channels of one feature row as bf16 (fp32 features are rounded here: the heads are a single-MFMA class)
__device__ __forceinline__ uint4 load_x8(const void* X, int x_bf16, long row, int ch0) {
    if (x_bf16) return *(const uint4*)((const bf16_t*)X + row * 32 + ch0);
    const float* s = (const float*)X + row * 32 + ch0;
    const float4 a = *(const float4*)s, b = *(const float4*)(s + 4);
    return make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
}

// element (row r, column k) of a tiled [R][512] tensor
__device__ __forceinline__ size_t hn_tiled(long r, int k) { return ((size_t)(r >> 7) * 16 + (k >> 5)) * 4096 + (size_t)(r & 127) * 32 + (k & 31); }

// ds_read_b128 the compiler does not see.  The fragment reads of the K loop come from LDS that LDS-DMA (global_load_lds) fills; hipcc cannot
// tell which ring stage a ds_read touches and puts s_waitcnt vmcnt(0) in front of the first LDS read behind ANY global_load_lds -- i.e. it
// awaited the slice it had just requested before every matrix phase (ISA inspected: one vmcnt(0) per K slice, 43 -> 63 us per GEMM).  These
// reads are ordered by the kernel's own counted waits and barriers instead; lds_wait_all() is the s_waitcnt their consumers need.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 lds_read128_raw(const unsigned char* ptr) {
    u32x4 v;
    const unsigned addr = (unsigned)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) unsigned char*)ptr);
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_addr(const void* ptr) { return (unsigned)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) unsigned char*)ptr); }
__device__ __forceinline__ void lds_write128_raw(void* ptr, u32x4 v) { asm volatile("ds_write_b128 %0, %1" :: "v"(lds_addr(ptr)), "v"(v)); }
__device__ __forceinline__ void lds_write64_raw(void* ptr, u32x2 v) { asm volatile("ds_write_b64 %0, %1" :: "v"(lds_addr(ptr)), "v"(v)); }
__device__ __forceinline__ float4 as_f4(u32x4 v) { return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)); }
#define LDS_WAIT4(a, b, c, d) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define LDS_WAIT6(a, b, c, d, e, f) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f))

// A-operand registers of one K slice (what a thread fetched from global memory for it)
template <int PRO> struct HnA { uint4 a, e; float4 f0, f1; uint4 w0[2]; };

template <int PRO, int EPI>
__global__ __launch_bounds__(512, 1) void hn_gemm_kernel(HnP p) {
    // LDS: A stages [2][128 rows x 80 B] (written through registers: the prologue transforms), B ring [3][512 rows x 64 B] filled by LDS-DMA
    // straight from the slice-major, XOR-swizzled weight image (no registers, no padding: chunk c of row n sits at c ^ ((n >> 2) & 3))
    constexpr int ASTG = HN_BM * HN_ROW, BSTG = 512 * 64, KLOOP = 2 * ASTG + 3 * BSTG;       // 20,480 + 98,304 B
    constexpr int LDSZ = EPI == 3 ? (HN_BM * HN_DY > KLOOP ? HN_BM * HN_DY : KLOOP) : KLOOP;
    __shared__ __attribute__((aligned(16))) unsigned char sm[LDSZ];
    __shared__ __attribute__((aligned(16))) float cst[(PRO == 1) ? 2 * 512 : (PRO == 3 ? 3 * 512 : 4)];   // PRO 1: scale | shift ; PRO 3: scale | shift | b0
    unsigned char* const Abase = sm;
    auto bstage = [&](int st) -> unsigned char* { return sm + 2 * ASTG + st * BSTG; };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const long row0 = (long)blockIdx.x * HN_BM;
    const long R = p.R;

    // ---- per-thread staging roles: A (PRO 1 / 4 / 5): one 16-B chunk of the 128 x 64-B slice piece ----
    const int arow = tid >> 2, achk = tid & 3;
    long agr = row0 + arow; if (agr >= R) agr = R - 1;                   // PRO 5 (row-major caller tensor): clamped
    float cg_ie = 0.f, cg_ir = 0.f, cg_pj = 0.f, cg_coef = 0.f;
    if (PRO == 4) {
        cg_coef = p.coef[0];
        const float ne = p.rowstats[3 * agr], nr = p.rowstats[3 * agr + 1], cc = p.rowstats[3 * agr + 2];
        cg_ie = 1.f / ne; cg_ir = 1.f / nr; cg_pj = (nr > 1e-12f) ? cc : 0.f;       // as loss.hip cos_grad_body
    }
    // PRO 3: this wave's (waves 0-3) 32 feature rows as B-operand fragments
    uint4 xf[2];
    if (PRO == 3 && wave < 4) {
        long gr = row0 + 32 * wave + i; if (gr >= R) gr = R - 1;
        xf[0] = load_x8(p.X, p.x_bf16, gr, 8 * h); xf[1] = load_x8(p.X, p.x_bf16, gr, 16 + 8 * h);
    }
    // Every load below is UNCONDITIONAL (slice indices past the end are clamped; their LDS stage is one nobody reads any more): the waits
    // are COUNTED (vmcnt(N) = "all but the N youngest"), and N must be the same on every path.
    // issue(sl): this wave's four 1-KB pieces of slice sl's B operand by LDS-DMA into ring stage st, then its A-operand registers
    auto issue = [&](int sl, int st, HnA<PRO>& r) {
        const int sc = sl < 16 ? sl : 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int piece = 4 * wave + j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.W + (size_t)sc * 16384 + (size_t)piece * 512 + (size_t)lane * 8),
                                             (__attribute__((address_space(3))) void*)(bstage(st) + piece * 1024), 16, 0, 0);
        }
        // tiled activations: the (block, slice) piece is 8 KB contiguous, thread t takes bytes [16 t, 16 t + 16)
        if (PRO == 1 || PRO == 4) r.a = *(const uint4*)(p.A + ((size_t)blockIdx.x * 16 + sc) * 4096 + (size_t)tid * 8);
        if (PRO == 4) r.e = *(const uint4*)(p.Bref + ((size_t)blockIdx.x * 16 + sc) * 4096 + (size_t)tid * 8);
        if (PRO == 5) { const float* g = p.Af + (size_t)agr * 512 + 32 * sc + 8 * achk; r.f0 = *(const float4*)g; r.f1 = *(const float4*)(g + 4); }
        if (PRO == 3) {      // (all eight waves load: the count of outstanding operations stays wave-independent; waves 4-7 ignore the data)
            const bf16_t* w0 = p.W0 + (size_t)(32 * sc + i) * 32 + 8 * h;
            r.w0[0] = *(const uint4*)w0; r.w0[1] = *(const uint4*)(w0 + 16);
        }
    };
    constexpr int NISSUE = 4 + (PRO == 1 ? 1 : 0) + (PRO == 4 ? 2 : 0) + (PRO == 5 ? 2 : 0) + (PRO == 3 ? 2 : 0);     // VMEM operations per issue()
    auto stage_a = [&](int stage, int sl, const HnA<PRO>& r) {
        unsigned char* const As = Abase + stage * ASTG;
        if (PRO == 1) {
            float v[8];
            unpack8(r.a, v);
            const int k = 32 * sl + 8 * achk;
            u32x4 q0 = lds_read128_raw((const unsigned char*)(cst + k)), q1 = lds_read128_raw((const unsigned char*)(cst + k + 4));
            u32x4 q2 = lds_read128_raw((const unsigned char*)(cst + 512 + k)), q3 = lds_read128_raw((const unsigned char*)(cst + 512 + k + 4));
            LDS_WAIT4(q0, q1, q2, q3);
            const float4 s0 = as_f4(q0), s1 = as_f4(q1), t0 = as_f4(q2), t1 = as_f4(q3);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
            { const uint4 o = pack8(v); lds_write128_raw(As + arow * HN_ROW + 16 * achk, u32x4{o.x, o.y, o.z, o.w}); }
        } else if (PRO == 4) {
            float a[8], b[8];
            unpack8(r.a, a); unpack8(r.e, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = cos_grad_elem(cg_coef, a[e], b[e], cg_ie, cg_ir, cg_pj);
            { const uint4 o = pack8(a); lds_write128_raw(As + arow * HN_ROW + 16 * achk, u32x4{o.x, o.y, o.z, o.w}); }
        } else if (PRO == 5) {
            lds_write128_raw(As + arow * HN_ROW + 16 * achk, u32x4{pack_bf2(r.f0.x, r.f0.y), pack_bf2(r.f0.z, r.f0.w), pack_bf2(r.f1.x, r.f1.y), pack_bf2(r.f1.z, r.f1.w)});
        } else if (PRO == 3) {
            if (wave < 4) {
                // D[hidden 32][row 32] = W0[slice] x^T: lane i = feature row, accumulator rows = hidden units 8 j + 4 h + q of the slice
                f32x16 ha, hb;
                {   // b0 of this lane's 4 x 4 hidden units seeds the first accumulation chain (raw LDS reads: see lds_read128_raw)
                    u32x4 cb0 = lds_read128_raw((const unsigned char*)(cst + 1024 + 32 * sl + 4 * h)), cb1 = lds_read128_raw((const unsigned char*)(cst + 1024 + 32 * sl + 8 + 4 * h));
                    u32x4 cb2 = lds_read128_raw((const unsigned char*)(cst + 1024 + 32 * sl + 16 + 4 * h)), cb3 = lds_read128_raw((const unsigned char*)(cst + 1024 + 32 * sl + 24 + 4 * h));
                    LDS_WAIT4(cb0, cb1, cb2, cb3);
                    const float4 b0 = as_f4(cb0), b1 = as_f4(cb1), b2 = as_f4(cb2), b3 = as_f4(cb3);
                    ha[0] = b0.x; ha[1] = b0.y; ha[2] = b0.z; ha[3] = b0.w; ha[4] = b1.x; ha[5] = b1.y; ha[6] = b1.z; ha[7] = b1.w;
                    ha[8] = b2.x; ha[9] = b2.y; ha[10] = b2.z; ha[11] = b2.w; ha[12] = b3.x; ha[13] = b3.y; ha[14] = b3.z; ha[15] = b3.w;
#pragma unroll
                    for (int r = 0; r < 16; ++r) hb[r] = 0.f;
                }
                ha = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r.w0[0]), __builtin_bit_cast(bf16x8, xf[0]), ha, 0, 0, 0);
                hb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r.w0[1]), __builtin_bit_cast(bf16x8, xf[1]), hb, 0, 0, 0);
                const int row = 32 * wave + i;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {                 // BatchNorm scale | shift two quads at a time (register budget)
                    const int k = 32 * sl + 16 * jp + 4 * h;
                    u32x4 c0 = lds_read128_raw((const unsigned char*)(cst + k)), c1 = lds_read128_raw((const unsigned char*)(cst + k + 8));
                    u32x4 c2 = lds_read128_raw((const unsigned char*)(cst + 512 + k)), c3 = lds_read128_raw((const unsigned char*)(cst + 512 + k + 8));
                    LDS_WAIT4(c0, c1, c2, c3);
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * jp + jj;
                        const float4 sc = as_f4(jj ? c1 : c0), sh = as_f4(jj ? c3 : c2);
                        const float v0 = fmaxf(fmaf(ha[4 * j] + hb[4 * j], sc.x, sh.x), 0.f), v1 = fmaxf(fmaf(ha[4 * j + 1] + hb[4 * j + 1], sc.y, sh.y), 0.f);
                        const float v2 = fmaxf(fmaf(ha[4 * j + 2] + hb[4 * j + 2], sc.z, sh.z), 0.f), v3 = fmaxf(fmaf(ha[4 * j + 3] + hb[4 * j + 3], sc.w, sh.w), 0.f);
                        lds_write64_raw(As + row * HN_ROW + 16 * j + 8 * h, u32x2{pack_bf2(v0, v1), pack_bf2(v2, v3)});
                    }
                }
            }
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto mma = [&](int sl, int st) {
        const unsigned char* As = Abase + (sl & 1) * ASTG; const unsigned char* Bs = bstage(st);
        const int sw = (i >> 2) & 3;                              // the B image's swizzle for this lane's rows (n = 64 wave + 32 b + i)
        u32x4 fa0, fa1, fa2, fa3, fb0, fb1, ga0, ga1, ga2, ga3, gb0, gb1;          // k-step 0 / 1: four A, two B fragments
        constexpr bool DEEP = EPI != 3 && PRO != 3;                          // both k-steps' fragments in flight at once where the registers allow it
        auto rd_a = [&](int ks, int a) { return lds_read128_raw(As + (32 * a + i) * HN_ROW + 32 * ks + 16 * h); };
        auto rd_b = [&](int ks, int b) { return lds_read128_raw(Bs + (64 * wave + 32 * b + i) * 64 + 16 * ((2 * ks + h) ^ sw)); };
        auto mm8 = [&](const u32x4& a0, const u32x4& a1, const u32x4& a2, const u32x4& a3, const u32x4& b0, const u32x4& b1) {
            const bf16x8 A0 = __builtin_bit_cast(bf16x8, a0), A1 = __builtin_bit_cast(bf16x8, a1), A2 = __builtin_bit_cast(bf16x8, a2), A3 = __builtin_bit_cast(bf16x8, a3);
            const bf16x8 B0 = __builtin_bit_cast(bf16x8, b0), B1 = __builtin_bit_cast(bf16x8, b1);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B0, acc[0][0], 0, 0, 0); acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A0, B1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B0, acc[1][0], 0, 0, 0); acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, acc[1][1], 0, 0, 0);
            acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B0, acc[2][0], 0, 0, 0); acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B1, acc[2][1], 0, 0, 0);
            acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, B0, acc[3][0], 0, 0, 0); acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, B1, acc[3][1], 0, 0, 0);
        };
        fa0 = rd_a(0, 0); fa1 = rd_a(0, 1); fa2 = rd_a(0, 2); fa3 = rd_a(0, 3); fb0 = rd_b(0, 0); fb1 = rd_b(0, 1);
        if (DEEP) {
            ga0 = rd_a(1, 0); ga1 = rd_a(1, 1); ga2 = rd_a(1, 2); ga3 = rd_a(1, 3); gb0 = rd_b(1, 0); gb1 = rd_b(1, 1);
            LDS_WAIT6(fa0, fa1, fa2, fa3, fb0, fb1);
            LDS_WAIT6(ga0, ga1, ga2, ga3, gb0, gb1);
            mm8(fa0, fa1, fa2, fa3, fb0, fb1);
            mm8(ga0, ga1, ga2, ga3, gb0, gb1);
        } else {
            LDS_WAIT6(fa0, fa1, fa2, fa3, fb0, fb1);
            mm8(fa0, fa1, fa2, fa3, fb0, fb1);
            ga0 = rd_a(1, 0); ga1 = rd_a(1, 1); ga2 = rd_a(1, 2); ga3 = rd_a(1, 3); gb0 = rd_b(1, 0); gb1 = rd_b(1, 1);
            LDS_WAIT6(ga0, ga1, ga2, ga3, gb0, gb1);
            mm8(ga0, ga1, ga2, ga3, gb0, gb1);
        }
    };

    if (PRO == 1) { cst[tid] = p.pscale[tid]; cst[512 + tid] = p.pshift[tid]; }
    if (PRO == 3) { cst[tid] = p.pscale[tid]; cst[512 + tid] = p.pshift[tid]; cst[1024 + tid] = p.b0[tid]; }
    HnA<PRO> r0, r1;
    issue(0, 0, r0);
    issue(1, 1, r1);
    if (PRO == 1 || PRO == 3) lds_barrier();                 // the constants are in LDS before the first A transform
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NISSUE) : "memory");         // slice 0 has landed (slice 1 may still fly)
    stage_a(0, 0, r0);
    lds_barrier();
    // iteration sl: loads of slice sl + 2 go out FIRST, then the MFMAs of slice sl (no wait in front of them), then slice sl + 1 -- issued a
    // whole iteration ago, two matrix phases of cover -- is awaited, its A operand transformed into the other A stage, and published
#define HN_ITER(SL, ST, RFILL, RUSE) do { \
        issue((SL) + 2, ((ST) + 2) % 3, RFILL); \
        mma((SL), (ST)); \
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NISSUE) : "memory"); \
        stage_a(((SL) + 1) & 1, (SL) + 1 < 16 ? (SL) + 1 : 15, RUSE); \
        lds_barrier(); } while (0)
#pragma unroll 1
    for (int s6 = 0; s6 < 12; s6 += 6) {                          // ring of three B stages x two A register sets: period 6
        HN_ITER(s6 + 0, 0, r0, r1); HN_ITER(s6 + 1, 1, r1, r0); HN_ITER(s6 + 2, 2, r0, r1);
        HN_ITER(s6 + 3, 0, r1, r0); HN_ITER(s6 + 4, 1, r0, r1); HN_ITER(s6 + 5, 2, r1, r0);
    }
    HN_ITER(12, 0, r0, r1); HN_ITER(13, 1, r1, r0); HN_ITER(14, 2, r0, r1); HN_ITER(15, 0, r1, r0);
#undef HN_ITER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the clamped tail loads land before the LDS is reused

    const int tq = (i & 3) + 4 * h, tc = 4 * (i >> 2);           // transposed layout: row offset within a group of eight, first of four columns
    if (EPI != 3) {
        // ---- bias, (column statistics,) narrow store in the quad-transposed layout: 8 B per lane, tiled tensor ----
        float rs0[16], rs1[16];                                  // EPI 2 / 5: this lane's partial row sums, index 4 a + g (row 32 a + 8 g + tq)
        if (EPI == 2 || EPI == 5) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { rs0[q] = 0.f; rs1[q] = 0.f; }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = 64 * wave + 32 * b + i;
            const float bias = p.bias ? p.bias[col] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[a][b][r] + bias;
                    if (EPI == 1 && row0 + 32 * a + acc_row(r, h) < R) { s1 += v; s2 += v * v; }
                    t[r] = v;
                }
                quad_transpose(t, lane);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int lr = 32 * a + 8 * g + tq;
                    const size_t off = ((size_t)blockIdx.x * 16 + 2 * wave + b) * 4096 + (size_t)lr * 32 + tc;
                    const uint2 o = f4_to_bf4(make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]));
                    *(uint2*)(p.C + off) = o;                   // (rows beyond R of the last block land in the tensor's padding)
                    if (EPI == 2 || EPI == 5) {
                        const float4 cv = bf4_to_f4(o);         // the STORED values: what every later reader of the tensor sees
                        rs0[4 * a + g] += cv.x * cv.x + cv.y * cv.y + cv.z * cv.z + cv.w * cv.w;
                        if (EPI == 5) {
                            const float4 ev = bf4_to_f4(*(const uint2*)(p.E + off));
                            rs1[4 * a + g] += cv.x * ev.x + cv.y * ev.y + cv.z * ev.z + cv.w * ev.w;
                        }
                    }
                }
            }
            if (EPI == 1) {        // every column belongs to ONE wave: its 128-row partial goes straight to the partials (no cross-wave reduction)
                s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
                if (h == 0) { p.part[((long)blockIdx.x * 2 + 0) * 512 + col] = s1; p.part[((long)blockIdx.x * 2 + 1) * 512 + col] = s2; }
            }
        }
        if (EPI == 2 || EPI == 5) {
            // the eight lanes i >> 2 = 0 .. 7 (same tq, h) hold one row's 32 columns of this wave: xor-shuffles over lane bits 2, 3, 4, then the
            // eight waves' partials meet in LDS (the K-loop stages are dead)
            float* const red = (float*)sm;                       // [8 waves][128 rows][2]
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float x = rs0[q], y = rs1[q];
                x += __shfl_xor(x, 4); x += __shfl_xor(x, 8); x += __shfl_xor(x, 16);
                if (EPI == 5) { y += __shfl_xor(y, 4); y += __shfl_xor(y, 8); y += __shfl_xor(y, 16); }
                if ((i >> 2) == 0) {
                    const int lr = 32 * (q >> 2) + 8 * (q & 3) + tq;
                    red[(wave * 128 + lr) * 2] = x; red[(wave * 128 + lr) * 2 + 1] = y;
                }
            }
            __syncthreads();
            float term = 0.f;
            if (tid < 128) {
                float cc = 0.f, ce = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) { cc += red[(w * 128 + tid) * 2]; ce += red[(w * 128 + tid) * 2 + 1]; }
                const long row = row0 + tid;
                if (row < R) {
                    if (EPI == 2) p.rs[row] = cc;
                    else {
                        const float ne = fmaxf(sqrtf(p.rs[row]), 1e-12f), nr = fmaxf(sqrtf(cc), 1e-12f);      // F.normalize eps (external_model_adapt.py:421-423)
                        const float cs = ce / (ne * nr);
                        p.rowstats_out[3 * row] = ne; p.rowstats_out[3 * row + 1] = nr; p.rowstats_out[3 * row + 2] = cs;
                        term = 2.f - 2.f * cs;
                    }
                }
            }
            if (EPI == 5) {
                // block partial of sum_r (2 - 2 cos) in a fixed order; the unused tail of the loss kernels' partial array is cleared
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) term += __shfl_xor(term, o);
                __syncthreads();
                if (tid < 128 && lane == 0) red[wave] = term;
                __syncthreads();
                if (tid == 0) p.cpart[blockIdx.x] = red[0] + red[1];
                for (int k = (int)gridDim.x + (int)blockIdx.x * 512 + tid; k < p.cpart_n; k += (int)gridDim.x * 512) p.cpart[k] = 0.f;
            }
        }
        return;
    }
    // ---- EPI 3: acc = d L / d (proj.3 input) before the ReLU mask.  Hidden h = x W0^T + b0 recomputed in the accumulator layout
    // (A = feature rows, B = W0 rows of the column tile, K = 32) for the mask and the BatchNorm-backward sums ----
    unsigned char* const DY = sm;                                // [128][HN_DY]: the K-loop stages are dead
    {
        uint4 xr[4][2];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            long gr = row0 + 32 * a + i; if (gr >= R) gr = R - 1;
            xr[a][0] = load_x8(p.X, p.x_bf16, gr, 8 * h); xr[a][1] = load_x8(p.X, p.x_bf16, gr, 16 + 8 * h);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = 64 * wave + 32 * b + i;
            const bf16_t* w0 = p.W0 + (size_t)col * 32 + 8 * h;
            const bf16x8 wa = __builtin_bit_cast(bf16x8, *(const uint4*)w0), wb = __builtin_bit_cast(bf16x8, *(const uint4*)(w0 + 16));
            const float b0c = p.b0[col], esc = p.escale[col], esh = p.eshift[col], emu = p.emean[col], eiv = p.einv[col];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x16 hh, h2;
#pragma unroll
                for (int r = 0; r < 16; ++r) { hh[r] = b0c; h2[r] = 0.f; }
                hh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xr[a][0]), wa, hh, 0, 0, 0);
                h2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xr[a][1]), wb, h2, 0, 0, 0);
                f32x16 t;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float hv = hh[r] + h2[r];
                    float v = acc[a][b][r];
                    v = (fmaf(hv, esc, esh) > 0.f) ? v : 0.f;
                    if (row0 + 32 * a + acc_row(r, h) < R) { s1 += v; s2 += v * (hv - emu) * eiv; }
                    t[r] = v * esc;                                  // x gamma x invstd (escale = gamma * invstd)
                }
                quad_transpose(t, lane);
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(uint2*)(DY + (32 * a + 8 * g + tq) * HN_DY + 2 * (64 * wave + 32 * b + tc)) = f4_to_bf4(make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]));
            }
            s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
            if (h == 0) { p.part[((long)blockIdx.x * 2 + 0) * 512 + col] = s1; p.part[((long)blockIdx.x * 2 + 1) * 512 + col] = s2; }
        }
    }
    lds_barrier();
    // ---- P[128][32] = dy[128][512] W0[512][32]: wave w = row tile (w & 3), K half (w >> 2): 16 k-steps each; A = the LDS plane, B = W0^T rows
    // (lane = channel, 8 consecutive hidden units) from L2 ----
    {
        const int a = wave & 3, kh = wave >> 2;
        f32x16 pacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) pacc[r] = 0.f;
        const bf16_t* wt = p.W0t + (size_t)i * 512 + 256 * kh + 8 * h;
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, *(const uint4*)(DY + (32 * a + i) * HN_DY + 2 * (256 * kh + 16 * kk) + 16 * h));
            const bf16x8 bw = __builtin_bit_cast(bf16x8, *(const uint4*)(wt + 16 * kk));
            pacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bw, pacc, 0, 0, 0);
        }
        lds_barrier();                                          // every wave is done reading the plane
        float* const PS = (float*)sm;                           // [4 row tiles][32 rows][32 channels] partials of the upper K half
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) PS[(a * 32 + acc_row(r, h)) * 32 + i] = pacc[r];
        }
        lds_barrier();
        if (kh == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pacc[r] += PS[(a * 32 + acc_row(r, h)) * 32 + i];
            quad_transpose(pacc, lane);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long row = row0 + 32 * a + 8 * g + tq;
                if (row < R) *(float4*)(p.P + row * 32 + tc) = make_float4(pacc[4 * g], pacc[4 * g + 1], pacc[4 * g + 2], pacc[4 * g + 3]);
            }
        }
    }
}

int ptta_hn_row_blocks(long R) { return (int)((R + HN_BM - 1) / HN_BM); }

int ptta_launch_hn_gemm(const HnGemmArgs& a, hipStream_t s) {
    if (!a.W || a.R < 1) return -22;
    HnP p;
    p.Af = a.Af; p.A = a.A; p.Bref = a.Bref; p.rowstats = a.rowstats; p.coef = a.coef; p.X = a.X; p.x_bf16 = a.x_bf16;
    p.E = a.E; p.rs = a.rs; p.rowstats_out = a.rowstats_out; p.cpart = a.cpart; p.cpart_n = a.cpart_n;
    p.W0 = a.W0; p.b0 = a.b0; p.pscale = a.pscale; p.pshift = a.pshift; p.W = a.W; p.bias = a.bias; p.C = a.C; p.part = a.part;
    p.escale = a.escale; p.eshift = a.eshift; p.emean = a.emean; p.einv = a.einv; p.W0t = a.W0t; p.P = a.P; p.R = a.R;
    const dim3 grid(ptta_hn_row_blocks(a.R));
    const int key = a.pro * 10 + a.epi;
    if ((a.pro == 1 && (!a.A || !a.pscale || !a.pshift)) || (a.pro == 3 && (!a.X || !a.W0 || !a.b0 || !a.pscale || !a.pshift)) ||
        (a.pro == 4 && (!a.A || !a.Bref || !a.rowstats || !a.coef)) || (a.pro == 5 && !a.Af) || (a.epi == 2 && !a.rs) ||
        (a.epi == 5 && (!a.E || !a.rs || !a.rowstats_out || !a.cpart || a.cpart_n < (int)grid.x)) || (a.epi != 3 && !a.C) || (a.epi == 1 && !a.part) ||
        (a.epi == 3 && (!a.X || !a.W0 || !a.W0t || !a.b0 || !a.escale || !a.eshift || !a.emean || !a.einv || !a.part || !a.P))) return -22;
#define HN_(PRO, EPI) hipLaunchKernelGGL((hn_gemm_kernel<PRO, EPI>), grid, dim3(512), 0, s, p)
    switch (key) {
        case 10: HN_(1, 0); break; case 11: HN_(1, 1); break; case 12: HN_(1, 2); break;
        case 30: HN_(3, 0); break; case 31: HN_(3, 1); break; case 35: HN_(3, 5); break;
        case 43: HN_(4, 3); break; case 53: HN_(5, 3); break;
        default: return -22;
    }
#undef HN_
    PTTA_CHECK_LAUNCH();
    return 0;
}


// [N = 512][K] bf16 row-major -> slice-major [K / 32][512][32] with the four 16-B chunks of a row at c ^ ((n >> 2) & 3): the LDS-DMA ring of
// hn_gemm_kernel copies it unpadded, and a 16-lane group of ds_read_b128 then touches every bank once
__global__ void hn_pack_w_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int K) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 512L * K) return;
    const int n = (int)(idx / K), k = (int)(idx % K);
    const int c = (k & 31) >> 3, cs = c ^ ((n >> 2) & 3);              // 16-B chunk position inside the row's 64 B: XOR-swizzled for the LDS ring
    dst[((size_t)(k >> 5) * 512 + n) * 32 + 8 * cs + (k & 7)] = src[idx];
}
void ptta_hn_pack_w(const bf16_t* w_hi_rowmajor, bf16_t* w_slice_major, int K, hipStream_t s) {
    hipLaunchKernelGGL(hn_pack_w_kernel, dim3((512 * K + 255) / 256), dim3(256), 0, s, w_hi_rowmajor, w_slice_major, K);
}
// tiled narrow [R][512] -> row-major fp32 (callers' tensors, tests)
__global__ void hn_untile_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long R) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < R * 512; idx += (long)gridDim.x * blockDim.x)
        dst[idx] = bf2f(src[hn_tiled(idx >> 9, (int)(idx & 511))]);
}
int ptta_launch_hn_untile(const void* src_tiled, float* dst, long R, hipStream_t s) {
    long b = (R * 512 + 255) / 256; if (b > 4096) b = 4096;
    hipLaunchKernelGGL(hn_untile_kernel, dim3((int)b), dim3(256), 0, s, (const bf16_t*)src_tiled, dst, R);
    PTTA_CHECK_LAUNCH();
    return 0;
}
long ptta_hn_tiled_elems(long R) { return (long)ptta_hn_row_blocks(R) * 128 * 512; }

// ---- train-mode BatchNorm1d statistics of h = X W0^T + b0 WITHOUT computing h (heads.hip head_moments_kernel's job, as two short launches):
// (1) second moments of the 32-channel feature rows per 256-row block: S[a][b] = sum_r x_a x_b, s[a] = sum_r x_a -- fp32 inside a
//     128-row sub-tile, fp64 across sub-tiles and blocks;
// (2) ONE block per pass reduces the blocks in a fixed order (fp64), centres the moments (mean m, covariance C = S / n - m m^T, rounded to
//     fp32 only AFTER the subtraction) and evaluates per hidden unit j: mean_j = w_j . m + b_j, var_j = w_j^T C w_j.  It hands
//     sum h = n mean_j and sum h^2 = n (var_j + mean_j^2) to the ordinary finalize as TWO float partial "blocks" (value, and what the
//     float rounding of it lost): the layout ptta_launch_bn_finalize / ptta_stat_sync reduce, [pass][2 blocks][2][512].
#define HNM_ROWS 128       // rows per block of the first stage (one sub-tile: 209 blocks per KITTI frame)
int ptta_hn_moment_blocks(long R) { return (int)((R + HNM_ROWS - 1) / HNM_ROWS); }
// doubles of scratch per pass: the first stage's partial sets, then the reduced 32 x 33 moments
long ptta_hn_moment_scratch(long R) { return (long)ptta_hn_moment_blocks(R) * 1056 + 1056; }
// stage 1: second moments [32][33] (column 32: the sums) of 128 feature rows per block, fp32 products of one sub-tile summed into doubles
template <typename TX>                                     // TX = float: fp32 feature rows; bf16_t: the proxy frames' narrow rows
__global__ __launch_bounds__(256) void hn_moments_kernel(const TX* __restrict__ X, long R, double* __restrict__ Sp, long pass_stride) {
    __shared__ __attribute__((aligned(16))) float xs[128][36];
    const int t = threadIdx.x, a = t >> 3, b4 = (t & 7) * 4;
    const TX* Xp = X + (size_t)blockIdx.y * R * 32;
    const long r0 = (long)blockIdx.x * HNM_ROWS;
    if constexpr (sizeof(TX) == 4) {
        float4 pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t + 256 * q;                   // float4 index: row = idx >> 3, channels 4 (idx & 7) ...
            const long row = r0 + (idx >> 3);
            pre[q] = *(const float4*)(Xp + (row < R ? row : R - 1) * 32 + 4 * (idx & 7));      // unconditional: the four loads fly together
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = t + 256 * q;
            *(float4*)&xs[idx >> 3][4 * (idx & 7)] = r0 + (idx >> 3) < R ? pre[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = t + 256 * q;                   // 16-byte index: row = idx >> 2, channels 8 (idx & 3) ...
            const long row = r0 + (idx >> 2);
            uint4 u = *(const uint4*)(Xp + (row < R ? row : R - 1) * 32 + 8 * (idx & 3));
            if (row >= R) u = make_uint4(0u, 0u, 0u, 0u);
            float* d = &xs[idx >> 2][8 * (idx & 3)];
            *(float4*)d = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
            *(float4*)(d + 4) = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u));
        }
    }
    __syncthreads();
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, fs = 0.f;
#pragma unroll 16
    for (int r = 0; r < 128; ++r) {
        const float xa = xs[r][a];
        const float4 xb = *(const float4*)&xs[r][b4];
        f0 = fmaf(xa, xb.x, f0); f1 = fmaf(xa, xb.y, f1); f2 = fmaf(xa, xb.z, f2); f3 = fmaf(xa, xb.w, f3);
        fs += xa;
    }
    double* out = Sp + (size_t)blockIdx.y * pass_stride + (size_t)blockIdx.x * (32 * 33);
    out[a * 33 + b4] = (double)f0; out[a * 33 + b4 + 1] = (double)f1; out[a * 33 + b4 + 2] = (double)f2; out[a * 33 + b4 + 3] = (double)f3;
    if (b4 == 0) out[a * 33 + 32] = (double)fs;
}
// stage 2, 33 blocks per pass: block g sums the partial sets of elements 32 g .. 32 g + 31 -- thread (element i, lane k of 32) takes the sets
// k, k + 32, ... (up to eight independent loads in flight), the 32 lanes' sums are added in a fixed order.  (A last-block-done form of stages
// 2 + 3 in one launch was measured: its agent-scope release writes back the whole L2 under the other chains' kernels -- the step got 2 % slower.)
#define HNM_MAXSETS 256
__global__ __launch_bounds__(1024) void hn_moment_reduce_kernel(double* __restrict__ Sp, int nblk, long pass_stride) {
    __shared__ double S[32 * 32];
    const int t = threadIdx.x, g = blockIdx.x;
    double* sp = Sp + (size_t)blockIdx.y * pass_stride;
    const int i = t & 31, kl = t >> 5, e = g * 32 + i;
    double v[HNM_MAXSETS / 32];
#pragma unroll
    for (int j = 0; j < HNM_MAXSETS / 32; ++j) { const int k = kl + 32 * j; v[j] = k < nblk ? sp[(size_t)k * 1056 + e] : 0.0; }
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < HNM_MAXSETS / 32; ++j) acc += v[j];
    for (int k = kl + HNM_MAXSETS; k < nblk; k += 32) acc += sp[(size_t)k * 1056 + e];      // (frames of more than 32K rows)
    S[kl * 32 + i] = acc;
    __syncthreads();
    if (t < 32) {
        double r = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) r += S[k * 32 + t];          // fixed order
        sp[(size_t)nblk * 1056 + g * 32 + t] = r;
    }
}
// stage 3, one block per pass: the complete moments -> BatchNorm1d statistics of Linear(32, 512): mean_j = w_j . m + b_j, var_j = w_j^T C w_j,
// written as the two-"block" column partials bn_finalize expects.  Two threads per column (rows a < 16 / a >= 16 of C), four accumulators each.
__global__ __launch_bounds__(1024) void hn_moment_stats_kernel(const double* __restrict__ Sp, int nblk, long pass_stride, long R, const float* __restrict__ W0,
                                                               const float* __restrict__ b0, float* __restrict__ part, HnBnOut bo) {
    __shared__ double S[32 * 33];
    __shared__ __attribute__((aligned(16))) float Cf[32][32];
    __shared__ float mf[32];
    __shared__ float vpart[2][512], mpart[2][512];
    const int t = threadIdx.x, pass = blockIdx.x;
    const double* Sg = Sp + (size_t)pass * pass_stride + (size_t)nblk * 1056;
    const int col = t & 511, hf = t >> 9;
    float w[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float4 v = *(const float4*)(W0 + (size_t)col * 32 + 4 * q); w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
    const float bj = b0[col];
    for (int e = t; e < 32 * 33; e += 1024) S[e] = Sg[e];
    __syncthreads();
    const double n = (double)R;
    {
        const int a = t >> 5, b = t & 31;
        const double ma = S[a * 33 + 32] / n, mb = S[b * 33 + 32] / n;
        Cf[a][b] = (float)(S[a * 33 + b] / n - ma * mb);
        if (b == 0) mf[a] = (float)ma;
    }
    __syncthreads();
    float wm = 0.f, var = 0.f;
#pragma unroll
    for (int aa = 0; aa < 16; ++aa) {
        const int a = 16 * hf + aa;
        float wa = 0.f;
#pragma unroll
        for (int q = 0; q < 32; ++q) wa = (q == a) ? w[q] : wa;           // (w[] stays in registers: no dynamic indexing)
        wm = fmaf(wa, mf[a], wm);
        float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 cv = *(const float4*)&Cf[a][4 * q];
            r0 = fmaf(cv.x, w[4 * q], r0); r1 = fmaf(cv.y, w[4 * q + 1], r1); r2 = fmaf(cv.z, w[4 * q + 2], r2); r3 = fmaf(cv.w, w[4 * q + 3], r3);
        }
        var = fmaf(wa, (r0 + r1) + (r2 + r3), var);
    }
    vpart[hf][col] = var; mpart[hf][col] = wm;
    __syncthreads();
    if (t >= 512) return;
    var = vpart[0][t] + vpart[1][t]; wm = mpart[0][t] + mpart[1][t];
    const double mean = (double)wm + (double)bj, vr = var > 0.f ? (double)var : 0.0;
    if (bo.mean) {
        // no statistics exchange: mean and (biased) variance go straight into the BatchNorm's state -- bn_finalize_kernel's arithmetic
        // without its detour through n * mean and n * E[h^2]
        if (t == 0 && bo.nbt) *bo.nbt += 1;
        const float iv = (float)(1.0 / sqrt(vr + (double)bo.eps));
        bo.mean[t] = (float)mean; bo.inv[t] = iv;
        const float sc = bo.gamma[t] * iv;
        bo.scale[t] = sc; bo.shift[t] = bo.beta[t] - (float)mean * sc;
        if (bo.rm) {
            const double unb = R > 1 ? vr * n / (n - 1.0) : vr;
            bo.rm[t] = (1.f - bo.momentum) * bo.rm[t] + bo.momentum * (float)mean;
            bo.rv[t] = (1.f - bo.momentum) * bo.rv[t] + bo.momentum * (float)unb;
        }
        return;
    }
    const double sh = n * mean, sh2 = n * (vr + mean * mean);
    float* o = part + (size_t)pass * 2 * 2 * 512;
    const float h0 = (float)sh, q0 = (float)sh2;
    o[t] = h0; o[512 + t] = q0;                                                    // "block" 0: the float values
    o[1024 + t] = (float)(sh - (double)h0); o[1536 + t] = (float)(sh2 - (double)q0);      // "block" 1: what the rounding lost
}
// X: `npass` consecutive groups of R rows x 32 (fp32, or bf16 with x_bf16); scratch: npass * ptta_hn_moment_scratch(R) doubles; part: [npass][2][2][512]
// floats (the column partials a statistics exchange + bn_finalize take), or, with `bn` (npass == 1), the BatchNorm's state directly
int ptta_launch_hn_moments(const void* X, int x_bf16, long R, int npass, const float* W0, const float* b0, double* scratch, float* part, const HnBnOut* bn,
                           hipStream_t s) {
    if (!X || !W0 || !b0 || !scratch || (!part && !bn) || R < 1 || npass < 1 || (bn && npass != 1)) return -22;
    const int nb = ptta_hn_moment_blocks(R);
    const long ps = ptta_hn_moment_scratch(R);
    if (x_bf16) hipLaunchKernelGGL((hn_moments_kernel<bf16_t>), dim3(nb, npass), dim3(256), 0, s, (const bf16_t*)X, R, scratch, ps);
    else hipLaunchKernelGGL((hn_moments_kernel<float>), dim3(nb, npass), dim3(256), 0, s, (const float*)X, R, scratch, ps);
    hipLaunchKernelGGL(hn_moment_reduce_kernel, dim3(33, npass), dim3(1024), 0, s, scratch, nb, ps);
    hipLaunchKernelGGL(hn_moment_stats_kernel, dim3(npass), dim3(1024), 0, s, scratch, nb, ps, R, W0, b0, part, bn ? *bn : HnBnOut());
    PTTA_CHECK_LAUNCH();
    return 0;
}
