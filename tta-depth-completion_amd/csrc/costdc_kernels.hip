// CostDCNet-specific kernels; paths relative to the reference root,
// CD = external_src/costdcnet/CostDCNet_adapt.py, U3 = external_src/costdcnet/models/unet3d.py, E3 = .../models/encoder3d.py.
//
// 3-D feature volumes are stored [pass][frame][plane][y][x][channel] (NDHWC): a volume is a batch of frames x planes NHWC
// images for the 1x3x3 convolutions and a batch of frames [plane][y*x] images for the 3x1x1 ones (gnet.h `review`).
//   * input staging (cat(image, sparse) -> 16-channel NHWC, proxy frames see a zero image), dual-corner padding
//   * depth2MDP (CD:356-388) + the sparse 3-D encoder (E3:33-103) as gather-convolutions over dense index volumes
//     (MinkowskiEngine is absent from the reference tree: semantics as restated in oracle/minkowski_lite.py, parity unpinned)
//   * fusion (CD:390-406), MaxPool3d(2) / nearest interpolation of the P3D UNet (U3:86-113), per-plane pixel shuffle +
//     softmax + expected plane (CD:408-424), feature rows of the MLP heads (CD:243-251), and their gradients.
#include "ptta_common.h"
#include "ptta_kernels.h"
#include "costdc.h"

namespace {
inline int nbk(long total, int cap = 16384) { long b = (total + 255) / 256; if (b > cap) b = cap; if (b < 1) b = 1; return (int)b; }
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ---- dual-corner zero padding (src/costdcnet_model_adapt.py:134-185): item k=0 pads top/right, k=1 bottom/left ----------
// norm: Transforms.normalize_images fused into the padding (the reference normalises BEFORE it pads, src/tta_main.py:602 then
// src/costdcnet_model_adapt.py:134-210: padded pixels stay exactly zero, in-frame pixels become (v / div - mean[c]) / std[c])
struct CdNorm { int on; float div; float mean[3]; float stdv[3]; };
__global__ void cd_pad_dual_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W, int Hp, int Wp, int pt, int pr, CdNorm nm) {
    const long total = (long)2 * N * C * Hp * Wp;
    GRID_STRIDE(idx, total) {
        const int x = (int)(idx % Wp); long t_ = idx / Wp;
        const int y = (int)(t_ % Hp); t_ /= Hp;
        const int ch = (int)(t_ % C); t_ /= C;
        const int n = (int)(t_ % N); const int k = (int)(t_ / N);
        const int sy = k == 0 ? y - pt : y, sx = k == 0 ? x : x - pr;
        float v = 0.f;
        if (sy >= 0 && sy < H && sx >= 0 && sx < W) {
            v = src[(((long)n * C + ch) * H + sy) * W + sx];
            if (nm.on) v = (v / nm.div - nm.mean[ch]) / nm.stdv[ch];
        }
        dst[idx] = v;
    }
}
__global__ void cd_crop_avg_kernel(const float* __restrict__ net, float* __restrict__ out, int N, int H, int W, int Hp, int Wp, int pt, int pr) {
    const long total = (long)N * H * W;
    GRID_STRIDE(idx, total) {
        const int x = (int)(idx % W); long t_ = idx / W;
        const int y = (int)(t_ % H); const int n = (int)(t_ / H);
        out[idx] = (net[((long)n * Hp + y + pt) * Wp + x] + net[((long)(N + n) * Hp + y) * Wp + x + pr]) / 2.0f;
    }
}
__global__ void cd_scatter_dual_grad_kernel(const float* __restrict__ g, float* __restrict__ gnet, int N, int H, int W, int Hp, int Wp, int pt, int pr) {
    const long total = (long)2 * N * Hp * Wp;
    GRID_STRIDE(idx, total) {
        const int x = (int)(idx % Wp); long t_ = idx / Wp;
        const int y = (int)(t_ % Hp); t_ /= Hp;
        const int n = (int)(t_ % N); const int k = (int)(t_ / N);
        const int sy = k == 0 ? y - pt : y, sx = k == 0 ? x : x - pr;
        gnet[idx] = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? 0.5f * g[((long)n * H + sy) * W + sx] : 0.f;
    }
}

// ---- input staging: in_2d = cat([image, sparse], 1) (CD:210), zero image for the proxy frames (CD:235) -------------------
// out: [passes*N][H][W][16]: channels 0..2 image (normalised on the fly), 3 sparse depth, 4..15 zero
__global__ void cd_stage_kernel(const float* __restrict__ image, const float* __restrict__ sparse, float* __restrict__ out, int N, int passes, int H, int W, int C,
                                int norm, float div, float m0, float m1, float m2, float s0, float s1, float s2) {
    const long P = (long)H * W, total = (long)passes * N * P;
    GRID_STRIDE(idx, total) {
        const long pix = idx % P; const int b = (int)(idx / P); const int n = b % N; const bool proxy = b >= N;
        float4 v0 = make_float4(0.f, 0.f, 0.f, sparse[(long)n * P + pix]);
        if (!proxy) {
            const float* ip = image + (long)n * 3 * P + pix;
            v0.x = ip[0]; v0.y = ip[P]; v0.z = ip[2 * P];
            if (norm) { v0.x = (v0.x / div - m0) / s0; v0.y = (v0.y / div - m1) / s1; v0.z = (v0.z / div - m2) / s2; }
        }
        float4* o = (float4*)(out + idx * C);           // C = 16 (zero-padded for the matrix-core kernel) or 4
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        o[0] = v0;
        if (C == 16) { o[1] = z; o[2] = z; o[3] = z; }
    }
}
__global__ void cd_clamp_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, float maxd) {
    GRID_STRIDE(i, n) { float v = src[i]; if (maxd >= 0.f) v = fminf(fmaxf(v, 0.f), maxd); dst[i] = v; }
}

// ---- sparse voxel sets ---------------------------------------------------------------------------------------------------
// level l holds voxels (frame, plane z, y, x) whose y, x are multiples of ts = 1 << l (tensor stride (1, ts, ts)); `vol` is
// the dense index volume [N][16][H >> l][W >> l] (-1 = empty); points are numbered in (frame, plane, row, column) order by a
// row count + scan + fill, so every reduction over points has a fixed order.
__device__ __forceinline__ int cd_plane_of(float d, float z_step) { int z = (int)rintf(d / z_step); return z < 0 ? 0 : (z > 15 ? 15 : z); }
// one wave per image row: lanes stride over the columns, ballot + popcount
__global__ __launch_bounds__(256) void cd_l0_rowcount_kernel(const float* __restrict__ sparse, int N, int H, int W, float z_step, int* __restrict__ rowcnt) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N * H) return;
    const float* sp = sparse + (long)row * W;
    int c = 0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const bool on = x < W && cd_plane_of(sp[x], z_step) != 0;
        c += __popcll(__ballot(on));
    }
    if (lane == 0) rowcnt[row] = c;
}
// single block: exclusive scan of `n` counts -> offsets; total -> *cnt
__global__ __launch_bounds__(1024) void cd_scan_kernel(const int* __restrict__ cntin, int n, int* __restrict__ off, int* __restrict__ total) {
    __shared__ int buf[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n ? cntin[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int t = threadIdx.x >= o ? buf[threadIdx.x - o] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n) off[i] = carry + buf[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
// depth2MDP (CD:356-388): plane = clamp(round(d / z_step), 0, 15) (round half to even like torch.round), voxel kept when
// plane != 0, feature = (d - plane * z_step) / z_step
__global__ __launch_bounds__(256) void cd_l0_fill_kernel(const float* __restrict__ sparse, int N, int H, int W, float z_step, const int* __restrict__ rowoff,
                                                         int4* __restrict__ coords, float* __restrict__ feat, int* __restrict__ vol) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N * H) return;
    const int n = row / H, y = row % H;
    const float* sp = sparse + (long)row * W;
    int base = rowoff[row];
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int x = x0 + lane;
        const float d = x < W ? sp[x] : 0.f;
        const int z = cd_plane_of(d, z_step);
        const bool on = x < W && z != 0;
        const unsigned long long m = __ballot(on);
        if (on) {
            const int id = base + __popcll(m & ((1ull << lane) - 1ull));          // column order within the row
            coords[id] = make_int4(n, z, y, x);
            feat[id] = (d - (float)z * z_step) / z_step;
            vol[(((long)n * 16 + z) * H + y) * W + x] = id;
        }
        base += __popcll(m);
    }
}
// coarser level: mark the cells floor(c / 2) of the finer level's voxels (idempotent stores), then count / scan / fill
__global__ void cd_mark_kernel(const int4* __restrict__ coords, const int* __restrict__ cnt, int shift, int Hc, int Wc, int* __restrict__ vol) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < *cnt; i += (long)gridDim.x * blockDim.x) {
        const int4 c = coords[i];
        vol[(((long)c.x * 16 + c.y) * Hc + (c.z >> shift)) * Wc + (c.w >> shift)] = -2;          // occupied, id assigned below
    }
}
__global__ __launch_bounds__(256) void cd_lc_rowcount_kernel(const int* __restrict__ vol, int rows, int Wc, int* __restrict__ rowcnt) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    int c = 0;
    for (int x0 = 0; x0 < Wc; x0 += 64) { const int x = x0 + lane; c += __popcll(__ballot(x < Wc && vol[(long)row * Wc + x] != -1)); }
    if (lane == 0) rowcnt[row] = c;
}
__global__ __launch_bounds__(256) void cd_lc_fill_kernel(int* __restrict__ vol, int rows, int Hc, int Wc, int shift, const int* __restrict__ rowoff, int4* __restrict__ coords) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int y = row % Hc, z = (row / Hc) % 16, n = row / (Hc * 16);
    int base = rowoff[row];
    for (int x0 = 0; x0 < Wc; x0 += 64) {
        const int x = x0 + lane;
        const bool on = x < Wc && vol[(long)row * Wc + x] != -1;
        const unsigned long long m = __ballot(on);
        if (on) {
            const int id = base + __popcll(m & ((1ull << lane) - 1ull));
            vol[(long)row * Wc + x] = id; coords[id] = make_int4(n, z, y << shift, x << shift);
        }
        base += __popcll(m);
    }
}
__global__ void cd_fill_int_kernel(int* __restrict__ p, long n, int v) { GRID_STRIDE(i, n) p[i] = v; }

// generalized sparse convolution: out[u] = sum_k W[k] in[u + off_k * ts_in] over existing inputs; kernel offsets with the
// plane axis fastest, k = (dz+1) + 3*(dy+1) + 9*(dx+1); one block per output voxel, thread = output channel
__global__ __launch_bounds__(64) void cd_sparse_conv_kernel(const float* __restrict__ fin, const int* __restrict__ vol_in, int sh_in, int Hin, int Win,
                                                            const int4* __restrict__ coords_out, const int* __restrict__ cnt_out,
                                                            const float* __restrict__ Wk, int ksize, int Ci, int Co, float* __restrict__ fout) {
    const int co = threadIdx.x;
    const int nk = ksize == 3 ? 27 : 1;
    for (int u = blockIdx.x; u < *cnt_out; u += gridDim.x) {
        const int4 c = coords_out[u];
        float acc = 0.f;
        for (int k = 0; k < nk; ++k) {
            int dz = 0, dy = 0, dx = 0;
            if (ksize == 3) { dz = k % 3 - 1; dy = (k / 3) % 3 - 1; dx = k / 9 - 1; }
            const int z = c.y + dz, y = (c.z >> sh_in) + dy, x = (c.w >> sh_in) + dx;       // offsets are multiples of the input stride
            if (z < 0 || z > 15 || y < 0 || y >= Hin || x < 0 || x >= Win) continue;
            const int id = vol_in[(((long)c.x * 16 + z) * Hin + y) * Win + x];
            if (id < 0) continue;
            if (co < Co) {
                const float* f = fin + (long)id * Ci;
                const float* w = Wk + (long)k * Ci * Co + co;
                for (int ci = 0; ci < Ci; ++ci) acc = fmaf(f[ci], w[(long)ci * Co], acc);
            }
        }
        if (co < Co) fout[(long)u * Co + co] = acc;
    }
}
// BatchNorm1d over the voxels (MinkowskiBatchNorm): partial {sum, sum^2} per channel over fixed chunks of 256 voxels
#define SBN_BLOCKS 256
__global__ __launch_bounds__(64) void cd_sbn_stats_kernel(const float* __restrict__ f, const int* __restrict__ cnt, int C, float* __restrict__ part) {
    const int c = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) for (int u = blockIdx.x; u < *cnt; u += gridDim.x) { const float v = f[(long)u * C + c]; s1 += v; s2 += v * v; }
    if (c < C) { part[((long)blockIdx.x * 2) * C + c] = s1; part[((long)blockIdx.x * 2 + 1) * C + c] = s2; }
}
// train: batch statistics (biased variance), running statistics updated `repeats` times (the reference runs the sparse
// encoder once per pass on the same input); eval: running statistics.  st = [scale, shift][C]
// SyncBatchNorm over the ranks (MinkowskiBatchNorm wraps an nn.BatchNorm1d, which convert_sync_batchnorm converts too):
// double sums {sum[C], sum^2[C], voxel count} for the caller's all-reduce
__global__ void cd_sbn_collapse_kernel(const float* __restrict__ part, const int* __restrict__ cnt, int C, double* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) out[2 * C] = (double)(*cnt);
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < SBN_BLOCKS; ++b) { s1 += (double)part[((long)b * 2) * C + c]; s2 += (double)part[((long)b * 2 + 1) * C + c]; }
    out[c] = s1; out[C + c] = s2;
}
__global__ void cd_sbn_finalize_kernel(const float* __restrict__ part, const int* __restrict__ cnt, int C, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* rm, float* rv, long long* nbt, int train, int repeats, float* __restrict__ st,
                                       const double* __restrict__ gsum = nullptr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && train && nbt) *nbt += repeats;
    if (c >= C) return;
    float mu, var;
    if (train) {
        double s1 = 0.0, s2 = 0.0, R;
        if (gsum) { s1 = gsum[c]; s2 = gsum[C + c]; R = gsum[2 * C] > 0.0 ? gsum[2 * C] : 1.0; }      // global sums / voxel count
        else {
#pragma unroll 8
            for (int b = 0; b < SBN_BLOCKS; ++b) { s1 += (double)part[((long)b * 2) * C + c]; s2 += (double)part[((long)b * 2 + 1) * C + c]; }
            R = (double)(*cnt > 0 ? *cnt : 1);
        }
        const double m = s1 / R; double v = s2 / R - m * m; if (v < 0.0) v = 0.0;
        mu = (float)m; var = (float)v;
        if (rm && rv) {
            const float unb = (float)(R > 1.0 ? v * R / (R - 1.0) : v);
            float a = rm[c], b2 = rv[c];
            for (int k = 0; k < repeats; ++k) { a = 0.9f * a + 0.1f * mu; b2 = 0.9f * b2 + 0.1f * unb; }
            rm[c] = a; rv[c] = b2;
        }
    } else { mu = rm[c]; var = rv[c]; }
    const float iv = 1.f / sqrtf(var + 1e-5f);
    const float sc = gamma[c] * iv;
    st[c] = sc; st[C + c] = beta[c] - mu * sc; st[2 * C + c] = mu; st[3 * C + c] = iv;      // [scale, shift, mean, inv][C]
}
__global__ void cd_sbn_apply_kernel(const float* __restrict__ f, const float* __restrict__ res, const int* __restrict__ cnt, int C, const float* __restrict__ st,
                                    int relu, float* __restrict__ out) {
    const long total = (long)(*cnt) * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        float v = fmaf(f[i], st[c], st[C + c]);
        if (res) v += res[i];
        out[i] = relu ? fmaxf(v, 0.f) : v;
    }
}
// SparseTensor.dense() + the placement of fusion() (CD:393-399): voxels at stride (1,4,4) -> [N][16][h4][w4][C]
__global__ void cd_densify_kernel(const float* __restrict__ f, const int4* __restrict__ coords, const int* __restrict__ cnt, int C, int h4, int w4,
                                  float* __restrict__ dense) {
    const long total = (long)(*cnt) * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C); const int4 q = coords[i / C];
        const int y = q.z >> 2, x = q.w >> 2;
        if (q.y < 16 && y < h4 && x < w4) dense[((((long)q.x * 16 + q.y) * h4 + y) * w4 + x) * C + c] = f[i];
    }
}

// ---- backward through the sparse encoder: only the shared-parameter (DDP) run needs it -- after convert_syncbn() the reference's
// adapt_parameters('meta_bn') adapts the BatchNorm1d inside every MinkowskiBatchNorm too (src/costdcnet_model_adapt.py:364-372), so
// their gamma / beta gradients flow back from the fused volume through densify, the 1x1x1 head and the three BasicBlocks.
// Transposed generalized sparse convolution: gx[j] = sum_k W[k] gy[u] over the output voxels u that read input voxel j with offset k,
// i.e. u = j - off_k * ts_in when that coordinate lies on the output grid and exists.  One block per input voxel, thread = input channel.
__global__ __launch_bounds__(64) void cd_sparse_conv_bwd_kernel(const float* __restrict__ gy, const int* __restrict__ vol_out, int sh_in, int sh_out, int Hout, int Wout,
                                                                const int4* __restrict__ coords_in, const int* __restrict__ cnt_in,
                                                                const float* __restrict__ Wk, int ksize, int Ci, int Co, float* __restrict__ gx, int acc) {
    const int ci = threadIdx.x;
    const int nk = ksize == 3 ? 27 : 1;
    const int ts_in = 1 << sh_in, mask_out = (1 << sh_out) - 1;
    for (int j = blockIdx.x; j < *cnt_in; j += gridDim.x) {
        const int4 c = coords_in[j];
        float a = 0.f;
        for (int k = 0; k < nk; ++k) {
            int dz = 0, dy = 0, dx = 0;
            if (ksize == 3) { dz = k % 3 - 1; dy = (k / 3) % 3 - 1; dx = k / 9 - 1; }
            const int z = c.y - dz, y = c.z - dy * ts_in, x = c.w - dx * ts_in;        // the output voxel's coordinate
            if (z < 0 || z > 15 || y < 0 || x < 0 || (y & mask_out) || (x & mask_out)) continue;
            const int yo = y >> sh_out, xo = x >> sh_out;
            if (yo >= Hout || xo >= Wout) continue;
            const int u = vol_out[(((long)c.x * 16 + z) * Hout + yo) * Wout + xo];
            if (u < 0) continue;
            if (ci < Ci) {
                const float* g = gy + (long)u * Co;
                const float* w = Wk + ((long)k * Ci + ci) * Co;
                for (int co = 0; co < Co; ++co) a = fmaf(g[co], w[co], a);
            }
        }
        if (ci < Ci) { float* o = gx + (long)j * Ci + ci; *o = acc ? *o + a : a; }
    }
}
// BatchNorm over voxels, backward: y = relu?(bn(f) + res?); g1 = g * (y > 0) when relu; partial {sum g1, sum g1 xhat}
__global__ __launch_bounds__(64) void cd_sbn_bwd_stats_kernel(const float* __restrict__ f, const float* __restrict__ y, const float* __restrict__ g,
                                                              const int* __restrict__ cnt, int C, const float* __restrict__ st, int relu, float* __restrict__ part) {
    const int c = threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    if (c < C) {
        const float mu = st[2 * C + c], iv = st[3 * C + c];
        for (int u = blockIdx.x; u < *cnt; u += gridDim.x) {
            float gv = g[(long)u * C + c];
            if (relu && !(y[(long)u * C + c] > 0.f)) gv = 0.f;
            s1 += gv; s2 += gv * (f[(long)u * C + c] - mu) * iv;
        }
        part[((long)blockIdx.x * 2) * C + c] = s1; part[((long)blockIdx.x * 2 + 1) * C + c] = s2;
    }
}
__global__ void cd_sbn_bwd_collapse_kernel(const float* __restrict__ part, const int* __restrict__ cnt, int C, double* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) out[2 * C] = (double)(*cnt);
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int b = 0; b < SBN_BLOCKS; ++b) { s1 += (double)part[((long)b * 2) * C + c]; s2 += (double)part[((long)b * 2 + 1) * C + c]; }
    out[c] = s1; out[C + c] = s2;
}
// dgamma, dbeta (scaled by 1 / world under SyncBatchNorm: the DDP-averaged local gradients, as gbn.hip) and bw = [gamma * inv, c1, c2][C]
__global__ void cd_sbn_bwd_finalize_kernel(const float* __restrict__ part, const int* __restrict__ cnt, int C, const float* __restrict__ gamma,
                                           const float* __restrict__ st, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ bw,
                                           const double* __restrict__ gsum, float grad_scale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0, R;
    if (gsum) { s1 = gsum[c]; s2 = gsum[C + c]; R = gsum[2 * C] > 0.0 ? gsum[2 * C] : 1.0; }
    else {
        for (int b = 0; b < SBN_BLOCKS; ++b) { s1 += (double)part[((long)b * 2) * C + c]; s2 += (double)part[((long)b * 2 + 1) * C + c]; }
        R = (double)(*cnt > 0 ? *cnt : 1);
    }
    dbeta[c] = (float)s1 * grad_scale; dgamma[c] = (float)s2 * grad_scale;
    bw[c] = gamma[c] * st[3 * C + c]; bw[C + c] = (float)(s1 / R); bw[2 * C + c] = (float)(s2 / R);
}
// gx (+)= gamma inv (g1 - c1 - xhat c2);  gres (+)= g1
__global__ void cd_sbn_bwd_apply_kernel(const float* __restrict__ f, const float* __restrict__ y, const float* __restrict__ g, const int* __restrict__ cnt, int C,
                                        const float* __restrict__ st, const float* __restrict__ bw, int relu, float* __restrict__ gx, float* __restrict__ gres,
                                        int acc_res) {
    const long total = (long)(*cnt) * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C);
        float gv = g[i];
        if (relu && !(y[i] > 0.f)) gv = 0.f;
        const float xh = (f[i] - st[2 * C + c]) * st[3 * C + c];
        gx[i] = bw[c] * (gv - bw[C + c] - xh * bw[2 * C + c]);
        if (gres) gres[i] = acc_res ? gres[i] + gv : gv;
    }
}
// gradient of densify: g[u][c] = gvol[(frame, plane, y/4, x/4)][c0 + c] at the level-2 voxels (gvol has `ldv` channels per voxel of the volume)
__global__ void cd_densify_bwd_kernel(const float* __restrict__ gvol, int ldv, int c0, const int4* __restrict__ coords, const int* __restrict__ cnt, int C, int h4, int w4,
                                      float* __restrict__ g) {
    const long total = (long)(*cnt) * C;
    GRID_STRIDE(i, total) {
        const int c = (int)(i % C); const int4 q = coords[i / C];
        const int y = q.z >> 2, x = q.w >> 2;
        g[i] = (q.y < 16 && y < h4 && x < w4) ? gvol[((((long)q.x * 16 + q.y) * h4 + y) * w4 + x) * ldv + c0 + c] : 0.f;
    }
}

// ---- fusion (CD:390-406) ----------------------------------------------------------------------------------------------------
// mask = any_c(feat3d != 0); mask_ = mask + 1 - sum_planes(mask); vol = [feat2d * mask_ | feat3d]  (32 channels)
__global__ void cd_fusion_fwd_kernel(const float* __restrict__ feat2d, const float* __restrict__ feat3d, float* __restrict__ vol, float* __restrict__ maskw,
                                     int N, int passes, int h, int w) {
    const long P = (long)h * w, total = (long)N * P;
    GRID_STRIDE(idx, total) {
        const int n = (int)(idx / P); const long pix = idx % P;
        float occ[16]; float k = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            const float4* f = (const float4*)(feat3d + (((long)n * 16 + d) * P + pix) * 16);
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) { const float4 v = f[q]; any = any || v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f; }
            occ[d] = any ? 1.f : 0.f; k += occ[d];
        }
        for (int d = 0; d < 16; ++d) {
            const float m = occ[d] + (1.f - k);
            maskw[((long)n * 16 + d) * P + pix] = m;
            const float4* f3 = (const float4*)(feat3d + (((long)n * 16 + d) * P + pix) * 16);
            for (int pass = 0; pass < passes; ++pass) {
                const float4* f2 = (const float4*)(feat2d + (((long)pass * N + n) * P + pix) * 16);
                float4* o = (float4*)(vol + ((((long)pass * N + n) * 16 + d) * P + pix) * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) { const float4 v = f2[q]; o[q] = make_float4(v.x * m, v.y * m, v.z * m, v.w * m); }
#pragma unroll
                for (int q = 0; q < 4; ++q) o[4 + q] = f3[q];
            }
        }
    }
}
__global__ void cd_fusion_bwd_kernel(const float* __restrict__ gvol, const float* __restrict__ maskw, float* __restrict__ gfeat2d, int N, int h, int w) {
    const long P = (long)h * w, total = (long)N * P * 16;
    GRID_STRIDE(idx, total) {
        const int c = (int)(idx % 16); const long t_ = idx / 16; const long pix = t_ % P; const int n = (int)(t_ / P);
        float acc = 0.f;
        for (int d = 0; d < 16; ++d) acc = fmaf(gvol[((((long)n * 16 + d) * P + pix) * 32) + c], maskw[((long)n * 16 + d) * P + pix], acc);
        gfeat2d[idx] = acc;
    }
}

// ---- MaxPool3d(2) (U3:91-94), items = frames x planes -------------------------------------------------------------------------
__global__ void cd_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long items_out, int H, int W, int C) {
    const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    const long total = items_out * Ho * Wo * C4;
    GRID_STRIDE(idx, total) {
        const int c = (int)(idx % C4) << 2; long t_ = idx / C4;
        const int xo = (int)(t_ % Wo); t_ /= Wo; const int yo = (int)(t_ % Ho); const long io = t_ / Ho;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 v = *(const float4*)(x + ((((2 * io + (k >> 2)) * H + 2 * yo + ((k >> 1) & 1)) * W) + 2 * xo + (k & 1)) * C + c);
            m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
        }
        *(float4*)(y + idx * 4) = m;
    }
}
// gradient to the FIRST maximum of each window in (plane, row, column) scan order, as ATen's max_pool3d backward
__global__ void cd_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gx, long items_in, int H, int W, int C, int acc) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = items_in * H * W * C;
    GRID_STRIDE(idx, total) {
        const int c = (int)(idx % C); long t_ = idx / C;
        const int xi = (int)(t_ % W); t_ /= W; const int yi = (int)(t_ % H); const long ii = t_ / H;
        float g = 0.f;
        const int xo = xi >> 1, yo = yi >> 1; const long io = ii >> 1;
        if (xo < Wo && yo < Ho) {
            const int me = (int)((ii & 1) << 2) | ((yi & 1) << 1) | (xi & 1);
            float best = -INFINITY; int arg = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float v = x[((((2 * io + (k >> 2)) * H + 2 * yo + ((k >> 1) & 1)) * W) + 2 * xo + (k & 1)) * C + c];
                if (v > best) { best = v; arg = k; }
            }
            if (arg == me) g = gy[(((io * Ho + yo) * Wo) + xo) * C + c];
        }
        gx[idx] = acc ? gx[idx] + g : g;
    }
}

// ---- F.interpolate(size=skip.size()[2:], mode='nearest') (U3:110): src = min(floor(dst * in / out), in - 1) per axis ------
__device__ __forceinline__ int nn_src(int dst, float scale, int in) { const int s = (int)floorf((float)dst * scale); return s < in - 1 ? s : in - 1; }
__global__ void cd_up_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long frames, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C) {
    const int C4 = C >> 2;
    const float sd = (float)Di / Do, sh = (float)Hi / Ho, sw = (float)Wi / Wo;
    const long total = frames * Do * Ho * Wo * C4;
    GRID_STRIDE(idx, total) {
        const int c = (int)(idx % C4) << 2; long t_ = idx / C4;
        const int xo = (int)(t_ % Wo); t_ /= Wo; const int yo = (int)(t_ % Ho); t_ /= Ho; const int zo = (int)(t_ % Do); const long f = t_ / Do;
        *(float4*)(y + idx * 4) = *(const float4*)(x + ((((f * Di + nn_src(zo, sd, Di)) * Hi + nn_src(yo, sh, Hi)) * Wi) + nn_src(xo, sw, Wi)) * C + c);
    }
}
__global__ void cd_up_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, long frames, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int acc) {
    const float sd = (float)Di / Do, sh = (float)Hi / Ho, sw = (float)Wi / Wo;
    const long total = frames * Di * Hi * Wi * C;
    GRID_STRIDE(idx, total) {
        const int c = (int)(idx % C); long t_ = idx / C;
        const int xi = (int)(t_ % Wi); t_ /= Wi; const int yi = (int)(t_ % Hi); t_ /= Hi; const int zi = (int)(t_ % Di); const long f = t_ / Di;
        float g = 0.f;
        // destinations that map to this source: a small window around dst = src * out / in
        const int z0 = max(0, (int)((long)zi * Do / Di) - 1), z1 = min(Do - 1, (int)((long)(zi + 1) * Do / Di) + 1);
        const int y0 = max(0, (int)((long)yi * Ho / Hi) - 1), y1 = min(Ho - 1, (int)((long)(yi + 1) * Ho / Hi) + 1);
        const int x0 = max(0, (int)((long)xi * Wo / Wi) - 1), x1 = min(Wo - 1, (int)((long)(xi + 1) * Wo / Wi) + 1);
        for (int z = z0; z <= z1; ++z) {
            if (nn_src(z, sd, Di) != zi) continue;
            for (int y = y0; y <= y1; ++y) {
                if (nn_src(y, sh, Hi) != yi) continue;
                for (int x = x0; x <= x1; ++x)
                    if (nn_src(x, sw, Wi) == xi) g += gy[((((f * Do + z) * Ho + y) * Wo) + x) * C + c];
            }
        }
        gx[idx] = acc ? gx[idx] + g : g;
    }
}

// ---- upsampling + disparity_regression (CD:408-424): pred[n, 4y+i, 4x+j] = z_step * sum_d d * softmax_d(cost[n, d, y, x, 4i+j]) ----
__global__ void cd_regress_fwd_kernel(const float* __restrict__ cost, float* __restrict__ pred, int N, int h, int w, float z_step) {
    const int H = 4 * h, W = 4 * w;
    const long total = (long)N * H * W;
    GRID_STRIDE(idx, total) {
        const int X = (int)(idx % W); long t_ = idx / W; const int Y = (int)(t_ % H); const int n = (int)(t_ / H);
        const int c = ((Y & 3) << 2) | (X & 3);
        const float* p = cost + ((((long)n * 16) * h + (Y >> 2)) * w + (X >> 2)) * 16 + c;
        const long ps = (long)h * w * 16;
        float v[16], m = -INFINITY;
#pragma unroll
        for (int d = 0; d < 16; ++d) { v[d] = p[d * ps]; m = fmaxf(m, v[d]); }
        float s = 0.f, e = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) { const float q = expf(v[d] - m); s += q; e += q * (float)d; }
        pred[idx] = (e / s) * z_step;
    }
}
__global__ void cd_regress_bwd_kernel(const float* __restrict__ cost, const float* __restrict__ gpred, float* __restrict__ gcost, int N, int h, int w, float z_step) {
    const int H = 4 * h, W = 4 * w;
    const long total = (long)N * H * W;
    GRID_STRIDE(idx, total) {
        const int X = (int)(idx % W); long t_ = idx / W; const int Y = (int)(t_ % H); const int n = (int)(t_ / H);
        const int c = ((Y & 3) << 2) | (X & 3);
        const long base = ((((long)n * 16) * h + (Y >> 2)) * w + (X >> 2)) * 16 + c;
        const long ps = (long)h * w * 16;
        float v[16], m = -INFINITY;
#pragma unroll
        for (int d = 0; d < 16; ++d) { v[d] = cost[base + d * ps]; m = fmaxf(m, v[d]); }
        float s = 0.f, e = 0.f;
#pragma unroll
        for (int d = 0; d < 16; ++d) { v[d] = expf(v[d] - m); s += v[d]; e += v[d] * (float)d; }
        const float E = e / s, g = gpred[idx] * z_step;
#pragma unroll
        for (int d = 0; d < 16; ++d) gcost[base + d * ps] = g * (v[d] / s) * ((float)d - E);
    }
}

// ---- heads' input rows (CD:243-251): feat (b, c, d, h, w) -> reshape(b, c*d, h, w) -> rows (b*h*w, c*d), column = c*D + d --------
__global__ void cd_rows_fwd_kernel(const float* __restrict__ feat, float* __restrict__ rows, int N, int D, int h, int w, int C) {
    const long P = (long)h * w, total = (long)N * P * C * D;
    GRID_STRIDE(idx, total) {
        const int col = (int)(idx % (C * D)); const long row = idx / (C * D);
        const int c = col / D, d = col % D; const int n = (int)(row / P); const long pix = row % P;
        rows[idx] = feat[(((long)n * D + d) * P + pix) * C + c];
    }
}
__global__ void cd_rows_bwd_kernel(const float* __restrict__ grows, float* __restrict__ gfeat, int N, int D, int h, int w, int C, int acc) {
    const long P = (long)h * w, total = (long)N * D * P * C;
    GRID_STRIDE(idx, total) {
        const int c = (int)(idx % C); long t_ = idx / C; const long pix = t_ % P; t_ /= P; const int d = (int)(t_ % D); const int n = (int)(t_ / D);
        const float g = grows[((long)n * P + pix) * (C * D) + c * D + d];
        gfeat[idx] = acc ? gfeat[idx] + g : g;
    }
}
}  // namespace

#define LAUNCH_OK() do { if (hipGetLastError() != hipSuccess) return -5; return 0; } while (0)

int cd_launch_pad_dual(const float* src, float* dst, int N, int C, int H, int W, int Hp, int Wp, int pt, int pr, hipStream_t s, int norm, float div,
                       const float* mean, const float* stdv) {
    CdNorm nm{0, 1.f, {0.f, 0.f, 0.f}, {1.f, 1.f, 1.f}};
    if (norm && C == 3) { nm.on = 1; nm.div = div; for (int k = 0; k < 3; ++k) { nm.mean[k] = mean[k]; nm.stdv[k] = stdv[k]; } }
    hipLaunchKernelGGL(cd_pad_dual_kernel, dim3(nbk((long)2 * N * C * Hp * Wp)), dim3(256), 0, s, src, dst, N, C, H, W, Hp, Wp, pt, pr, nm); LAUNCH_OK();
}
int cd_launch_crop_avg(const float* net, float* out, int N, int H, int W, int Hp, int Wp, int pt, int pr, hipStream_t s) {
    hipLaunchKernelGGL(cd_crop_avg_kernel, dim3(nbk((long)N * H * W)), dim3(256), 0, s, net, out, N, H, W, Hp, Wp, pt, pr); LAUNCH_OK();
}
int cd_launch_scatter_dual_grad(const float* g, float* gnet, int N, int H, int W, int Hp, int Wp, int pt, int pr, hipStream_t s) {
    hipLaunchKernelGGL(cd_scatter_dual_grad_kernel, dim3(nbk((long)2 * N * Hp * Wp)), dim3(256), 0, s, g, gnet, N, H, W, Hp, Wp, pt, pr); LAUNCH_OK();
}
int cd_launch_stage(const float* image, const float* sparse, float* out, int N, int passes, int H, int W, int C, int norm, float div, const float* mean,
                    const float* stdv, hipStream_t s) {
    if (C != 4 && C != 16) return -22;
    hipLaunchKernelGGL(cd_stage_kernel, dim3(nbk((long)passes * N * H * W)), dim3(256), 0, s, image, sparse, out, N, passes, H, W, C, norm, div, mean[0], mean[1],
                       mean[2], stdv[0], stdv[1], stdv[2]); LAUNCH_OK();
}
int cd_launch_clamp(const float* src, float* dst, long n, float maxd, hipStream_t s) {
    hipLaunchKernelGGL(cd_clamp_kernel, dim3(nbk(n)), dim3(256), 0, s, src, dst, n, maxd); LAUNCH_OK();
}

int cd_sparse_levels_build(const CdSparse& q, const float* sparse, float z_step, hipStream_t s) {
    const int N = q.N, H = q.H, W = q.W;
    for (int l = 0; l < 3; ++l)
        hipLaunchKernelGGL(cd_fill_int_kernel, dim3(nbk((long)N * 16 * (H >> l) * (W >> l))), dim3(256), 0, s, q.vol[l], (long)N * 16 * (H >> l) * (W >> l), -1);
    hipLaunchKernelGGL(cd_l0_rowcount_kernel, dim3((N * H + 3) / 4), dim3(256), 0, s, sparse, N, H, W, z_step, q.rowcnt);
    hipLaunchKernelGGL(cd_scan_kernel, dim3(1), dim3(1024), 0, s, q.rowcnt, N * H, q.rowoff, q.cnt + 0);
    hipLaunchKernelGGL(cd_l0_fill_kernel, dim3((N * H + 3) / 4), dim3(256), 0, s, sparse, N, H, W, z_step, q.rowoff, q.coords[0], q.feat_in, q.vol[0]);
    for (int l = 1; l < 3; ++l) {
        const int Hc = H >> l, Wc = W >> l, rows = N * 16 * Hc;
        hipLaunchKernelGGL(cd_mark_kernel, dim3(1024), dim3(256), 0, s, q.coords[l - 1], q.cnt + (l - 1), l, Hc, Wc, q.vol[l]);
        hipLaunchKernelGGL(cd_lc_rowcount_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, q.vol[l], rows, Wc, q.rowcnt);
        hipLaunchKernelGGL(cd_scan_kernel, dim3(1), dim3(1024), 0, s, q.rowcnt, rows, q.rowoff, q.cnt + l);
        hipLaunchKernelGGL(cd_lc_fill_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, q.vol[l], rows, Hc, Wc, l, q.rowoff, q.coords[l]);
    }
    LAUNCH_OK();
}
int cd_launch_sparse_conv(const CdSparse& q, const float* fin, int lin, int lout, const float* Wk, int ksize, int Ci, int Co, float* fout, hipStream_t s) {
    if (Co > 64) return -22;
    hipLaunchKernelGGL(cd_sparse_conv_kernel, dim3(4096), dim3(64), 0, s, fin, q.vol[lin], lin, q.H >> lin, q.W >> lin, q.coords[lout], q.cnt + lout, Wk, ksize,
                       Ci, Co, fout); LAUNCH_OK();
}
int cd_launch_sparse_bn(const CdSparse& q, const float* f, const float* res, int level, int C, const float* gamma, const float* beta, float* rm, float* rv,
                        long long* nbt, int train, int repeats, int relu, float* out, hipStream_t s, const PttaStatSync* sync, float* st) {
    if (C > 64) return -22;
    if (!st) st = q.bn_st;
    if (train) hipLaunchKernelGGL(cd_sbn_stats_kernel, dim3(SBN_BLOCKS), dim3(64), 0, s, f, q.cnt + level, C, q.bn_part);
    const double* gsum = nullptr;
    if (train && sync && sync->on()) {
        if (2 * C + 1 > sync->cap) return -22;
        hipLaunchKernelGGL(cd_sbn_collapse_kernel, dim3(1), dim3(64), 0, s, q.bn_part, q.cnt + level, C, sync->buf);
        const int rc = sync->exchange(2 * C + 1, s);
        if (rc) return rc;
        gsum = sync->buf;
    }
    hipLaunchKernelGGL(cd_sbn_finalize_kernel, dim3(1), dim3(64), 0, s, q.bn_part, q.cnt + level, C, gamma, beta, rm, rv, nbt, train, repeats, st, gsum);
    hipLaunchKernelGGL(cd_sbn_apply_kernel, dim3(2048), dim3(256), 0, s, f, res, q.cnt + level, C, st, relu, out); LAUNCH_OK();
}
int cd_launch_sparse_conv_bwd(const CdSparse& q, const float* gy, int lin, int lout, const float* Wk, int ksize, int Ci, int Co, float* gx, int acc, hipStream_t s) {
    if (Ci > 64) return -22;
    hipLaunchKernelGGL(cd_sparse_conv_bwd_kernel, dim3(4096), dim3(64), 0, s, gy, q.vol[lout], lin, lout, q.H >> lout, q.W >> lout, q.coords[lin], q.cnt + lin, Wk,
                       ksize, Ci, Co, gx, acc); LAUNCH_OK();
}
// y = relu?(bn(f) [+ res]) with the forward's saved st = [scale, shift, mean, inv][C]: dgamma, dbeta, gx (written), gres (the residual branch, optional)
int cd_launch_sparse_bn_bwd(const CdSparse& q, const float* f, const float* y, const float* g, int level, int C, const float* gamma, const float* st, int relu,
                            float* dgamma, float* dbeta, float* gx, float* gres, int acc_res, float* bw, hipStream_t s, const PttaStatSync* sync) {
    if (C > 64) return -22;
    hipLaunchKernelGGL(cd_sbn_bwd_stats_kernel, dim3(SBN_BLOCKS), dim3(64), 0, s, f, y, g, q.cnt + level, C, st, relu, q.bn_part);
    const double* gsum = nullptr; float gsc = 1.f;
    if (sync && sync->on()) {
        if (2 * C + 1 > sync->cap) return -22;
        hipLaunchKernelGGL(cd_sbn_bwd_collapse_kernel, dim3(1), dim3(64), 0, s, q.bn_part, q.cnt + level, C, sync->buf);
        const int rc = sync->exchange(2 * C + 1, s);
        if (rc) return rc;
        gsum = sync->buf; gsc = 1.f / (float)sync->world;
    }
    hipLaunchKernelGGL(cd_sbn_bwd_finalize_kernel, dim3(1), dim3(64), 0, s, q.bn_part, q.cnt + level, C, gamma, st, dgamma, dbeta, bw, gsum, gsc);
    hipLaunchKernelGGL(cd_sbn_bwd_apply_kernel, dim3(2048), dim3(256), 0, s, f, y, g, q.cnt + level, C, st, bw, relu, gx, gres, acc_res); LAUNCH_OK();
}
int cd_launch_densify_bwd(const CdSparse& q, const float* gvol, int ldv, int c0, int C, float* g, hipStream_t s) {
    hipLaunchKernelGGL(cd_densify_bwd_kernel, dim3(2048), dim3(256), 0, s, gvol, ldv, c0, q.coords[2], q.cnt + 2, C, q.H >> 2, q.W >> 2, g); LAUNCH_OK();
}
int cd_launch_densify(const CdSparse& q, const float* f, int C, float* dense, hipStream_t s) {
    const int h4 = q.H >> 2, w4 = q.W >> 2;
    if (hipMemsetAsync(dense, 0, (size_t)q.N * 16 * h4 * w4 * C * sizeof(float), s) != hipSuccess) return -5;
    hipLaunchKernelGGL(cd_densify_kernel, dim3(2048), dim3(256), 0, s, f, q.coords[2], q.cnt + 2, C, h4, w4, dense); LAUNCH_OK();
}
int cd_launch_fusion_fwd(const float* feat2d, const float* feat3d, float* vol, float* maskw, int N, int passes, int h, int w, hipStream_t s) {
    hipLaunchKernelGGL(cd_fusion_fwd_kernel, dim3(nbk((long)N * h * w)), dim3(256), 0, s, feat2d, feat3d, vol, maskw, N, passes, h, w); LAUNCH_OK();
}
int cd_launch_fusion_bwd(const float* gvol, const float* maskw, float* gfeat2d, int N, int h, int w, hipStream_t s) {
    hipLaunchKernelGGL(cd_fusion_bwd_kernel, dim3(nbk((long)N * h * w * 16)), dim3(256), 0, s, gvol, maskw, gfeat2d, N, h, w); LAUNCH_OK();
}
int cd_launch_pool_fwd(const float* x, float* y, long items_out, int H, int W, int C, hipStream_t s) {
    hipLaunchKernelGGL(cd_pool_fwd_kernel, dim3(nbk(items_out * (H >> 1) * (W >> 1) * (C >> 2))), dim3(256), 0, s, x, y, items_out, H, W, C); LAUNCH_OK();
}
int cd_launch_pool_bwd(const float* x, const float* gy, float* gx, long items_in, int H, int W, int C, int acc, hipStream_t s) {
    hipLaunchKernelGGL(cd_pool_bwd_kernel, dim3(nbk(items_in * H * W * C)), dim3(256), 0, s, x, gy, gx, items_in, H, W, C, acc); LAUNCH_OK();
}
int cd_launch_up_fwd(const float* x, float* y, long frames, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, hipStream_t s) {
    hipLaunchKernelGGL(cd_up_fwd_kernel, dim3(nbk(frames * Do * Ho * Wo * (C >> 2))), dim3(256), 0, s, x, y, frames, Di, Hi, Wi, Do, Ho, Wo, C); LAUNCH_OK();
}
int cd_launch_up_bwd(const float* gy, float* gx, long frames, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int acc, hipStream_t s) {
    hipLaunchKernelGGL(cd_up_bwd_kernel, dim3(nbk(frames * Di * Hi * Wi * C)), dim3(256), 0, s, gy, gx, frames, Di, Hi, Wi, Do, Ho, Wo, C, acc); LAUNCH_OK();
}
int cd_launch_regress_fwd(const float* cost, float* pred, int N, int h, int w, float z_step, hipStream_t s) {
    hipLaunchKernelGGL(cd_regress_fwd_kernel, dim3(nbk((long)N * 16 * h * w)), dim3(256), 0, s, cost, pred, N, h, w, z_step); LAUNCH_OK();
}
int cd_launch_regress_bwd(const float* cost, const float* gpred, float* gcost, int N, int h, int w, float z_step, hipStream_t s) {
    hipLaunchKernelGGL(cd_regress_bwd_kernel, dim3(nbk((long)N * 16 * h * w)), dim3(256), 0, s, cost, gpred, gcost, N, h, w, z_step); LAUNCH_OK();
}
int cd_launch_rows_fwd(const float* feat, float* rows, int N, int D, int h, int w, int C, hipStream_t s) {
    hipLaunchKernelGGL(cd_rows_fwd_kernel, dim3(nbk((long)N * h * w * C * D)), dim3(256), 0, s, feat, rows, N, D, h, w, C); LAUNCH_OK();
}
int cd_launch_rows_bwd(const float* grows, float* gfeat, int N, int D, int h, int w, int C, int acc, hipStream_t s) {
    hipLaunchKernelGGL(cd_rows_bwd_kernel, dim3(nbk((long)N * h * w * C * D)), dim3(256), 0, s, grows, gfeat, N, D, h, w, C, acc); LAUNCH_OK();
}
