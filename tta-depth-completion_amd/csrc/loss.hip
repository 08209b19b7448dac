// ProxyTTA adaptation loss and its analytic gradients.
//
// Follows ExternalModel_Adapt.adapt_loss (src/external_model_adapt.py:371-441):
//   L = w_sd * sparse_depth_consistency (src/loss_utils.py:116-137)
//     + w_sm * edge-aware smoothness     (src/loss_utils.py:139-169, gradient_yx :624-638)
//     + w_cos * mean_rows(2 - 2 <e/|e|, r/|r|>)   (F.normalize eps 1e-12, :421-423)
//   with the data-dependent gate  L_cos < 0.3  =>  w_cos = 0  (:424-425) evaluated ON DEVICE
//   (the reference pays a device->host sync for it).
// Reductions are wave-shuffle + LDS per block, then a fixed-order second stage (bitwise
// reproducible, no float atomics).  Gradients w.r.t. the depth map and w.r.t. `reference`
// replace autograd's backward through these ops (src/tta_main.py:632).
#include <cstdlib>
#include "ptta_common.h"
#include "ptta_kernels.h"

#define LOSS_PB 256          // depth-reduction blocks per sample (<= 256: one finalize thread each)
#define LOSS_CB_MIN 1024     // cosine-reduction blocks (and partial slots) up to 131,072 rows; beyond: one slot per 128-row GEMM block
#define WS_SCAL 0            // [0] coef_cos [1] coef_smx [2] coef_smy
#define WS_SD 16             // per-sample w_sd / (N * sum_w)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

static __host__ __device__ inline long ws_depth_off(int N) { return WS_SD + ((N + 15) / 16) * 16; }
// slots of the cosine term's block partials: the narrow heads' GEMM epilogue leaves one per 128-row block (heads_n.hip, epi 5), so the count
// follows the rows (N frames per call: 209 N blocks at 352x1216); never below the 1024 the row kernels of this file launch with
static __host__ __device__ inline int loss_cb(long R) { const long b = (R + 127) / 128; return b <= LOSS_CB_MIN ? LOSS_CB_MIN : (int)((b + 255) & ~255L); }
static __host__ __device__ inline long ws_cnt_off(int N) { return ws_depth_off(N) + (long)N * LOSS_PB * 4 + LOSS_CB_MIN; }     // [N][LOSS_PB] valid-weight partials (loss_valid_count_kernel)
static __host__ __device__ inline long ws_rows_off(int N) { return ws_cnt_off(N) + (long)N * LOSS_PB; }
static __host__ __device__ inline long ws_cos_off(int N, long R) { return ws_rows_off(N) + 3 * R + 64; }                        // [loss_cb(R)] behind the per-row statistics

int ptta_loss_ws_floats(int N, int H, int W, long R) { return (int)(ws_cos_off(N, R) + loss_cb(R) + 64); }
long ptta_loss_ws_rows_off(int N) { return ws_rows_off(N); }
long ptta_loss_ws_cos_off(int N, long R) { return ws_cos_off(N, R); }
int ptta_loss_cos_blocks(long R) { return loss_cb(R); }

// validity_map of the TTA step = where(sparse > 0, 1, sparse) on the RAW sparse depth (src/tta_main.py:583-586); computed on the
// fly when the caller passes no map
__device__ __forceinline__ float valid_w(const float* __restrict__ validity, const float* __restrict__ sparse, size_t i) {
    if (validity) return validity[i];
    const float s = sparse[i];
    return s > 0.f ? 1.f : s;
}
__device__ __forceinline__ float clampd(float d, float max_d) { return max_d >= 0.f ? fminf(fmaxf(d, 0.f), max_d) : d; }
__device__ __forceinline__ float edge_w(const float* img, size_t plane, size_t a, size_t b) {
    const float m = (fabsf(img[a] - img[b]) + fabsf(img[plane + a] - img[plane + b]) +
                     fabsf(img[2 * plane + a] - img[2 * plane + b])) / 3.0f;
    return expf(-m);
}
// the same weight from values already in registers (a = the pixel, b = its neighbour; same operation order)
__device__ __forceinline__ float edge_w3(float a0, float a1, float a2, float b0, float b1, float b2) {
    const float m = (fabsf(a0 - b0) + fabsf(a1 - b1) + fabsf(a2 - b2)) / 3.0f;
    return expf(-m);
}

// bx / nbx: this block's index and the block count of the reduction (the merged launches below give each loss its own block range)
__device__ __forceinline__ void depth_reduce_body(int bx, int nbx, int n, const float* __restrict__ depth, const float* __restrict__ image,
                                                  const float* __restrict__ sparse, const float* __restrict__ validity,
                                                  float max_d, int H, int W, float* __restrict__ part) {
    __shared__ float red[4][4];
    const size_t plane = (size_t)H * W;
    const float* D = depth + n * plane;
    const float* S = sparse + n * plane;
    const float* V = validity ? validity + n * plane : nullptr;
    const float* I = image + (size_t)n * 3 * plane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (size_t idx = (size_t)bx * blockDim.x + threadIdx.x; idx < plane; idx += (size_t)nbx * blockDim.x) {
        const int x = (int)(idx % W), y = (int)(idx / W);
        // every load of the pixel issued before the first use (clamped neighbours, selected afterwards): behind `if (x < W - 1)` the
        // compiler waits for each neighbour's loads in turn
        const bool xr = x < W - 1, yd = y < H - 1;
        const size_t ir = xr ? idx + 1 : idx, id = yd ? idx + W : idx;
        const float d = D[idx], dr = D[ir], dd = D[id], sv = S[idx], vv = (V ? V : S)[idx];      // (one address for either case: no branch between the loads)
        const float w = V ? vv : (sv > 0.f ? 1.f : sv);
        const float c0 = I[idx], c1 = I[plane + idx], c2 = I[2 * plane + idx];
        const float r0 = I[ir], r1 = I[plane + ir], r2 = I[2 * plane + ir];
        const float e0 = I[id], e1 = I[plane + id], e2 = I[2 * plane + id];
        a0 += w * fabsf(clampd(sv, max_d) - d);
        a1 += w;
        a2 += xr ? edge_w3(c0, c1, c2, r0, r1, r2) * fabsf(d - dr) : 0.f;
        a3 += yd ? edge_w3(c0, c1, c2, e0, e1, e2) * fabsf(d - dd) : 0.f;
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; red[wave][3] = a3; }
    __syncthreads();
    if (threadIdx.x < 4)
        part[((size_t)n * LOSS_PB + bx) * 4 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ __launch_bounds__(256) void loss_depth_reduce_kernel(const float* __restrict__ depth, const float* __restrict__ image,
                                                                const float* __restrict__ sparse, const float* __restrict__ validity,
                                                                float max_d, int H, int W, float* __restrict__ part) {
    depth_reduce_body(blockIdx.x, gridDim.x, blockIdx.y, depth, image, sparse, validity, max_d, H, W, part);
}

// The sum of the validity weights alone -- the ONLY data-dependent input of the depth gradient's coefficients (w_sd / (N sum_w); the
// smoothness coefficients are constants) and a function of the step's INPUTS, not of the depth map: the fused step computes it at the start of
// the auxiliary stream's work, and the depth gradient then starts behind decoder 3 without waiting for the reduction of the loss VALUES
// (depth_reduce_body above, ~11 us), which moves off the critical path to the auxiliary stream.  Same decomposition and the same summation
// order as the a1 accumulator of depth_reduce_body: the partials -- and with them the coefficients and the gradient -- are bit-identical.
__global__ __launch_bounds__(256) void loss_valid_count_kernel(const float* __restrict__ sparse, const float* __restrict__ validity, int H, int W,
                                                               float* __restrict__ cnt) {
    __shared__ float red[4];
    const int bx = blockIdx.x, nbx = gridDim.x, n = blockIdx.y;
    const size_t plane = (size_t)H * W;
    const float* S = sparse + n * plane;
    const float* V = validity ? validity + n * plane : nullptr;
    float a1 = 0.f;
    for (size_t idx = (size_t)bx * blockDim.x + threadIdx.x; idx < plane; idx += (size_t)nbx * blockDim.x) {
        const float sv = S[idx], vv = (V ? V : S)[idx];
        a1 += V ? vv : (sv > 0.f ? 1.f : sv);
    }
    a1 = wave_sum(a1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a1;
    __syncthreads();
    if (threadIdx.x == 0) cnt[(size_t)n * LOSS_PB + bx] = red[0] + red[1] + red[2] + red[3];
}
int ptta_launch_loss_valid_count(const float* sparse, const float* validity, int N, int H, int W, float* ws, hipStream_t s) {
    hipLaunchKernelGGL(loss_valid_count_kernel, dim3(LOSS_PB, N), dim3(256), 0, s, sparse, validity, H, W, ws + ws_cnt_off(N));
    PTTA_CHECK_LAUNCH();
    return 0;
}

// one wave per row of `emb`/`ref` (D = 512: 8 values per lane); T = bf16_t: the narrow embeddings of the mixed mode (one 16-B load per tensor)
template <typename T>
__device__ __forceinline__ void cos_rows_body(int bx, int nbx, const T* __restrict__ emb, const T* __restrict__ ref, long R, int D,
                                              float* __restrict__ rowstats, float* __restrict__ part) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (long row = (long)bx * 4 + wave; row < R; row += (long)nbx * 4) {
        const T* e = emb + row * D;
        const T* r = ref + row * D;
        float ee = 0.f, rr = 0.f, er = 0.f;
        if constexpr (sizeof(T) == 4) {
            for (int k = 4 * lane; k < D; k += 256) {
                const float4 a = *(const float4*)(e + k), b = *(const float4*)(r + k);
                ee += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
                rr += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
                er += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
            }
        } else {
            for (int k = 8 * lane; k < D; k += 512) {
                const uint4 ua = *(const uint4*)(e + k), ub = *(const uint4*)(r + k);
                const float4 a0 = bf4_to_f4(make_uint2(ua.x, ua.y)), a1 = bf4_to_f4(make_uint2(ua.z, ua.w));
                const float4 b0 = bf4_to_f4(make_uint2(ub.x, ub.y)), b1 = bf4_to_f4(make_uint2(ub.z, ub.w));
                ee += a0.x * a0.x + a0.y * a0.y + a0.z * a0.z + a0.w * a0.w + a1.x * a1.x + a1.y * a1.y + a1.z * a1.z + a1.w * a1.w;
                rr += b0.x * b0.x + b0.y * b0.y + b0.z * b0.z + b0.w * b0.w + b1.x * b1.x + b1.y * b1.y + b1.z * b1.z + b1.w * b1.w;
                er += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w + a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
            }
        }
        ee = wave_sum(ee); rr = wave_sum(rr); er = wave_sum(er);
        const float ne = fmaxf(sqrtf(ee), 1e-12f), nr = fmaxf(sqrtf(rr), 1e-12f);
        const float c = er / (ne * nr);
        if (lane == 0) { rowstats[3 * row] = ne; rowstats[3 * row + 1] = nr; rowstats[3 * row + 2] = c; }
        acc += 2.f - 2.f * c;
    }
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[bx] = red[0] + red[1] + red[2] + red[3];
}
template <typename T>
__global__ __launch_bounds__(256) void cos_rows_kernel(const T* __restrict__ emb, const T* __restrict__ ref, long R, int D,
                                                       float* __restrict__ rowstats, float* __restrict__ part) {
    cos_rows_body<T>(blockIdx.x, gridDim.x, emb, ref, R, D, rowstats, part);
}
// both reductions of the step in ONE launch: blocks [0, N * LOSS_PB) the depth terms (latency-bound: 1.7 MB per frame), the rest the cosine
// rows (bandwidth-bound: 110 MB) -- the short one hides inside the long one, and one launch less on the critical path
__global__ __launch_bounds__(256) void loss_forward_merged_kernel(const float* __restrict__ depth, const float* __restrict__ image,
                                                                  const float* __restrict__ sparse, const float* __restrict__ validity,
                                                                  float max_d, int N, int H, int W, float* __restrict__ dpart,
                                                                  const float* __restrict__ emb, const float* __restrict__ ref, long R, int D,
                                                                  float* __restrict__ rowstats, float* __restrict__ cpart) {
    const int nd = N * LOSS_PB;
    if ((int)blockIdx.x < nd) depth_reduce_body(blockIdx.x % LOSS_PB, LOSS_PB, blockIdx.x / LOSS_PB, depth, image, sparse, validity, max_d, H, W, dpart);
    else cos_rows_body<float>(blockIdx.x - nd, gridDim.x - nd, emb, ref, R, D, rowstats, cpart);
}

// one block: parallel fixed-shape reductions of the partials (deterministic), then thread 0 finishes
__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// The ONE finalisation of the loss (sparse-depth / smoothness / cosine terms, the gate of external_model_adapt.py:424-425, the gradient
// coefficients).  The fused step runs it redundantly in EVERY block of its two gradient kernels (a few KB of partials from L2 per block)
// instead of as a 1-block launch between them: one ~20 us latency-bound kernel less on the step's critical path; the split API
// (ptta_loss_forward) runs it as loss_finalize_kernel below.  fin (LDS): [0] coef_cos [1] cx [2] cy [3..3+N) per-sample sparse-depth
// coefficient (first LOSS_FIN_MAXN samples); with `loss_info` given, the scalars and the coefficients also go to the workspace.
#define LOSS_FIN_MAXN 16
__device__ void loss_finalize_block(const float* __restrict__ ws, int N, int H, int W, long R, int has_cos, const float* __restrict__ w3,
                                    float* __restrict__ fin, float* __restrict__ loss_info, float* __restrict__ ws_out) {
    __shared__ double red[4];
    const float w_sd = w3[0], w_sm = w3[1], w_cos = w3[2];
    const float* dp = ws + ws_depth_off(N);
    const int t = threadIdx.x;
    double l_sd = 0.0, smx = 0.0, smy = 0.0;
    for (int n = 0; n < N; ++n) {
        double q0 = 0, q1 = 0, q2 = 0, q3 = 0;
        if (t < LOSS_PB) { const float* q = dp + ((size_t)n * LOSS_PB + t) * 4; q0 = q[0]; q1 = q[1]; q2 = q[2]; q3 = q[3]; }
        const double num = block_sum_d(q0, red), den = block_sum_d(q1, red);
        smx += block_sum_d(q2, red); smy += block_sum_d(q3, red);
        l_sd += num / den;                                   // NaN if a sample has no valid point, as the reference
        if (t == 0) {
            const float sdn = (float)((double)w_sd / ((double)N * den));
            if (n < LOSS_FIN_MAXN) fin[3 + n] = sdn;
            if (loss_info) ws_out[WS_SD + n] = sdn;
        }
    }
    l_sd /= N;
    const double cntx = (double)N * H * (W - 1), cnty = (double)N * (H - 1) * W;
    const double l_sm = smx / cntx + smy / cnty;
    double l_cos = 0.0;
    float wc = w_cos;
    if (has_cos) {
        const float* cp = ws + ws_cos_off(N, R);
        double a = 0.0;
        for (int b = t; b < loss_cb(R); b += 256) a += cp[b];
        l_cos = block_sum_d(a, red) / (double)R;
        if ((float)l_cos < 0.3f) wc = 0.f;                   // external_model_adapt.py:424-425
    }
    if (t == 0) {
        fin[0] = has_cos ? (float)(-2.0 * wc / (double)R) : 0.f;
        fin[1] = (float)(w_sm / cntx);
        fin[2] = (float)(w_sm / cnty);
        if (loss_info) {
            loss_info[0] = (float)(w_sd * l_sd + w_sm * l_sm + (double)wc * l_cos);
            loss_info[1] = (float)l_sm; loss_info[2] = (float)l_sd; loss_info[3] = (float)l_cos;
            ws_out[WS_SCAL + 0] = fin[0]; ws_out[WS_SCAL + 1] = fin[1]; ws_out[WS_SCAL + 2] = fin[2];
        }
    }
    __syncthreads();
}

// the depth gradient's coefficients from the valid-weight partials alone (loss_valid_count_kernel): fin[1] cx, fin[2] cy, fin[3 + n] the
// per-sample sparse-depth coefficient -- the same expressions, on the same sums, as loss_finalize_block
__device__ void loss_coef_block(const float* __restrict__ cnt, int N, int H, int W, const float* __restrict__ w3, float* __restrict__ fin) {
    __shared__ double red[4];
    const float w_sd = w3[0], w_sm = w3[1];
    const int t = threadIdx.x;
    for (int n = 0; n < N; ++n) {
        const double q1 = t < LOSS_PB ? (double)cnt[(size_t)n * LOSS_PB + t] : 0.0;
        const double den = block_sum_d(q1, red);
        if (t == 0 && n < LOSS_FIN_MAXN) fin[3 + n] = (float)((double)w_sd / ((double)N * den));
    }
    const double cntx = (double)N * H * (W - 1), cnty = (double)N * (H - 1) * W;
    if (t == 0) { fin[0] = 0.f; fin[1] = (float)(w_sm / cntx); fin[2] = (float)(w_sm / cnty); }
    __syncthreads();
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(float* __restrict__ ws, int N, int H, int W, long R, int has_cos,
                                                            const float* __restrict__ w3, float* __restrict__ loss_info) {
    __shared__ float fin[3 + LOSS_FIN_MAXN];
    loss_finalize_block(ws, N, H, W, R, has_cos, w3, fin, loss_info, ws);
}

// workspace [0] = the cosine gradient's coefficient as loss_finalize_block forms it (same partials, same reduction, same gate) without the
// depth terms: the heads' backward starts as soon as the cosine rows are reduced, beside decoder 3's forward
__global__ __launch_bounds__(256) void loss_cos_coef_kernel(float* __restrict__ ws, int N, long R, const float* __restrict__ w3) {
    __shared__ double red[4];
    const float* cp = ws + ws_cos_off(N, R);
    double a = 0.0;
    for (int b = threadIdx.x; b < loss_cb(R); b += 256) a += cp[b];
    const double l_cos = block_sum_d(a, red) / (double)R;
    float wc = w3[2];
    if ((float)l_cos < 0.3f) wc = 0.f;                       // external_model_adapt.py:424-425
    if (threadIdx.x == 0) ws[WS_SCAL + 0] = (float)(-2.0 * wc / (double)R);
}
int ptta_launch_loss_cos_coef(float* ws, int N, long R, const float* w3_dev, hipStream_t s) {
    hipLaunchKernelGGL(loss_cos_coef_kernel, dim3(1), dim3(256), 0, s, ws, N, R, w3_dev);
    PTTA_CHECK_LAUNCH();
    return 0;
}

int ptta_launch_loss_forward(const float* depth, const float* image, const float* sparse, const float* validity,
                             float max_input_depth, const float* emb, const float* ref, long R, int D,
                             const float* w3_dev, int N, int H, int W,
                             float* ws, float* loss_info, hipStream_t s, int defer_finalize) {
    const int has_cos = (emb && ref) ? 1 : 0;
    if (has_cos) {
        hipLaunchKernelGGL(loss_forward_merged_kernel, dim3(N * LOSS_PB + loss_cb(R)), dim3(256), 0, s, depth, image, sparse, validity, max_input_depth,
                           N, H, W, ws + ws_depth_off(N), emb, ref, R, D, ws + ws_rows_off(N), ws + ws_cos_off(N, R));
    } else {
        hipLaunchKernelGGL(loss_depth_reduce_kernel, dim3(LOSS_PB, N), dim3(256), 0, s, depth, image, sparse, validity,
                           max_input_depth, H, W, ws + ws_depth_off(N));
        if (has_cos)
            hipLaunchKernelGGL((cos_rows_kernel<float>), dim3(loss_cb(R)), dim3(256), 0, s, emb, ref, R, D, ws + ws_rows_off(N), ws + ws_cos_off(N, R));
    }
    // defer_finalize (fused step, N <= LOSS_FIN_MAXN): ptta_launch_loss_backward(..., w3_dev, loss_info) finalises inside its kernels
    if (!defer_finalize || N > LOSS_FIN_MAXN)
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, ws, N, H, W, R, has_cos, w3_dev, loss_info);
    PTTA_CHECK_LAUNCH();
    return 0;
}

// The two halves of ptta_launch_loss_forward as separate launches and the finalisation as its own (the fused step's two branches, ptta_api.hip
// step_tail: the depth terms behind decoder 3 on one stream, the cosine rows behind the heads on the other -- neither waits for the other's
// forward): same kernels, same partials, same finalisation.
int ptta_launch_loss_depth_part(const float* depth, const float* image, const float* sparse, const float* validity, float max_input_depth,
                                int N, int H, int W, float* ws, hipStream_t s) {
    hipLaunchKernelGGL(loss_depth_reduce_kernel, dim3(LOSS_PB, N), dim3(256), 0, s, depth, image, sparse, validity, max_input_depth, H, W, ws + ws_depth_off(N));
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_launch_loss_cos_part(const void* emb, const void* ref, long R, int D, int N, float* ws, hipStream_t s, int narrow) {
    if (narrow) hipLaunchKernelGGL((cos_rows_kernel<bf16_t>), dim3(loss_cb(R)), dim3(256), 0, s, (const bf16_t*)emb, (const bf16_t*)ref, R, D, ws + ws_rows_off(N), ws + ws_cos_off(N, R));
    else hipLaunchKernelGGL((cos_rows_kernel<float>), dim3(loss_cb(R)), dim3(256), 0, s, (const float*)emb, (const float*)ref, R, D, ws + ws_rows_off(N), ws + ws_cos_off(N, R));
    PTTA_CHECK_LAUNCH();
    return 0;
}
int ptta_launch_loss_finalize(float* ws, int N, int H, int W, long R, int has_cos, const float* w3_dev, float* loss_info, hipStream_t s) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, ws, N, H, W, R, has_cos, w3_dev, loss_info);
    PTTA_CHECK_LAUNCH();
    return 0;
}

__device__ __forceinline__ void depth_grad_body(int bx, int nbx, const float* __restrict__ depth, const float* __restrict__ image,
                                                const float* __restrict__ sparse, const float* __restrict__ validity,
                                                float max_d, int N, int H, int W, float* __restrict__ ws,
                                                float* __restrict__ g, long R, int has_cos, const float* __restrict__ w3,
                                                float* __restrict__ loss_info, const float* __restrict__ cnt = nullptr) {
    __shared__ float fin[3 + LOSS_FIN_MAXN];
    const size_t plane = (size_t)H * W;
    const size_t total = (size_t)N * plane;
    if (w3 && cnt) loss_coef_block(cnt, N, H, W, w3, fin);           // fused step, the loss values reduced elsewhere (step_tail, ptta_api.hip)
    else if (w3) loss_finalize_block(ws, N, H, W, R, has_cos, w3, fin, bx == 0 ? loss_info : nullptr, ws);       // fused step
    const float cx = w3 ? fin[1] : ws[WS_SCAL + 1], cy = w3 ? fin[2] : ws[WS_SCAL + 2];
    for (size_t gi = (size_t)bx * blockDim.x + threadIdx.x; gi < total; gi += (size_t)nbx * blockDim.x) {
        const int n = (int)(gi / plane);
        const size_t idx = gi % plane;
        const int x = (int)(idx % W), y = (int)(idx / W);
        const float* D = depth + n * plane;
        const float* I = image + (size_t)n * 3 * plane;
        // all 22 loads of the pixel in flight together (clamped neighbours, selected afterwards; see depth_reduce_body)
        const bool xr = x < W - 1, xl = x > 0, yd = y < H - 1, yu = y > 0;
        const size_t ir = xr ? idx + 1 : idx, il = xl ? idx - 1 : idx, id = yd ? idx + W : idx, iu = yu ? idx - W : idx;
        const float d = D[idx], dr = D[ir], dl = D[il], dd = D[id], du = D[iu];
        const float sv = sparse[gi], vv = (validity ? validity : sparse)[gi];
        const float c0 = I[idx], c1 = I[plane + idx], c2 = I[2 * plane + idx];
        const float r0 = I[ir], r1 = I[plane + ir], r2 = I[2 * plane + ir];
        const float l0 = I[il], l1 = I[plane + il], l2 = I[2 * plane + il];
        const float e0 = I[id], e1 = I[plane + id], e2 = I[2 * plane + id];
        const float u0 = I[iu], u1 = I[plane + iu], u2 = I[2 * plane + iu];
        const float sd_n = w3 ? fin[3 + n] : ws[WS_SD + n];
        const float vw = validity ? vv : (sv > 0.f ? 1.f : sv);
        float v = sd_n * vw * sgn(d - clampd(sv, max_d));
        float tx = 0.f, ty = 0.f;
        tx += xr ? edge_w3(c0, c1, c2, r0, r1, r2) * sgn(d - dr) : 0.f;
        tx -= xl ? edge_w3(l0, l1, l2, c0, c1, c2) * sgn(dl - d) : 0.f;
        ty += yd ? edge_w3(c0, c1, c2, e0, e1, e2) * sgn(d - dd) : 0.f;
        ty -= yu ? edge_w3(u0, u1, u2, c0, c1, c2) * sgn(du - d) : 0.f;
        g[gi] = v + cx * tx + cy * ty;
    }
}
__global__ __launch_bounds__(256) void loss_depth_grad_kernel(const float* __restrict__ depth, const float* __restrict__ image,
                                                              const float* __restrict__ sparse, const float* __restrict__ validity,
                                                              float max_d, int N, int H, int W, float* __restrict__ ws,
                                                              float* __restrict__ g, long R, int has_cos, const float* __restrict__ w3,
                                                              float* __restrict__ loss_info, const float* __restrict__ cnt) {
    depth_grad_body(blockIdx.x, gridDim.x, depth, image, sparse, validity, max_d, N, H, W, ws, g, R, has_cos, w3, loss_info, cnt);
}

__device__ __forceinline__ void cos_grad_body(int bx, int nbx, const float* __restrict__ emb, const float* __restrict__ ref, long R, int D,
                                              const float* __restrict__ ws, const float* __restrict__ rowstats,
                                              float* __restrict__ gref, int N, int H, int W, const float* __restrict__ w3) {
    __shared__ float fin[3 + LOSS_FIN_MAXN];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (w3) loss_finalize_block(ws, N, H, W, R, 1, w3, fin, nullptr, nullptr);
    const float coef = w3 ? fin[0] : ws[WS_SCAL + 0];
    for (long row = (long)bx * 4 + wave; row < R; row += (long)nbx * 4) {
        const float ne = rowstats[3 * row], nr = rowstats[3 * row + 1], c = rowstats[3 * row + 2];
        const float ie = 1.f / ne, ir = 1.f / nr;
        // d(r/max(|r|,eps))/dr = (I - rhat rhat^T)/|r| above eps, I/eps below it
        const float proj = (nr > 1e-12f) ? c : 0.f;
        for (int k = 4 * lane; k < D; k += 256) {
            const float4 a = *(const float4*)(emb + row * D + k), b = *(const float4*)(ref + row * D + k);
            float4 o;
            o.x = cos_grad_elem(coef, a.x, b.x, ie, ir, proj);
            o.y = cos_grad_elem(coef, a.y, b.y, ie, ir, proj);
            o.z = cos_grad_elem(coef, a.z, b.z, ie, ir, proj);
            o.w = cos_grad_elem(coef, a.w, b.w, ie, ir, proj);
            *(float4*)(gref + row * D + k) = o;
        }
    }
}
__global__ __launch_bounds__(256) void cos_grad_kernel(const float* __restrict__ emb, const float* __restrict__ ref, long R, int D,
                                                       const float* __restrict__ ws, const float* __restrict__ rowstats,
                                                       float* __restrict__ gref, int N, int H, int W, const float* __restrict__ w3) {
    cos_grad_body(blockIdx.x, gridDim.x, emb, ref, R, D, ws, rowstats, gref, N, H, W, w3);
}
// both gradients in ONE launch (see loss_forward_merged_kernel): blocks [0, db) the depth gradient, the rest the cosine gradient
__global__ __launch_bounds__(256) void loss_backward_merged_kernel(const float* __restrict__ depth, const float* __restrict__ image,
                                                                   const float* __restrict__ sparse, const float* __restrict__ validity,
                                                                   float max_d, int N, int H, int W, float* __restrict__ ws, float* __restrict__ g,
                                                                   const float* __restrict__ emb, const float* __restrict__ ref, long R, int D,
                                                                   float* __restrict__ gref, const float* __restrict__ w3, float* __restrict__ loss_info, int db) {
    if ((int)blockIdx.x < db) depth_grad_body(blockIdx.x, db, depth, image, sparse, validity, max_d, N, H, W, ws, g, R, 1, w3, loss_info);
    else cos_grad_body(blockIdx.x - db, gridDim.x - db, emb, ref, R, D, ws, ws + ws_rows_off(N), gref, N, H, W, w3);
}

int ptta_launch_loss_backward(const float* depth, const float* image, const float* sparse, const float* validity,
                              float max_input_depth, const float* emb, const float* ref, long R, int D,
                              int N, int H, int W, float* ws, float* gdepth, float* gref, hipStream_t s, const float* w3_fused,
                              float* loss_info_fused, int cos_partials_ready, int valid_count_ready) {
    const size_t total = (size_t)N * H * W;
    if (w3_fused && N > LOSS_FIN_MAXN) w3_fused = nullptr;          // the forward launched the finalize kernel in that case
    int blocks = (int)((total + 255) / 256); if (blocks > (w3_fused ? 1024 : 4096)) blocks = w3_fused ? 1024 : 4096;
    const int has_cos = ((emb && ref) || cos_partials_ready) ? 1 : 0;
    if (emb && ref && gref && w3_fused) {
        long cb = (R + 3) / 4; if (cb > 1024) cb = 1024;
        if (blocks > 512) blocks = 512;
        hipLaunchKernelGGL(loss_backward_merged_kernel, dim3(blocks + (int)cb), dim3(256), 0, s, depth, image, sparse, validity, max_input_depth, N, H, W,
                           ws, gdepth, emb, ref, R, D, gref, w3_fused, loss_info_fused, blocks);
        PTTA_CHECK_LAUNCH();
        return 0;
    }
    // valid_count_ready (fused step only): the coefficients come from ptta_launch_loss_valid_count's partials; nothing of the loss VALUES is read
    // or written here (ptta_launch_loss_depth_part + ptta_launch_loss_finalize report them, on another stream)
    const float* cnt = (valid_count_ready && w3_fused && !(emb && ref)) ? ws + ws_cnt_off(N) : nullptr;
    hipLaunchKernelGGL(loss_depth_grad_kernel, dim3(blocks), dim3(256), 0, s, depth, image, sparse, validity,
                       max_input_depth, N, H, W, ws, gdepth, R, has_cos, w3_fused, cnt ? nullptr : loss_info_fused, cnt);
    if (emb && ref && gref) {
        long cb = (R + 3) / 4; if (cb > (w3_fused ? 1024 : 2048)) cb = w3_fused ? 1024 : 2048;
        hipLaunchKernelGGL(cos_grad_kernel, dim3((int)cb), dim3(256), 0, s, emb, ref, R, D, ws, ws + ws_rows_off(N), gref, N, H, W, w3_fused);
    }
    PTTA_CHECK_LAUNCH();
    return 0;
}

// ---- validation metrics on device (src/tta_main.py:779-798 + src/eval_utils.py:117-174) --------------
// mask = gt > 0 and min_eval <= gt <= max_eval; MAE / RMSE in millimetres, iMAE / iRMSE in 1/km, over
// the masked pixels of the whole batch.  Replaces the D2H copy of a full depth map per frame
// (tta_main.py:769-770) with a 16-byte result.
#define MET_BLOCKS 256
__global__ __launch_bounds__(256) void eval_metrics_reduce_kernel(const float* __restrict__ depth, const float* __restrict__ gt, long n,
                                                                  float min_eval, float max_eval, double* __restrict__ part) {
    __shared__ double red[4][5];
    double a[5] = {0, 0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float g = gt[i];
        if (!(g > 0.f) || g < min_eval || g > max_eval) continue;
        const float o = depth[i];
        const float d = 1000.0f * g - 1000.0f * o;
        const float id = 1.0f / (0.001f * g + 1e-9f) - 1.0f / (0.001f * o + 1e-9f);
        a[0] += 1.0; a[1] += fabsf(d); a[2] += (double)d * d; a[3] += fabsf(id); a[4] += (double)id * id;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_xor(a[k], o);
    }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 5; ++k) red[threadIdx.x >> 6][k] = a[k];
    __syncthreads();
    if (threadIdx.x < 5) part[blockIdx.x * 5 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void eval_metrics_finalize_kernel(const double* __restrict__ part, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    double s[5] = {0, 0, 0, 0, 0};
    for (int b = 0; b < MET_BLOCKS; ++b)
        for (int k = 0; k < 5; ++k) s[k] += part[b * 5 + k];
    out[0] = (float)(s[1] / s[0]);              // MAE   [mm]
    out[1] = (float)sqrt(s[2] / s[0]);          // RMSE  [mm]
    out[2] = (float)(s[3] / s[0]);              // iMAE  [1/km]
    out[3] = (float)sqrt(s[4] / s[0]);          // iRMSE [1/km]
}
int ptta_launch_eval_metrics(const float* depth, const float* gt, long n, float min_eval, float max_eval, double* scratch, float* out4,
                             hipStream_t s) {
    hipLaunchKernelGGL(eval_metrics_reduce_kernel, dim3(MET_BLOCKS), dim3(256), 0, s, depth, gt, n, min_eval, max_eval, scratch);
    hipLaunchKernelGGL(eval_metrics_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, out4);
    PTTA_CHECK_LAUNCH();
    return 0;
}
