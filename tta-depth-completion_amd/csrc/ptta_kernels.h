// Host-side launch interface of the kernel translation units (internal; the public C-ABI is
// include/ptta.h).  All pointers are device pointers; every launcher only enqueues on `s`.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include "ptta_common.h"

// ---- the library's whole environment ---------------------------------------------------------------
// Three validation switches, read here and nowhere else, once per ptta_create: they decide what a handle allocates and which
// arithmetic its kernels run, so they cannot be options of a live handle.  Every other switch is ptta_set_option (include/ptta.h).
//   PTTA_CONV_IMPL=naive   direct fp32 kernels everywhere (the check of the matrix-core kernels)
//   PTTA_ARITH=exact       fp32 MFMA instead of bf16x3 (MSG_CHN, fp32 mode)
//   PTTA_GRAPH=0|1         initial value of the "graph" option (default 0: direct launches)
struct PttaCreateEnv { int naive = 0, exact = 0, graph = -1; };
inline PttaCreateEnv ptta_create_env() {
    PttaCreateEnv e;
    const char* v = getenv("PTTA_CONV_IMPL"); e.naive = (v && strcmp(v, "naive") == 0) ? 1 : 0;
    v = getenv("PTTA_ARITH"); e.exact = (v && strcmp(v, "exact") == 0) ? 1 : 0;
    v = getenv("PTTA_GRAPH"); e.graph = (v && (strcmp(v, "0") == 0 || strcmp(v, "1") == 0)) ? (v[0] - '0') : -1;
    return e;
}

// ---- conv32.hip ------------------------------------------------------------------------------
struct ConvW {           // one packed 3x3 32->32 filter, three formats (device memory)
    float* mf32;         // fp32 MFMA fragments  [9][4][64][4]
    bf16_t* mbf16;       // bf16 MFMA fragments  [9][2][64][8] (= hi half of the bf16x3 split)
    bf16_t* mlo;         // lo half of the bf16x3 split, same layout
    float* canon;        // [tap][cin][cout] for the direct kernel / wgrad checks
};
struct Conv32Args {
    const void* in = nullptr; int in_nb = 1;
    const ConvW* w = nullptr;
    const float* bias = nullptr;
    const void* up = nullptr; int up_nb = 1;
    const void* mask = nullptr; int mask_nb = 1;
    const void* add1 = nullptr; int add1_nb = 1;
    const void* add2 = nullptr; int add2_nb = 1;
    void* out_raw = nullptr; void* out_sum = nullptr;
    int B = 1, Hin = 0, Win = 0;    // input spatial size; output size follows from mode
    int mode = CONV_S1; int relu_in = 0; int bf16 = 0; int naive = 0;
    int x3 = 0;          // fp32 storage: bf16x3 arithmetic on the bf16 matrix cores (stride-1 only)
    int w2 = 0;          // narrow storage: the weight operand as hi + lo, two MFMAs per product (the mixed mode's data gradients)
    // sign-bit masks (ptta_common.h Epi): fp32 storage, non-naive kernels only -- the caller passes them only in that mode
    const uint32_t* mask_bits = nullptr;       // backward: replaces the reads of `mask` (which stays set for the other modes)
    uint32_t* bits_out = nullptr; int bits_nb = 0; int bits_sum = 0;      // forward: bits of out_sum (bits_sum) or out_raw, frames b < bits_nb
};
void ptta_pack_conv32(const float* src, const ConvW& w, int in_major, int flip, hipStream_t s, int row_stride = 32, int col_off = 0);
int ptta_launch_conv32(const Conv32Args& a, hipStream_t s);
// layer-loop trial (conv32.hip conv32_s1_small_loop_kernel; tools/bench_chain.py): `reps` dependent plain stride-1 layers in one launch
int ptta_launch_conv32_loop(const Conv32Args& a, void* buf_a, void* buf_b, int reps, unsigned* ctr, unsigned* base_io, int* err, int variant, hipStream_t s);

// ---- conv_small.hip ---------------------------------------------------------------------------
struct Plane { const float* p = nullptr; int nb = 1; long bstride = 0;
               // optional photometric normalisation applied while loading (transforms.py:668-710):
               // v -> (v / div - mean) / stdv, in-bounds samples only (the conv's zero padding stays zero)
               int norm = 0; float div = 1.f, mean = 0.f, stdv = 1.f; };
struct ConvInArgs {      // planar fp32 (cin = 1..3) -> 32-channel NHWC
    Plane pl[3]; int cin = 1; int zero_from_b = 1 << 30;
    const float* wfrag = nullptr;   // [14][64] fp32 MFMA fragments
    const float* wcanon = nullptr;  // [tap][cin][32]
    const float* bias = nullptr;
    const void* up = nullptr; int up_nb = 1;
    const void* mask = nullptr; int mask_nb = 1;
    const void* add1 = nullptr; int add1_nb = 1;
    void* out_raw = nullptr; void* out_sum = nullptr;
    int B = 1, H = 0, W = 0; int bf16 = 0; int naive = 0;
    // fused first-kernel forms only (ptta_launch_conv32_first): sign bits of `mask` / bits of the a_out map
    const uint32_t* mask_bits = nullptr; uint32_t* a_bits = nullptr;
    int a_bits_nb = 0;              // unfused launch (fp32 output): also write the sign-bit plane `a_bits` of out_raw for frames b < a_bits_nb
};
void ptta_pack_conv_in(const float* src, int cin_total, int cin_first, int cin, int transpose_flip,
                       float* wfrag, float* wcanon, hipStream_t s);
int ptta_launch_conv_in(const ConvInArgs& a, hipStream_t s);
// the first two layers of an encoder stage in one launch (conv32.hip conv32_s1_first_kernel); returns 1 when not applicable
int ptta_launch_conv32_first(const Conv32Args& a, const ConvInArgs& f, void* a_out, int a_nb, hipStream_t s);

struct ConvOut1Args {    // 32-channel NHWC -> planar fp32, 1 channel
    const void* in = nullptr; int in_nb = 1;
    const float* w = nullptr;       // [tap][32]
    const float* bias = nullptr;    // [1] or null
    const float* add = nullptr; int add_nb = 1;
    float* out = nullptr;
    int B = 1, H = 0, W = 0; int relu_in = 0; int bf16 = 0;
};
void ptta_pack_conv_out1(const float* src, int src_cin_total, int src_cin_index, int from_conv_in, float* w, hipStream_t s);
int ptta_launch_conv_out1(const ConvOut1Args& a, hipStream_t s);

// ---- resample.hip -----------------------------------------------------------------------------
int ptta_launch_prep(const float* sparse, float max_input_depth, float* dclamp, float* d12, float* d14,
                     int N, int H, int W, hipStream_t s);
int ptta_launch_up2_1ch(const float* in, float* out, int B, int Hin, int Win, hipStream_t s);
int ptta_launch_up2T_1ch(const float* gout, float* gin, int B, int Hin, int Win, hipStream_t s);
int ptta_launch_up2T_32(const void* gout, const void* add, void* gin, int B, int Hin, int Win, int bf16, hipStream_t s);

int ptta_launch_outlier_removal(const float* sparse, const float* validity, float* sparse_out, float* validity_out,
                                int N, int H, int W, int ksize, float threshold, float* scratch, hipStream_t s);

// ---- SyncBatchNorm across ranks (the reference converts every BatchNorm to SyncBatchNorm before its DDP run,
// src/tta_main.py:326): the partial statistics of a BatchNorm [pass][block][2][C] are collapsed to double sums in the caller's
// exchange buffer, summed over the ranks by the caller's collective (RCCL through torch.distributed), and written back as
// block 0 (+ a low-order correction in block 1) so that the unchanged finalize kernels see GLOBAL sums; the row count they
// divide by is multiplied by the world size (equal local batches, as DistributedSampler + drop_last give).
typedef int (*ptta_allreduce_cb)(void* user, double* dev_buf, long long count, void* stream);
int ptta_rccl_sum_f64(void* comm, double* buf, long long count, hipStream_t s);                                // rccl_sync.hip
struct PttaStatSync {
    ptta_allreduce_cb fn = nullptr; void* user = nullptr;     // caller's collective (torch.distributed: gloo or RCCL through Python) ...
    void* comm = nullptr;                                     // ... or the library's own RCCL communicator (ptta_set_stat_sync_rccl): enqueued here, capturable
    double* buf = nullptr; long cap = 0; int world = 1;
    bool on() const { return (fn || comm) && (world > 1 || comm); }       // a communicator of ONE rank still exchanges (plumbing test)
    int exchange(long long count, hipStream_t s) const {                // SUM over the ranks of buf[0..count), in place, ordered on s
        if (comm) return ptta_rccl_sum_f64(comm, buf, count, s);
        const int rc = fn(user, buf, count, (void*)s);
        return rc ? (rc < 0 ? rc : -rc) : 0;
    }
};
int ptta_stat_sync(const PttaStatSync* sy, float* part, int nblocks, int C, int npass, hipStream_t s);      // gbn.hip

// ---- heads.hip --------------------------------------------------------------------------------
struct GemmArgs {
    const void* A = nullptr; int a_bf16 = 0;   // [R][K] (fp32, or bf16 NHWC features when a_bf16)
    const float* A2 = nullptr;                 // second A tensor (h1) for the BN-backward prologue
    const float* W = nullptr;                  // [N][K] row-major
    const float* bias = nullptr;               // [N] or null
    float* C = nullptr;                        // [R][N]
    int R = 0, K = 0, N = 0;
    int pro = 0;                               // 0 none, 1 relu(a*scale[k]+shift[k]), 2 BN-backward
    const float* pscale = nullptr; const float* pshift = nullptr;    // [K]
    const float* pmean = nullptr; const float* pinv = nullptr;       // [K] (pro 2)
    const float* pc1 = nullptr; const float* pc2 = nullptr;          // [K] (pro 2)
    int epi = 0;                               // 0 none, 1 column sum/sumsq, 2 relu-mask + sum g, sum g*xhat
    const float* eH = nullptr;                 // [R][N] pre-BN activations (epi 2)
    const float* escale = nullptr; const float* eshift = nullptr; const float* emean = nullptr; const float* einv = nullptr;
    float* part = nullptr;                     // [row_blocks][2][N] partial column statistics
    int x3 = 0; const bf16_t* Whi = nullptr; const bf16_t* Wlo = nullptr;   // bf16x3 arithmetic: pre-split [N][K] weight
    const bf16_t* Wil = nullptr;               // the same, interleaved per 32-k slice (one 128-B line per row and slice)
    // ---- heads v2: the 512-wide hidden of proj = Linear(32,512) is never materialised (N = 512 kernel only) ----
    // pro 3: A[r][k] = relu(bn(X[r][:] . W0[k][:] + b0[k])) computed by the staging waves on the matrix cores (K = 512 hidden units)
    // epi 3: data gradient through proj.3 with the hidden RECOMPUTED for the ReLU mask / BatchNorm-backward sums, and the masked gradient
    //        (x gamma x invstd) contracted with W0 inside the block: P[col block][r][32] instead of a 55-MB [R][512] store
    const float* X = nullptr;                  // [R][32] fp32 input rows of the first Linear
    const void* W0frag = nullptr;              // pro 3: W0 in MFMA A-fragment order (ptta_pack_w0_frag)
    const float* b0 = nullptr;                 // [512] bias of the first Linear
    const bf16_t* W0hi = nullptr; const bf16_t* W0lo = nullptr;       // epi 3: [512][32] bf16 planes
    const bf16_t* W0thi = nullptr; const bf16_t* W0tlo = nullptr;     // epi 3: [32][512] bf16 planes (transposed)
    float* P = nullptr;                        // epi 3: [2][R][32]
    // pro 4 (with epi 3): A = the cosine term's gradient wrt `ref`, computed while staged from A = emb, Bref = ref, the loss forward's
    // per-row statistics and the gated weight (loss.hip: ptta_loss_ws_rows_off, WS_SCAL[0]) -- the [R][512] gradient tensor is never written
    const float* Bref = nullptr; const float* rowstats = nullptr; const float* coef = nullptr;
};
// W0 [512][32] fp32 -> [slice 16][kstep 2][hi, lo][lane 64] uint4 fragments (8 bf16: hidden unit 32*slice + (lane & 31), channels 16*kstep + 8*(lane >> 5) ...)
void ptta_pack_w0_frag(const float* W0, void* frag, hipStream_t s);
// Train-mode BatchNorm1d statistics of h = X W0^T + b0 WITHOUT computing h: per 256-row block the second moments of X (fp64), then per hidden
// unit sum h = n b + w . sum x and sum h^2 = w^T S w + 2 b w . sum x + n b^2 -> partials in the layout ptta_launch_bn_finalize reduces
// ([pass][block][2][512]); X holds `npass` consecutive groups of R rows.
int ptta_head_moment_blocks(long R);
int ptta_launch_head_moments(const float* X, long R, int npass, const float* W0, const float* b0, float* part, hipStream_t s);
// after the epi-3 GEMM + ptta_launch_bn_bwd_finalize: dX = P[0] + P[1] - X M - u (M, u from the BatchNorm-backward means, derived per block in fp64)
// halves: P holds 2 column-block halves (heads.hip EPI 3) or 1 (heads_n.hip); dX_bf16: the gradient map is narrow
int ptta_launch_head_bwd_finish(const double* k12, const float* W0, const float* P, const float* X, long R, void* dX, hipStream_t s, int halves = 2, int dX_bf16 = 0);
void ptta_split_weight(const float* w, bf16_t* hi, bf16_t* lo, bf16_t* il, long n, int K, hipStream_t s);
int ptta_gemm_row_blocks(int R);
int ptta_gemm_part_blocks(const GemmArgs& a);
int ptta_launch_gemm(const GemmArgs& a, hipStream_t s);
// BatchNorm1d (train) statistics from partials; also updates running stats (momentum 0.1, unbiased var)
int ptta_launch_bn_finalize(const float* part, int row_blocks, int R, int N, const float* gamma, const float* beta,
                            float eps, float momentum, float* running_mean, float* running_var, long long* nbt,
                            float* mean, float* invstd, float* scale, float* shift, hipStream_t s);
int ptta_launch_bn_bwd_finalize(const float* part, int row_blocks, int R, int N, const float* gamma, const float* invstd,
                                float* gscale, float* c1, float* c2, hipStream_t s, float* dgamma = nullptr, float* dbeta = nullptr,
                                double* k12 = nullptr, const float* b0 = nullptr, const float* mean = nullptr);    // k12: [k1 512 | k2 512] for ptta_launch_head_bwd_finish

// ---- heads_n.hip: the heads of the mixed mode (narrow storage, one bf16 MFMA per product) --------
struct HnGemmArgs {
    int pro = 1, epi = 0;          // pro 1 / 3 / 4 / 5, epi 0 / 1 / 3 (heads_n.hip header)
    const bf16_t* A = nullptr;     // pro 1: previous layer's output [R][512]; pro 4: emb
    const float* Af = nullptr;     // pro 5: fp32 gradient tensor [R][512]
    const bf16_t* Bref = nullptr; const float* rowstats = nullptr; const float* coef = nullptr;      // pro 4
    const void* X = nullptr; int x_bf16 = 0;        // pro 3 / epi 3: the 32-channel feature rows (fp32 real frames / narrow proxy frames)
    const bf16_t* W0 = nullptr; const float* b0 = nullptr;           // proj.0: [512][32] bf16, bias
    const float* pscale = nullptr; const float* pshift = nullptr;    // BatchNorm1d + ReLU applied to the A operand
    const bf16_t* W = nullptr; const float* bias = nullptr;          // [512][512] bf16 row-major ([out][in]), bias
    bf16_t* C = nullptr; float* part = nullptr;                      // narrow output (TILED, heads_n.hip); [row blocks of 128][2][512] column partials (epi 1 / 3)
    const bf16_t* E = nullptr; float* rs = nullptr;                  // epi 2: rs[r] = |c_r|^2 out; epi 5: E = the epi-2 tensor, rs = its row sums (in)
    float* rowstats_out = nullptr; float* cpart = nullptr; int cpart_n = 0;      // epi 5: (|e|, |c|, cos) per row; block partials of sum (2 - 2 cos), array length
    const float *escale = nullptr, *eshift = nullptr, *emean = nullptr, *einv = nullptr;             // epi 3: proj.1's state
    const bf16_t* W0t = nullptr; float* P = nullptr;                 // epi 3: [32][512] bf16; P [R][32] fp32
    long R = 0;
};
int ptta_hn_row_blocks(long R);
long ptta_hn_tiled_elems(long R);                                   // elements of a tiled [R][512] tensor (padded to whole 128-row blocks)
void ptta_hn_pack_w(const bf16_t* w_hi_rowmajor, bf16_t* w_slice_major, int K, hipStream_t s);
int ptta_launch_hn_untile(const void* src_tiled, float* dst, long R, hipStream_t s);
int ptta_hn_moment_blocks(long R);
long ptta_hn_moment_scratch(long R);       // doubles of scratch per pass
struct HnBnOut {                 // what bn_finalize_kernel (heads.hip) writes: with `mean` set the stats kernel finalises the BatchNorm itself
    const float *gamma = nullptr, *beta = nullptr; float eps = 1e-5f, momentum = 0.1f;
    float *rm = nullptr, *rv = nullptr; long long* nbt = nullptr;
    float *mean = nullptr, *inv = nullptr, *scale = nullptr, *shift = nullptr;
};
int ptta_launch_hn_moments(const void* X, int x_bf16, long R, int npass, const float* W0, const float* b0, double* scratch, float* part, const HnBnOut* bn,
                           hipStream_t s);
long ptta_loss_ws_cos_off(int N, long R); int ptta_loss_cos_blocks(long R);        // loss.hip: where the cosine term's block partials live in the workspace
int ptta_launch_hn_gemm(const HnGemmArgs& a, hipStream_t s);

// ---- loss.hip ---------------------------------------------------------------------------------
struct LossScalars;      // device-resident scalars, see loss.hip
int ptta_loss_ws_floats(int N, int H, int W, long R);
long ptta_loss_ws_rows_off(int N);        // floats from the workspace start to the per-row statistics [R][3] = (|e|, |ref|, cos); [0] = the gated cosine coefficient
int ptta_launch_loss_forward(const float* depth, const float* image, const float* sparse, const float* validity,
                             float max_input_depth, const float* emb, const float* ref, long R, int D,
                             const float* w3_dev /* w_sd, w_sm, w_cos */, int N, int H, int W,
                             float* ws, float* loss_info, hipStream_t s, int defer_finalize = 0);
int ptta_launch_loss_depth_part(const float* depth, const float* image, const float* sparse, const float* validity, float max_input_depth,
                                int N, int H, int W, float* ws, hipStream_t s);
int ptta_launch_loss_cos_part(const void* emb, const void* ref, long R, int D, int N, float* ws, hipStream_t s, int narrow = 0);      // narrow: bf16 [R][512]
int ptta_launch_loss_finalize(float* ws, int N, int H, int W, long R, int has_cos, const float* w3_dev, float* loss_info, hipStream_t s);
// the gated cosine coefficient alone (workspace [0]): all the heads' backward needs of the finalisation -- it depends on the cosine partials only
int ptta_launch_loss_cos_coef(float* ws, int N, long R, const float* w3_dev, hipStream_t s);
int ptta_launch_loss_backward(const float* depth, const float* image, const float* sparse, const float* validity,
                              float max_input_depth, const float* emb, const float* ref, long R, int D,
                              int N, int H, int W, float* ws, float* gdepth, float* gref, hipStream_t s, const float* w3_fused = nullptr,
                              float* loss_info_fused = nullptr, int cos_partials_ready = 0, int valid_count_ready = 0);
// the valid-weight partials of the sparse-depth term: all the depth gradient's coefficients depend on (a function of the step's inputs alone);
// with valid_count_ready the depth gradient reads them instead of the loss partials and reports nothing (fused step: ptta_api.hip step_tail)
int ptta_launch_loss_valid_count(const float* sparse, const float* validity, int N, int H, int W, float* ws, hipStream_t s);
// (cos_partials_ready with emb == NULL: the depth gradient alone, its in-kernel finalisation includes the cosine term and the gate)
// validity may be NULL in both calls: where(sparse > 0, 1, sparse) is then computed on the fly (src/tta_main.py:583-586)

int ptta_launch_eval_metrics(const float* depth, const float* gt, long n, float min_eval, float max_eval, double* scratch, float* out4,
                             hipStream_t s);

// ---- wgrad_adam.hip ---------------------------------------------------------------------------
int ptta_wgrad_chunks(long pixels);
int ptta_launch_wgrad32(const void* x, const void* gy, int bf16, int B, int H, int W, float* part,
                        float* gw, float* gb, hipStream_t s, int co_stride = 288);

// ---- meta2.hip (2layers meta layer: BatchNorm2d / LeakyReLU pieces) ----------------------------
int ptta_chan_stats_blocks();
int ptta_launch_chan_stats32(const void* x, const void* g, int bf16, long pix0, long npix, const float* fscale,
                             const float* fshift, const float* mean, const float* inv, float slope, float* part, hipStream_t s);
int ptta_launch_bn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale, float* shift, int C, hipStream_t s);
int ptta_launch_bn_apply32(const void* x, const void* res, void* y, int bf16, long npix, long pix_per_pass, const float* scale,
                           const float* shift, int stat_stride, float slope, hipStream_t s);
int ptta_launch_bn_bwd_apply32(const void* x, const void* g, void* dx, int bf16, long npix, const float* fscale, const float* fshift,
                               const float* mean, const float* inv, const float* gscale, const float* c1, const float* c2,
                               float slope, hipStream_t s);
int ptta_launch_bn2d_bwd_finalize(const float* part, int nblocks, long R, const float* gamma, const float* inv, float* dgamma,
                                  float* dbeta, float* gscale, float* c1, float* c2, hipStream_t s);
int ptta_launch_adam(float* p, float* m, float* v, const float* g, long n, const float* hyper /*lr,b1,b2,eps,wd*/,
                     const int* step_dev, hipStream_t s);
int ptta_launch_step_inc(int* step_dev, hipStream_t s);
// every adapted tensor in one launch; also increments the step count (torch.optim.Adam's state['step'])
// rep: how many times the optimizer's parameter list names this tensor (torch.optim.Adam updates a parameter once per occurrence, each
// with its own step count: occurrence r of step t uses (t - 1) * rep + r + 1); 0 / 1 = the usual single update
struct PttaAdamEntry { float *p, *m, *v; const float* g; long n, off; int rep = 1; };
// torch.optim.Adam's update of ONE element, the same instructions wherever it runs (adam_kernel, adam_multi_kernel, the weight gradient's
// reduction with Adam inside): contraction is off in this body and the two fused multiply-adds are spelled out -- left to hipcc, `b1 m + (1 - b1) g`
// becomes fma(b1, m, (1 - b1) g) in one kernel and fma(1 - b1, g, b1 m) in another, and the "same" update differs in the last bit.
__device__ __forceinline__ void ptta_adam_update(float& pk, float& mk, float& vk, float g0, float wd, float b1, float b2, float eps,
                                                 float step_size, float bc2s) {
#pragma clang fp contract(off)
    float gg = g0;
    if (wd != 0.f) gg = __builtin_fmaf(wd, pk, gg);
    mk = __builtin_fmaf(b1, mk, (1.f - b1) * gg);
    vk = __builtin_fmaf(b2, vk, ((1.f - b2) * gg) * gg);
    pk = pk - step_size * (mk / (sqrtf(vk) / bc2s + eps));
}
int ptta_launch_adam_multi(const PttaAdamEntry* tab_dev, int nt, long total, const float* hyper, int* step_dev, unsigned* ticket_dev, hipStream_t s);
int ptta_launch_set_floats(float* dst, const float* host_src, int n /*<= 8*/, hipStream_t s);   // by kernel argument: no sync
int ptta_launch_set_int(int* dst, int v, hipStream_t s);

// ---- head_train.hip (stage-2 head trainer: Linear weight gradients, prepare loss, EMA) -----------
struct LinWgradArgs {
    const float* G = nullptr;      // [R][O] upstream gradient (pre-transform)
    const float* Gh = nullptr;     // [R][O] pre-BatchNorm hidden: when set, G is BN-backward transformed on the fly
    const float *gscale = nullptr, *gc1 = nullptr, *gc2 = nullptr, *gmean = nullptr, *ginv = nullptr;
    const float* X = nullptr;      // [R][I] layer input
    const float *xscale = nullptr, *xshift = nullptr;   // when set, X = relu(X * xscale + xshift) on the fly
    float* Wpart = nullptr;        // [chunks][O][I]
    float* bpart = nullptr;        // [chunks][O]
    long R = 0; int O = 512, I = 512, rchunk = 0;
};
int ptta_linear_wgrad_chunks(long R, int* rchunk_out);
int ptta_launch_linear_wgrad(LinWgradArgs a, float* dW /*[O][I]*/, float* db /*[O] or NULL*/, hipStream_t s);
int ptta_launch_prepare_loss(const float* emb, const float* ref, long R, int C, float* g_emb, float* part /* >= 1024 floats */, float* loss,
                             hipStream_t s);
int ptta_launch_ema_multi(const PttaAdamEntry* tab_dev, int nt, long total, const float* tau_dev /* {tau, 1 - tau} */, hipStream_t s);

// ---- dcn.hip (modulated deformable convolution, NCHW fp32) -------------------------------------
struct DcnArgs {
    const float *in = nullptr, *weight = nullptr, *bias = nullptr, *offset = nullptr, *mask = nullptr, *gout = nullptr;
    float *out = nullptr, *gin = nullptr, *goff = nullptr, *gmask = nullptr, *gweight = nullptr, *gbias = nullptr;
    int B = 1, C = 1, H = 0, W = 0, Co = 1, kh = 3, kw = 3, sh = 1, sw = 1, ph = 1, pw = 1, dh = 1, dw = 1, group = 1, dg = 1;
};
int ptta_launch_dcn_forward(const DcnArgs& a, hipStream_t s);
int ptta_launch_dcn_backward(const DcnArgs& a, hipStream_t s);

// ---- generic NHWC layers of the NLSPN backbone: gconv.hip / gbn.hip / nlspn_prop.hip -----------------------------
enum { GACT_NONE = 0, GACT_RELU = 1, GACT_LRELU = 2, GACT_SIGMOID = 3, GACT_ELU = 4 };
struct GView {           // strided NHWC view: element (b,y,x,c) at p[((b*H + y)*W + x)*ld + c]; p already includes the channel offset
    float* p = nullptr; int B = 0, H = 0, W = 0, C = 0, ld = 0;
};
struct GConvArgs {
    GView x, y;
    const float* w = nullptr;       // packed [tap][cin][cout] (ptta_gpack)
    const float* bias = nullptr;    // [cout] or null
    int k = 3, stride = 1, transposed = 0, act = GACT_NONE, accumulate = 0;
    long wld = 0, wts = 0;          // weight row stride (columns of the packed matrix) and tap stride; 0 = dense (cout, cin*cout)
};
void ptta_gpack(const float* src, float* dst, int KK, int A, int B, long a_stride, long b_stride, int flip, hipStream_t s);
int ptta_launch_gconv_direct(const GConvArgs& a, hipStream_t s);
int ptta_launch_gact_bwd(const GView& g, const GView& y, int act, hipStream_t s);
int ptta_launch_gnchw_to_nhwc(const float* src, int src_nb, int src_c, const GView& y, int zero_from_b, int norm, float div, const float* mean,
                              const float* stdv, hipStream_t s);
int ptta_gwgrad_slabs(long pixels);
int ptta_launch_gwgrad(const GView& x, const GView& gy, float* part, float* gw, float* gb, hipStream_t s);
// weight / bias gradient of a 1x1 convolution = nn.Linear over the rows (ghead.hip): gw [Co][Ci], gb [Co] or NULL
int ptta_launch_glinear_wgrad(const GView& x, const GView& gy, float* gw, float* gb, hipStream_t s);

int ptta_gbn_part_floats(int C, int npass);
int ptta_launch_gbn_forward(const GView& x, const GView& res, const GView& y, int npass, int act, float eps, const float* gamma,
                            const float* beta, float* part, float* st, hipStream_t s, int fused_blocks = 0, int act_first = 0,
                            const PttaStatSync* sync = nullptr,
                            // tracked BatchNorm: running statistics updated (and the eval-mode affine written) by the finalize launch
                            float* rm = nullptr, float* rv = nullptr, long long* nbt = nullptr, float momentum = 0.1f, int repeats = 1,
                            float* st_eval = nullptr);
int ptta_gconv_x3_tiles(int B, int H, int W);
int ptta_launch_gbn_apply(const GView& x, const GView& res, const GView& y, int npass, int act, const float* st, int res_relu, hipStream_t s,
                          int act_first = 0);
int ptta_launch_gbn_backward(const GView& x, const GView& g, const GView& y, const GView& gx, const GView& gres, int npass, int act,
                             int res_relu, int acc_gx, int acc_gres, const float* gamma, const float* st, float* part, float* bw,
                             float* dgamma, float* dbeta, hipStream_t s, int act_first = 0, const PttaStatSync* sync = nullptr);

int ptta_launch_nl_affinity_fwd(const GView& oa, const float* conf, const float* S, int legacy, float* off9, float* aff9, hipStream_t s);
int ptta_launch_nl_pin(const float* feat, const float* fix, float* out, long n, hipStream_t s);       // out = fix > 0 ? fix : feat
// pinned: the propagated map with the sparse input already imposed; out = pin(raw sweep output) when pin_out, else the raw output
int ptta_launch_nl_prop_fwd(const float* pinned, const float* fix, const float* off9, const float* aff9, float* out, int pin_out, int B, int H,
                            int W, hipStream_t s);
int ptta_launch_nl_prop_bwd(const float* pinned, const float* fix, const float* off9, const float* aff9, const float* gout, float* gfeat,
                            float* goff9, float* gaff9, int B, int H, int W, hipStream_t s);
int ptta_launch_nl_affinity_bwd(const GView& oa, const float* conf, const float* S, int legacy, const float* goff9, const float* gaff9,
                                const GView& goa, float* gconf, hipStream_t s);

// ---- gconv_mfma.hip: matrix-core (bf16x3) stride-1 convolution over 1-2 NHWC sources -------------------------------
struct GX3Args {
    const float* x0 = nullptr; int C0 = 0, ld0 = 0;
    const float* x1 = nullptr; int C1 = 0, ld1 = 0;
    int B = 0, H = 0, W = 0;
    const uint4* whi = nullptr; const uint4* wlo = nullptr;   // fragments [co tile][chunk][tap][2][64 lanes] x 8 bf16
    int nchunks = 0, nf0 = 0, nnf = 0;                        // first output-channel tile of this launch, number of tiles
    float* y = nullptr; int ldy = 0, Cy = 0;                  // output view starting at channel 32*nf0 of the packed matrix
    const float* bias = nullptr; int act = GACT_NONE, accumulate = 0;
    // optional fused BatchNorm statistics of the OUTPUT (stride-1 kernel): per (pass, pixel tile) partial sums
    // {sum y, sum y^2} per channel, layout [pass][tile][2][stat_C]; stat_npass equal slices of the batch
    float* stat_part = nullptr; int stat_C = 0, stat_npass = 1;
    int vert = 0;                                             // stride-1 3x3 only: fragments hold the three vertical taps (3x1x1 Conv3d)
    // bf16x6: the first six_B images of the batch (the real frames of a [real | proxy] launch) are computed with a three-way operand
    // split x = h + m + l, w = h + m + l and the six products down to 2^-16 (hh, hm, mh, mm, hl, lh): fp32-grade products where the
    // two-way split's 2^-17 operand error is too much (CostDCNet: gradient signs after the first Adam step); wl2 = the weights' l plane
    const uint4* wl2 = nullptr; int six_B = 0;
    // mixed mode of the generic engine: images b >= x1_from_B take ONE bf16 MFMA per product (hi x hi) instead of three -- the proxy frames
    // of a [real | proxy] launch (the reference's no_grad pass) and every data-gradient launch (x1_from_B = 0)
    int x1_from_B = 1 << 30;
    // ... with the WEIGHT operand kept as hi + lo (two MFMAs: a_hi w_lo, a_hi w_hi): the data gradients.  A bf16-rounded weight is a systematic
    // error of the gradient's direction (conv32.hip mma_step W2), a rounded gradient value is noise
    int x1_w2 = 0;
};
void ptta_gfrag_pack(const float* canon, long wld, long wts, int KK, int C0, int C1, int c0_0, int c0_1, int Co, bf16_t* hi, bf16_t* lo,
                     hipStream_t s, bf16_t* l2 = nullptr);
long ptta_gfrag_elems(int KK, int C0, int C1, int Co);
int ptta_launch_gconv_x3(const GX3Args& a, int ks, hipStream_t s);
int ptta_launch_gconv_x3_strided(const GX3Args& a, int ks, int mode, int hin, int win, hipStream_t s);
long ptta_gwgrad_mfma_part_floats(long pixels, int Ci, int Co);
// Adam applied by the weight gradient's reduction itself (gconv_mfma.hip gwgrad_mfma_reduce_kernel; <= 32 x 32 channels only): parameter + both
// moments of the weight and of the bias, the hyper-parameter block, the device step count and its ticket (as ptta_launch_adam_multi)
struct GwAdam { float *pw = nullptr, *mw = nullptr, *vw = nullptr, *pb = nullptr, *mb = nullptr, *vb = nullptr; const float* hyper = nullptr; int* step = nullptr; unsigned* ticket = nullptr; };
int ptta_launch_gwgrad_mfma(const GView& x, const GView& gy, float* part, float* gw, float* gb, hipStream_t s, int gy_bf16 = 0, const GwAdam* adam = nullptr);
