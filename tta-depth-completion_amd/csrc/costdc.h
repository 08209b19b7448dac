// Launch interface of costdc_kernels.hip (internal; CostDCNet backbone).
#pragma once
#include <hip/hip_runtime.h>
#include "ptta_kernels.h"

// sparse voxel sets of the 3-D encoder: three levels with tensor stride (1, 1<<l, 1<<l); counts live on the device
struct CdSparse {
    int N = 0, H = 0, W = 0;
    int* vol[3] = {nullptr, nullptr, nullptr};        // dense index volumes [N][16][H>>l][W>>l], -1 = empty
    int4* coords[3] = {nullptr, nullptr, nullptr};    // (frame, plane, y, x) per voxel
    int* cnt = nullptr;                               // [3] voxel counts
    int *rowcnt = nullptr, *rowoff = nullptr;         // scan scratch (N * 16 * H entries)
    float* feat_in = nullptr;                         // level-0 input feature (residual to the plane), [cap][1]
    float *bn_part = nullptr, *bn_st = nullptr;
};

int cd_launch_pad_dual(const float* src, float* dst, int N, int C, int H, int W, int Hp, int Wp, int pt, int pr, hipStream_t s, int norm = 0, float div = 1.f,
                       const float* mean = nullptr, const float* stdv = nullptr);
int cd_launch_crop_avg(const float* net, float* out, int N, int H, int W, int Hp, int Wp, int pt, int pr, hipStream_t s);
int cd_launch_scatter_dual_grad(const float* g, float* gnet, int N, int H, int W, int Hp, int Wp, int pt, int pr, hipStream_t s);
int cd_launch_stage(const float* image, const float* sparse, float* out, int N, int passes, int H, int W, int C, int norm, float div, const float* mean,
                    const float* stdv, hipStream_t s);
int cd_launch_clamp(const float* src, float* dst, long n, float maxd, hipStream_t s);
int cd_sparse_levels_build(const CdSparse& q, const float* sparse, float z_step, hipStream_t s);
int cd_launch_sparse_conv(const CdSparse& q, const float* fin, int lin, int lout, const float* Wk, int ksize, int Ci, int Co, float* fout, hipStream_t s);
int cd_launch_sparse_bn(const CdSparse& q, const float* f, const float* res, int level, int C, const float* gamma, const float* beta, float* rm, float* rv,
                        long long* nbt, int train, int repeats, int relu, float* out, hipStream_t s, const PttaStatSync* sync = nullptr, float* st = nullptr);
int cd_launch_sparse_conv_bwd(const CdSparse& q, const float* gy, int lin, int lout, const float* Wk, int ksize, int Ci, int Co, float* gx, int acc, hipStream_t s);
int cd_launch_sparse_bn_bwd(const CdSparse& q, const float* f, const float* y, const float* g, int level, int C, const float* gamma, const float* st, int relu,
                            float* dgamma, float* dbeta, float* gx, float* gres, int acc_res, float* bw, hipStream_t s, const PttaStatSync* sync = nullptr);
int cd_launch_densify_bwd(const CdSparse& q, const float* gvol, int ldv, int c0, int C, float* g, hipStream_t s);
int cd_launch_densify(const CdSparse& q, const float* f, int C, float* dense, hipStream_t s);
int cd_launch_fusion_fwd(const float* feat2d, const float* feat3d, float* vol, float* maskw, int N, int passes, int h, int w, hipStream_t s);
int cd_launch_fusion_bwd(const float* gvol, const float* maskw, float* gfeat2d, int N, int h, int w, hipStream_t s);
int cd_launch_pool_fwd(const float* x, float* y, long items_out, int H, int W, int C, hipStream_t s);
int cd_launch_pool_bwd(const float* x, const float* gy, float* gx, long items_in, int H, int W, int C, int acc, hipStream_t s);
int cd_launch_up_fwd(const float* x, float* y, long frames, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, hipStream_t s);
int cd_launch_up_bwd(const float* gy, float* gx, long frames, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C, int acc, hipStream_t s);
int cd_launch_regress_fwd(const float* cost, float* pred, int N, int h, int w, float z_step, hipStream_t s);
int cd_launch_regress_bwd(const float* cost, const float* gpred, float* gcost, int N, int h, int w, float z_step, hipStream_t s);
int cd_launch_rows_fwd(const float* feat, float* rows, int N, int D, int h, int w, int C, hipStream_t s);
int cd_launch_rows_bwd(const float* grows, float* gfeat, int N, int D, int h, int w, int C, int acc, hipStream_t s);
