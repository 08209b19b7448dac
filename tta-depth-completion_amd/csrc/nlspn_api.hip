// NLSPN engine of libptta_hip (BASELINE config 3): the ProxyTTA step of
// NLSPNModel_Adapt._rgbd_meta_contrast (external_src/NLSPN/src/model/nlspnmodel_adapt.py:850-944) with
// adapt_parameters('meta_bn') (src/nlspn_model_adapt.py:322-337): conv1_rgb_meta + every BatchNorm2d gamma/beta
// adapt (88 tensors / 40,048 values), every BatchNorm2d normalises with batch statistics in train and eval mode.
//
// The network is a flat op list built once per handle ("program"): CONV (one or two channel-concatenated
// sources, so torch.cat never materialises), BN (+activation, + residual add of a BasicBlock), then the fused
// propagation kernels of nlspn_prop.hip.  The grad pass and the no-grad proxy pass (zero image) of a training
// step are batched as [real | proxy] through the encoder with separate batch statistics per pass; decoder,
// propagation and the backward sweep touch the real half only.  The backward sweep is the op list in reverse:
// data gradients everywhere (each is again a CONV with re-packed weights), BatchNorm gradients for gamma/beta,
// one weight gradient (conv1_rgb_meta), then Adam on device for all adapted tensors.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "nlspn.h"
#include "ptta_common.h"
#include "ptta_kernels.h"

int ptta_launch_adam(float* p, float* m, float* v, const float* g, long n, const float* hyper, const int* step_dev, hipStream_t s);
int ptta_launch_step_inc(int* step_dev, hipStream_t s);

#define NCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_), -100 - (int)e_); } while (0)
#define NRUN(x) do { int r_ = (x); if (r_ != 0) return fail(std::string(#x) + " failed", r_ < 0 ? r_ : -r_); } while (0)

namespace {

enum { W_BOTH = 0, W_GRAD = 1, W_PROXY = 2 };
enum { K_CONV = 0, K_BN = 1 };
const float BN_EPS = 1e-5f;
const int PROP_TIME = 18;

struct Tn {
    std::string name;
    float *p = nullptr, *g = nullptr;
    int items = 0, H = 0, W = 0, C = 0, ld = 0;
    bool need_grad = false;
};

struct GConvW {            // frozen convolution weights, packed once at load time
    float *wf = nullptr, *wb = nullptr, *bias = nullptr;
    int Ci = 0, Co = 0, k = 3, stride = 1, transposed = 0;
    int C0 = 0, C1 = 0;                                   // source split of the input channels (torch.cat order)
    int Co_pad = 0;
    int Ci_real = 0;                                      // > 0: the state_dict tensor has fewer input channels than the zero-padded activation
    float* gpad = nullptr;                                // zero-padded copy of the output gradient (Co_pad channels)
    bf16_t *ff_hi = nullptr, *ff_lo = nullptr, *fb_hi = nullptr, *fb_lo = nullptr;     // bf16x3 MFMA fragments (forward / data gradient)
    bool mf = false, mb = false;                          // matrix-core kernel usable for forward / data gradient
    bool loaded = false, has_bias = false;
};

struct Op {
    int kind = K_CONV;
    bool train_only = false, bwd = true;
    // conv
    int nsrc = 1, x[2] = {-1, -1}, c0[2] = {0, 0}, xw[2] = {W_BOTH, W_BOTH};
    int y = -1, yw = W_BOTH;
    int k = 3, stride = 1, transposed = 0, act = GACT_NONE;
    std::string wname;
    int ad_w = -1, ad_b = -1;           // adapted (bound) weight / bias: conv1_rgb_meta
    bool first_x[2] = {true, true};
    // bn
    int res = -1;
    std::string bname;
    int ad_g = -1, ad_beta = -1;
    float* st = nullptr;
    float* part = nullptr;            // this BatchNorm's partial-statistics buffer
    int fused_from = -1;              // index of the producing CONV op whose epilogue fills `part` (stride-1 matrix-core kernel)
    int stat_to = -1;                 // (CONV) index of the BN op that consumes this conv's fused statistics
    bool first_raw = true, first_res = true;
};

struct Adapted { std::string name; long n = 0, goff = 0; float *p = nullptr, *m = nullptr, *v = nullptr; };

__global__ void clamp_dup_kernel(const float* __restrict__ src, float* __restrict__ dst, long per, int copies, float maxd) {
    const long total = per * copies;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = src[i % per];
        if (maxd >= 0.f) v = fminf(fmaxf(v, 0.f), maxd);
        dst[i] = v;
    }
}
__global__ void relu_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = fmaxf(src[i], 0.f);
}
// gradient of clamp(min=0): passes where y >= 0
__global__ void clamp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = y[i] >= 0.f ? g[i] : 0.f;
}
__global__ void validity_kernel(const float* __restrict__ sp, float* __restrict__ v, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float s = sp[i]; v[i] = s > 0.f ? 1.f : s; }
}
__global__ void pad_channels_kernel(const float* __restrict__ src, int lds_, int C, float* __restrict__ dst, int Cp, long npix) {
    const long total = npix * Cp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp); const long p = i / Cp;
        dst[i] = c < C ? src[p * lds_ + c] : 0.f;
    }
}
inline int nb(long total) { long b = (total + 255) / 256; if (b > 16384) b = 16384; if (b < 1) b = 1; return (int)b; }

}  // namespace

struct nlspn_engine {
    int N = 1, H = 0, W = 0;
    ptta_hparams hp{};
    std::string err;
    std::vector<void*> allocs;
    bool oom = false;
    std::vector<Tn> T;
    std::map<std::string, int> tid;
    std::vector<Op> ops;
    std::map<std::string, GConvW> convs;
    std::map<std::string, std::pair<float*, float*>> frozen_bn;     // heads' BatchNorm1d gamma/beta (not adapted)
    std::vector<Adapted> adapted;
    std::map<std::string, int> aid;
    float* gall = nullptr; long gall_n = 0;
    float *meta_wf = nullptr, *meta_wb = nullptr;
    float* w3_tmp = nullptr;
    float *hyper = nullptr, *loss_ws = nullptr, *loss_info = nullptr, *validity_tmp = nullptr, *S = nullptr;
    int* step_dev = nullptr;
    float *bn_part = nullptr, *bn_bw = nullptr, *wg_part = nullptr;
    // inputs / propagation
    int t_sd16 = -1;
    int t_img = -1, t_sd = -1, t_pred = -1, t_oa = -1, t_conf = -1, t_fe6 = -1, t_emb = -1, t_ref = -1;
    float *off9 = nullptr, *aff9 = nullptr, *goff9 = nullptr, *gaff9 = nullptr, *feats = nullptr, *depth = nullptr, *gdepth = nullptr,
          *gy = nullptr, *gping = nullptr;
    int legacy = 0, naive = 0;
    bf16_t *meta_ff_hi = nullptr, *meta_ff_lo = nullptr, *meta_fb_hi = nullptr, *meta_fb_lo = nullptr;
    int norm_on = 0; float norm_div = 1.f, norm_mean[3] = {0, 0, 0}, norm_std[3] = {1, 1, 1};
    bool fwd_valid = false;

    int fail(const std::string& m, int code) { err = m; return code; }
    void* dalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { oom = true; return nullptr; }
        allocs.push_back(p);
        if (hipMemset(p, 0, bytes ? bytes : 16) != hipSuccess) { oom = true; return nullptr; }
        return p;
    }
    float* falloc(size_t n) { return (float*)dalloc(n * sizeof(float)); }

    int tensor(const std::string& name, int items, int h, int w, int c, bool need_grad) {
        Tn t; t.name = name; t.items = items; t.H = h; t.W = w; t.C = c; t.ld = c; t.need_grad = need_grad;
        t.p = falloc((size_t)items * h * w * c);
        if (need_grad) t.g = falloc((size_t)N * h * w * c);
        T.push_back(t); tid[name] = (int)T.size() - 1;
        return (int)T.size() - 1;
    }
    int slice(const std::string& name, int parent, int c_off, int c) {
        Tn t = T[parent]; t.name = name; t.C = c; t.p = T[parent].p + c_off; t.g = T[parent].g ? T[parent].g + c_off : nullptr;
        T.push_back(t); tid[name] = (int)T.size() - 1;
        return (int)T.size() - 1;
    }
    GView view(int t, int which, bool train, bool grad = false) const {
        const Tn& tn = T[t];
        GView v; v.H = tn.H; v.W = tn.W; v.C = tn.C; v.ld = tn.ld;
        v.B = (which == W_BOTH && train) ? 2 * N : N;
        v.p = (grad ? tn.g : tn.p);
        if (which == W_PROXY && !grad) v.p += (size_t)N * tn.H * tn.W * tn.ld;
        return v;
    }

    // ---- program construction --------------------------------------------------------------------------------------
    int add_adapted(const std::string& name, long n) {
        Adapted a; a.name = name; a.n = n; a.goff = gall_n; gall_n += n;
        adapted.push_back(a); aid[name] = (int)adapted.size() - 1;
        return (int)adapted.size() - 1;
    }
    void conv(const std::string& wname, int x0, int x1, int y, int k, int stride, int transposed, int act, int xw, int yw,
              bool train_only = false, bool bwd = true) {
        Op o; o.kind = K_CONV; o.wname = wname; o.x[0] = x0; o.x[1] = x1; o.nsrc = x1 >= 0 ? 2 : 1; o.c0[0] = 0; o.c0[1] = x1 >= 0 ? T[x0].C : 0;
        o.y = y; o.k = k; o.stride = stride; o.transposed = transposed; o.act = act; o.xw[0] = o.xw[1] = xw; o.yw = yw;
        o.train_only = train_only; o.bwd = bwd;
        GConvW& cw = convs[wname];
        cw.Ci = T[x0].C + (x1 >= 0 ? T[x1].C : 0); cw.Co = T[y].C; cw.k = k; cw.stride = stride; cw.transposed = transposed;
        cw.C0 = T[x0].C; cw.C1 = x1 >= 0 ? T[x1].C : 0;
        cw.mf = !naive && (stride == 1 && !transposed ? (cw.C0 % 8) == 0 && (cw.C1 % 8) == 0 : (cw.C0 % 16) == 0 && (cw.C1 % 16) == 0);
        cw.mb = !naive && ((cw.Co % 16) == 0 || (stride == 1 && !transposed)) && (cw.C1 == 0 || (cw.C0 % 32) == 0);
        cw.Co_pad = (cw.Co + 15) / 16 * 16;               // data gradient of a conv with < 16 output channels: gy is zero-padded
        ops.push_back(o);
    }
    // y = act(bn(x)) [+ res, relu]; adapted gamma/beta unless frozen (heads)
    void bn(const std::string& bname, int x, int y, int res, int act, int w, bool frozen = false, bool train_only = false, bool bwd = true) {
        Op o; o.kind = K_BN; o.bname = bname; o.x[0] = x; o.y = y; o.res = res; o.act = act; o.xw[0] = w; o.yw = w;
        o.train_only = train_only; o.bwd = bwd;
        const int C = T[x].C;
        if (frozen) frozen_bn[bname] = std::make_pair(falloc(C), falloc(C));
        else { o.ad_g = add_adapted(bname + ".weight", C); o.ad_beta = add_adapted(bname + ".bias", C); }
        o.st = falloc((size_t)4 * 2 * C);
        // statistics fused into the producing convolution's epilogue when that is the stride-1 matrix-core kernel
        for (int k = (int)ops.size() - 1; k >= 0; --k) {
            Op& pr = ops[k];
            if (pr.kind != K_CONV || pr.y != x) continue;
            const GConvW& cw = convs[pr.wname];
            if (cw.mf && pr.stride == 1 && !pr.transposed && pr.act == GACT_NONE && pr.yw == w && T[x].ld == T[x].C) {
                o.fused_from = k; pr.stat_to = (int)ops.size();
                o.part = falloc((size_t)2 * ptta_gconv_x3_tiles(2 * N, T[x].H, T[x].W) * C);
            }
            break;
        }
        ops.push_back(o);
    }

    void build() {
        const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4, H8 = H / 8, W8 = W / 8, H16 = H / 16, W16 = W / 16;
        const int N2 = 2 * N;
        // the adapted meta conv comes first in the reference's parameter list (src/nlspn_model_adapt.py:324-328)
        const int ad_mw = add_adapted("conv1_rgb_meta.weight", 48L * 48 * 9), ad_mb = add_adapted("conv1_rgb_meta.bias", 48);
        // the 3-channel image and the 1-channel sparse depth are staged zero-padded to 16 channels so that their first
        // convolutions run on the matrix-core kernel too (weight rows of the padding channels are zero)
        t_img = tensor("image", N2, H, W, naive ? 3 : 16, false);
        t_sd = tensor("sparse", N2, H, W, 1, false);
        const int sd16 = naive ? t_sd : tensor("sparse16", N2, H, W, 16, false);
        const int rgb1 = tensor("rgb1", N2, H, W, 48, false);
        const int fe1 = tensor("fe1", N2, H, W, 64, true);
        const int fe1_rgb = slice("fe1_rgb", fe1, 0, 48), fe1_dep = slice("fe1_dep", fe1, 48, 16);
        conv("conv1_rgb.0", t_img, -1, rgb1, 3, 1, 0, GACT_LRELU, W_BOTH, W_BOTH, false, false);
        conv("conv1_rgb_meta", rgb1, -1, fe1_rgb, 3, 1, 0, GACT_NONE, W_BOTH, W_BOTH);
        ops.back().ad_w = ad_mw; ops.back().ad_b = ad_mb;
        conv("conv1_dep.0", sd16, -1, fe1_dep, 3, 1, 0, GACT_LRELU, W_BOTH, W_BOTH, false, false);
        if (!naive) { convs["conv1_rgb.0"].Ci_real = 3; convs["conv1_dep.0"].Ci_real = 1; t_sd16 = sd16; }
        // ResNet34 stages (nlspnmodel_adapt.py:400-406; BasicBlock :70-116)
        const int planes[4] = {64, 128, 256, 512}, nblocks[4] = {3, 4, 6, 3};
        int hh = H, ww = W, cur = fe1, fe[7]; fe[1] = fe1;
        for (int st = 0; st < 4; ++st) {
            for (int b = 0; b < nblocks[st]; ++b) {
                const int stride = (b == 0 && st > 0) ? 2 : 1;
                const int ho = hh / stride, wo = ww / stride, C = planes[st];
                char pre[64]; snprintf(pre, sizeof(pre), "conv%d.%d", st + 2, b);
                const std::string P(pre);
                const int r1 = tensor(P + ".r1", N2, ho, wo, C, true), a1 = tensor(P + ".a1", N2, ho, wo, C, true);
                const int r2 = tensor(P + ".r2", N2, ho, wo, C, true), out = tensor(P + ".out", N2, ho, wo, C, true);
                conv(P + ".conv1", cur, -1, r1, 3, stride, 0, GACT_NONE, W_BOTH, W_BOTH);
                bn(P + ".bn1", r1, a1, -1, GACT_RELU, W_BOTH);
                conv(P + ".conv2", a1, -1, r2, 3, 1, 0, GACT_NONE, W_BOTH, W_BOTH);
                int idt = cur;
                if (stride != 1 || T[cur].C != C) {
                    const int rd = tensor(P + ".rd", N2, ho, wo, C, true), d = tensor(P + ".d", N2, ho, wo, C, true);
                    // module order inside the block: conv1, bn1, conv2, bn2, downsample (parameter order of the reference)
                    bn(P + ".bn2", r2, out, d, GACT_NONE, W_BOTH);
                    Op bn2 = ops.back(); ops.pop_back();
                    conv(P + ".downsample.0", cur, -1, rd, 1, stride, 0, GACT_NONE, W_BOTH, W_BOTH);
                    bn(P + ".downsample.1", rd, d, -1, GACT_NONE, W_BOTH);
                    ops.push_back(bn2);
                    if (bn2.fused_from >= 0) ops[bn2.fused_from].stat_to = (int)ops.size() - 1;     // bn2 moved behind the downsample ops
                    idt = d;
                } else {
                    bn(P + ".bn2", r2, out, idt, GACT_NONE, W_BOTH);
                }
                cur = out; hh = ho; ww = wo;
            }
            fe[st + 2] = cur;
        }
        {   // conv6 = conv_bn_relu(512, 512, 3, stride 2) (:409)
            const int r = tensor("conv6.r", N2, H16, W16, 512, true), o = tensor("fe6", N2, H16, W16, 512, true);
            conv("conv6.0", fe[5], -1, r, 3, 2, 0, GACT_NONE, W_BOTH, W_BOTH);
            bn("conv6.1", r, o, -1, GACT_LRELU, W_BOTH);
            fe[6] = o; t_fe6 = o;
        }
        // shared decoder (:413-424): transposed convs over [decoder | encoder skip]
        auto dec = [&](const char* name, int a, int b, int h, int w, int cout) {
            const int r = tensor(std::string(name) + ".r", N, h, w, cout, true), o = tensor(std::string(name) + ".o", N, h, w, cout, true);
            conv(std::string(name) + ".0", a, b, r, 3, 2, 1, GACT_NONE, W_GRAD, W_GRAD);
            bn(std::string(name) + ".1", r, o, -1, GACT_LRELU, W_GRAD);
            return o;
        };
        const int fd5 = dec("dec5", fe[6], -1, H8, W8, 256);
        const int fd4 = dec("dec4", fd5, fe[5], H4, W4, 128);
        const int fd3 = dec("dec3", fd4, fe[4], H2, W2, 64);
        const int fd2 = dec("dec2", fd3, fe[3], H, W, 64);
        auto head1 = [&](const char* name, int cout) {
            const int r = tensor(std::string(name) + ".r", N, H, W, cout, true), o = tensor(std::string(name) + ".o", N, H, W, cout, true);
            conv(std::string(name) + ".0", fd2, fe[2], r, 3, 1, 0, GACT_NONE, W_GRAD, W_GRAD);
            bn(std::string(name) + ".1", r, o, -1, GACT_LRELU, W_GRAD);
            return o;
        };
        // module order: id_dec1, id_dec0, gd_dec1, gd_dec0, cf_dec1, cf_dec0 (:428-452)
        const int id1 = head1("id_dec1", 64);
        t_pred = tensor("pred_init", N, H, W, 1, true);
        conv("id_dec0.0", id1, fe1, t_pred, 3, 1, 0, GACT_LRELU, W_GRAD, W_GRAD);
        const int gd1 = head1("gd_dec1", 64);
        const int guide = tensor("guide", N, H, W, 8, true);
        conv("gd_dec0.0", gd1, fe1, guide, 3, 1, 0, GACT_NONE, W_GRAD, W_GRAD);
        const int cf1 = head1("cf_dec1", 32);
        t_conf = tensor("confidence", N, H, W, 1, true);
        conv("cf_dec0.0", cf1, fe1, t_conf, 3, 1, 0, GACT_SIGMOID, W_GRAD, W_GRAD);
        t_oa = tensor("offset_aff", N, H, W, 24, true);
        conv("prop_layer.conv_offset_aff", guide, -1, t_oa, 3, 1, 0, GACT_NONE, W_GRAD, W_GRAD);
        // heads (:1340-1342, :932-934): emb = pred(proj(fe6 of the proxy pass)), ref = proj_t(fe6 of the grad pass)
        auto mlp = [&](const char* name, int x, int xw, int din, int dout, bool bwd) {
            (void)din;
            const int r = tensor(std::string(name) + ".h", N, H16, W16, 1024, bwd), a = tensor(std::string(name) + ".a", N, H16, W16, 1024, bwd);
            const int o = tensor(std::string(name) + ".out", N, H16, W16, dout, bwd);
            conv(std::string(name) + ".0", x, -1, r, 1, 1, 0, GACT_NONE, xw, W_GRAD, true, bwd);
            bn(std::string(name) + ".1", r, a, -1, GACT_RELU, W_GRAD, true, true, bwd);
            conv(std::string(name) + ".3", a, -1, o, 1, 1, 0, GACT_NONE, W_GRAD, W_GRAD, true, bwd);
            return o;
        };
        const int pz = mlp("proj", fe[6], W_PROXY, 512, 1024, false);
        t_emb = mlp("pred", pz, W_GRAD, 1024, 1024, false);
        tid["emb"] = t_emb;
        t_ref = mlp("proj_t", fe[6], W_GRAD, 512, 1024, true);
        tid["ref"] = t_ref;

        // ---- backward bookkeeping: first writer of every gradient buffer overwrites, later ones accumulate ----
        std::vector<char> written(T.size(), 0);
        auto first = [&](int t) { const bool f = !written[t]; written[t] = 1; return f; };
        first(t_pred); first(t_oa); first(t_conf); first(t_ref);      // written by the propagation / loss gradients
        for (int i = (int)ops.size() - 1; i >= 0; --i) {
            Op& o = ops[i];
            if (!o.bwd) continue;
            if (o.kind == K_CONV) {
                for (int s = 0; s < o.nsrc; ++s) if (T[o.x[s]].need_grad) o.first_x[s] = first(o.x[s]);
            } else {
                o.first_raw = first(o.x[0]);
                if (o.res >= 0) o.first_res = first(o.res);
            }
        }
        // ---- remaining workspace ----
        const long P = (long)H * W;
        off9 = falloc((size_t)N * P * 18); aff9 = falloc((size_t)N * P * 9);
        goff9 = falloc((size_t)N * P * 18); gaff9 = falloc((size_t)N * P * 9);
        feats = falloc((size_t)(PROP_TIME + 1) * N * P);
        depth = falloc((size_t)N * P); gdepth = falloc((size_t)N * P); gy = falloc((size_t)N * P); gping = falloc((size_t)N * P);
        validity_tmp = falloc((size_t)N * P);
        gall = falloc((size_t)gall_n);
        meta_wf = falloc(48 * 48 * 9); meta_wb = falloc(48 * 48 * 9);
        hyper = falloc(8); loss_info = falloc(4); S = falloc(1); w3_tmp = falloc(4);
        loss_ws = falloc((size_t)ptta_loss_ws_floats(N, H, W, rows()));
        step_dev = (int*)dalloc(sizeof(int));
        bn_part = falloc((size_t)ptta_gbn_part_floats(1024, 2)); bn_bw = falloc(3 * 1024);
        { const size_t a = (size_t)ptta_gwgrad_slabs((long)N * P) * (9 * 48 * 48 + 48), b = (size_t)ptta_gwgrad_mfma_part_floats((long)N * P, 48, 48); wg_part = falloc(a > b ? a : b); }
        for (auto& kv : convs) {
            GConvW& cw = kv.second;
            const int KK = cw.k * cw.k;
            if (cw.mf) { const size_t n = (size_t)ptta_gfrag_elems(KK, cw.C0, cw.C1, cw.Co); cw.ff_hi = (bf16_t*)dalloc(n * 2); cw.ff_lo = (bf16_t*)dalloc(n * 2); }
            if (cw.mb && !cw.Ci_real && cw.Co_pad != cw.Co) cw.gpad = falloc((size_t)N * T[0].H * T[0].W * cw.Co_pad);
            if (cw.mb && !cw.Ci_real) { const size_t n = (size_t)ptta_gfrag_elems(KK, cw.Co, 0, cw.Ci); cw.fb_hi = (bf16_t*)dalloc(n * 2); cw.fb_lo = (bf16_t*)dalloc(n * 2); }
            if (kv.first == "conv1_rgb_meta") continue;
            const size_t n = (size_t)KK * cw.Ci * cw.Co;
            cw.wf = falloc(n); cw.wb = falloc(n); cw.bias = falloc(cw.Co);
        }
    }
    long rows() const { return (long)N * (H / 16) * (W / 16); }

    // ---- weights ---------------------------------------------------------------------------------------------------
    int load(const char* name_c, const void* tensor_, const int64_t* shape, int ndim, hipStream_t s) {
        const std::string name(name_c);
        const float* src = (const float*)tensor_;
        auto ends = [&](const char* suf) { const size_t l = strlen(suf); return name.size() >= l && name.compare(name.size() - l, l, suf) == 0; };
        if (ends("num_batches_tracked") || ends("running_mean") || ends("running_var")) return 0;   // dropped by adapt_parameters('meta_bn') / train-mode BN1d
        if (aid.count(name)) return 0;                                   // adapted: the bound tensor is authoritative
        if (name == "prop_layer.aff_scale_const") { NCHK(hipMemcpyAsync(S, src, sizeof(float), hipMemcpyDeviceToDevice, s)); return 0; }
        if (name == "prop_layer.w" || name == "prop_layer.b" || name == "prop_layer.w_conf") return 0;   // constants ones / zero (nlspnmodel_adapt.py:239-247)
        const size_t dot = name.rfind('.');
        if (dot == std::string::npos) return fail("unknown state_dict key " + name, -2);
        const std::string base = name.substr(0, dot), leaf = name.substr(dot + 1);
        auto fb = frozen_bn.find(base);
        if (fb != frozen_bn.end()) {
            float* dst = leaf == "weight" ? fb->second.first : fb->second.second;
            NCHK(hipMemcpyAsync(dst, src, (size_t)shape[0] * sizeof(float), hipMemcpyDeviceToDevice, s));
            return 0;
        }
        auto it = convs.find(base);
        if (it == convs.end()) return fail("unknown state_dict key " + name, -2);
        GConvW& cw = it->second;
        if (leaf == "bias") {
            if (ndim != 1 || shape[0] != cw.Co) return fail("shape mismatch for " + name, -22);
            NCHK(hipMemcpyAsync(cw.bias, src, (size_t)cw.Co * sizeof(float), hipMemcpyDeviceToDevice, s));
            cw.has_bias = true;
            return 0;
        }
        if (leaf != "weight") return fail("unknown state_dict key " + name, -2);
        const int KK = cw.k * cw.k;
        const int Ci = cw.Ci_real ? cw.Ci_real : cw.Ci;
        if (cw.k == 1 && ndim == 2) {                  // nn.Linear (N, K)
            if (shape[0] != cw.Co || shape[1] != Ci) return fail("shape mismatch for " + name, -22);
        } else if (ndim != 4 || shape[2] != cw.k || shape[3] != cw.k ||
                   (cw.transposed ? (shape[0] != Ci || shape[1] != cw.Co) : (shape[0] != cw.Co || shape[1] != Ci)))
            return fail("shape mismatch for " + name, -22);
        if (!cw.transposed) {
            // forward P[t][ci][co] = W[co][ci][t]; data gradient P[t][co][ci] = W[co][ci][flip t] (stride 1: a conv with
            // flipped taps; stride 2: consumed by the transposed kernel, which wants the taps unflipped)
            ptta_gpack(src, cw.wf, KK, Ci, cw.Co, KK, (long)Ci * KK, 0, s);
            ptta_gpack(src, cw.wb, KK, cw.Co, Ci, (long)Ci * KK, KK, cw.stride == 1 ? 1 : 0, s);
        } else {
            // ConvTranspose2d weight (Ci, Co, k, k): forward P[t][ci][co] = W[ci][co][t]; gradient = stride-2 conv with
            // P[t][co][ci] = W[ci][co][t]
            ptta_gpack(src, cw.wf, KK, Ci, cw.Co, (long)cw.Co * KK, KK, 0, s);
            ptta_gpack(src, cw.wb, KK, cw.Co, Ci, KK, (long)cw.Co * KK, 0, s);
        }
        pack_frags(cw, cw.wf, cw.wb, s);
        cw.loaded = true;
        return 0;
    }
    void pack_frags(GConvW& cw, const float* wf, const float* wb, hipStream_t s) {
        const int KK = cw.k * cw.k;
        if (cw.mf && cw.Ci_real) ptta_gfrag_pack(wf, cw.Co, (long)cw.Ci_real * cw.Co, KK, cw.Ci_real, 0, 0, 0, cw.Co, cw.ff_hi, cw.ff_lo, s);
        else if (cw.mf) ptta_gfrag_pack(wf, cw.Co, (long)cw.Ci * cw.Co, KK, cw.C0, cw.C1, 0, cw.C0, cw.Co, cw.ff_hi, cw.ff_lo, s);
        if (cw.mb && !cw.Ci_real) ptta_gfrag_pack(wb, cw.Ci, (long)cw.Co * cw.Ci, KK, cw.Co, 0, 0, 0, cw.Ci, cw.fb_hi, cw.fb_lo, s);
    }
};


namespace {

// ---- execution ---------------------------------------------------------------------------------------------------------
int run_conv_fwd(nlspn_engine* e, const Op& o, bool train, hipStream_t s) {
    const bool meta = o.ad_w >= 0;
    const GConvW& cw = e->convs[o.wname];
    if (!meta && !cw.loaded) return e->fail("weights of " + o.wname + " not loaded (ptta_load_weights)", -3);
    const float* wf = meta ? e->meta_wf : cw.wf;
    const float* bias = meta ? e->adapted[o.ad_b].p : (cw.has_bias ? cw.bias : nullptr);
    if (cw.mf) {
        GX3Args a;
        const GView x0 = e->view(o.x[0], o.xw[0], train), y = e->view(o.y, o.yw, train);
        a.x0 = x0.p; a.C0 = x0.C; a.ld0 = x0.ld;
        if (o.nsrc == 2) { const GView x1 = e->view(o.x[1], o.xw[1], train); a.x1 = x1.p; a.C1 = x1.C; a.ld1 = x1.ld; }
        a.B = y.B; a.H = y.H; a.W = y.W;
        a.whi = (const uint4*)cw.ff_hi; a.wlo = (const uint4*)cw.ff_lo;
        a.nchunks = (a.C0 + 31) / 32 + (a.C1 + 31) / 32; a.nf0 = 0; a.nnf = (cw.Co + 31) / 32;
        a.y = y.p; a.ldy = y.ld; a.Cy = y.C; a.bias = bias; a.act = o.act;
        if (o.stat_to >= 0) { a.stat_part = e->ops[o.stat_to].part; a.stat_C = y.C; a.stat_npass = (o.yw == W_BOTH && train) ? 2 : 1; }
        int rc;
        if (o.stride == 1 && !o.transposed) rc = ptta_launch_gconv_x3(a, o.k, s);
        else rc = ptta_launch_gconv_x3_strided(a, o.k, o.transposed ? 2 : 1, x0.H, x0.W, s);
        if (rc) return e->fail("conv " + o.wname + " (matrix-core) launch failed", -5);
        return 0;
    }
    for (int sidx = 0; sidx < o.nsrc; ++sidx) {
        GConvArgs a;
        a.x = e->view(o.x[sidx], o.xw[sidx], train); a.y = e->view(o.y, o.yw, train);
        if (o.xw[sidx] == W_PROXY) a.x.B = e->N;
        a.w = wf + (size_t)o.c0[sidx] * cw.Co; a.wld = cw.Co; a.wts = (long)cw.Ci * cw.Co;
        a.bias = sidx == 0 ? bias : nullptr;
        a.k = o.k; a.stride = o.stride; a.transposed = o.transposed;
        a.accumulate = sidx > 0; a.act = sidx == o.nsrc - 1 ? o.act : GACT_NONE;
        const int rc = ptta_launch_gconv_direct(a, s);
        if (rc) return e->fail("conv " + o.wname + " launch failed", -5);
    }
    return 0;
}

const float* bn_gamma(nlspn_engine* e, const Op& o) { return o.ad_g >= 0 ? e->adapted[o.ad_g].p : e->frozen_bn[o.bname].first; }
const float* bn_beta(nlspn_engine* e, const Op& o) { return o.ad_beta >= 0 ? e->adapted[o.ad_beta].p : e->frozen_bn[o.bname].second; }

int run_bn_fwd(nlspn_engine* e, const Op& o, bool train, hipStream_t s) {
    const GView x = e->view(o.x[0], o.xw[0], train), y = e->view(o.y, o.yw, train);
    GView res; if (o.res >= 0) res = e->view(o.res, o.xw[0], train);
    const int npass = (o.xw[0] == W_BOTH && train) ? 2 : 1;
    const int fused = o.fused_from >= 0 ? ptta_gconv_x3_tiles(x.B / npass, x.H, x.W) : 0;
    if (ptta_launch_gbn_forward(x, res, y, npass, o.act, BN_EPS, bn_gamma(e, o), bn_beta(e, o), fused ? o.part : e->bn_part, o.st, s, fused))
        return e->fail("batch-norm " + o.bname + " launch failed", -5);
    return 0;
}

int forward(nlspn_engine* e, const float* image, const float* sparse, bool train, hipStream_t s) {
    for (auto& ad : e->adapted) if (!ad.p) return e->fail("adapted parameter " + ad.name + " not bound (ptta_bind_adapted)", -3);
    const int N = e->N, H = e->H, W = e->W;
    const long P = (long)H * W;
    const int Be = train ? 2 * N : N;
    // inputs: image -> NHWC (normalised on the fly; proxy half = zero image), sparse depth clamped (external_model_adapt.py:108)
    GView iv = e->view(e->t_img, W_BOTH, train);
    if (ptta_launch_gnchw_to_nhwc(image, N, 3, iv, N, e->norm_on, e->norm_div, e->norm_mean, e->norm_std, s)) return e->fail("image staging failed", -5);
    hipLaunchKernelGGL(clamp_dup_kernel, dim3(nb(P * Be)), dim3(256), 0, s, sparse, e->T[e->t_sd].p, (long)N * P, Be / N, e->hp.max_input_depth);
    if (e->t_sd16 >= 0) {
        GView sv = e->view(e->t_sd16, W_BOTH, train);
        if (ptta_launch_gnchw_to_nhwc(e->T[e->t_sd].p, Be, 1, sv, Be, 0, 1.f, nullptr, nullptr, s)) return e->fail("sparse staging failed", -5);
    }
    // the adapted conv is re-packed from the bound tensor on every forward
    ptta_gpack(e->adapted[0].p, e->meta_wf, 9, 48, 48, 9, 48L * 9, 0, s);
    ptta_gpack(e->adapted[0].p, e->meta_wb, 9, 48, 48, 48L * 9, 9, 1, s);
    e->pack_frags(e->convs["conv1_rgb_meta"], e->meta_wf, e->meta_wb, s);
    for (const Op& o : e->ops) {
        if (o.train_only && !train) continue;
        const int rc = o.kind == K_CONV ? run_conv_fwd(e, o, train, s) : run_bn_fwd(e, o, train, s);
        if (rc) return rc;
    }
    // propagation (nlspnmodel_adapt.py:340-373) and the final clamp (:900)
    const GView oa = e->view(e->t_oa, W_GRAD, train);
    if (ptta_launch_nl_affinity_fwd(oa, e->T[e->t_conf].p, e->S, e->legacy, e->off9, e->aff9, s)) return e->fail("affinity launch failed", -5);
    if (hipMemcpyAsync(e->feats, e->T[e->t_pred].p, (size_t)N * P * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    for (int k = 0; k < PROP_TIME; ++k)
        if (ptta_launch_nl_prop_fwd(e->feats + (size_t)k * N * P, e->T[e->t_sd].p, e->off9, e->aff9, e->feats + (size_t)(k + 1) * N * P, N, H, W, s))
            return e->fail("propagation launch failed", -5);
    hipLaunchKernelGGL(relu_copy_kernel, dim3(nb((long)N * P)), dim3(256), 0, s, e->feats + (size_t)PROP_TIME * N * P, e->depth, (long)N * P);
    if (hipGetLastError() != hipSuccess) return e->fail("launch failed", -5);
    e->fwd_valid = train;
    return 0;
}

int run_conv_bwd(nlspn_engine* e, const Op& o, hipStream_t s) {
    const bool meta = o.ad_w >= 0;
    const GConvW& cw = e->convs[o.wname];
    GView gy = e->view(o.y, W_GRAD, true, true);
    const GView yv = e->view(o.y, W_GRAD, true);
    if (o.act != GACT_NONE && ptta_launch_gact_bwd(gy, yv, o.act, s)) return e->fail("activation gradient failed", -5);
    bool padded = false;
    for (int sidx = 0; sidx < o.nsrc; ++sidx) {
        if (!e->T[o.x[sidx]].need_grad) continue;
        if (cw.mb && (o.c0[sidx] % 32) == 0) {
            GX3Args a;
            const GView gx = e->view(o.x[sidx], W_GRAD, true, true);
            a.x0 = gy.p; a.C0 = gy.C; a.ld0 = gy.ld;
            if (cw.gpad) {                                   // < 16 gradient channels: matrix-core kernel on a zero-padded copy
                if (sidx == 0 || !padded) {
                    hipLaunchKernelGGL(pad_channels_kernel, dim3(nb((long)gy.B * gy.H * gy.W * cw.Co_pad)), dim3(256), 0, s, gy.p, gy.ld, gy.C,
                                       cw.gpad, cw.Co_pad, (long)gy.B * gy.H * gy.W);
                    padded = true;
                }
                a.x0 = cw.gpad; a.C0 = cw.Co_pad; a.ld0 = cw.Co_pad;
            }
            a.B = gy.B; a.H = gy.H; a.W = gy.W;
            a.whi = (const uint4*)cw.fb_hi; a.wlo = (const uint4*)cw.fb_lo;
            a.nchunks = (a.C0 + 31) / 32; a.nf0 = o.c0[sidx] / 32; a.nnf = (gx.C + 31) / 32;
            a.y = gx.p; a.ldy = gx.ld; a.Cy = gx.C; a.accumulate = o.first_x[sidx] ? 0 : 1;
            a.B = gx.B; a.H = gx.H; a.W = gx.W;                  // output geometry = the source's
            int rc;
            if (o.stride == 1 && !o.transposed) rc = ptta_launch_gconv_x3(a, o.k, s);
            else rc = ptta_launch_gconv_x3_strided(a, o.k, o.transposed ? 1 : 2, gy.H, gy.W, s);     // convT -> strided conv, strided conv -> convT
            if (rc) return e->fail("data gradient of " + o.wname + " (matrix-core) failed", -5);
            continue;
        }
        GConvArgs a;
        a.x = gy; a.y = e->view(o.x[sidx], W_GRAD, true, true);
        a.w = (meta ? e->meta_wb : cw.wb) + o.c0[sidx]; a.wld = cw.Ci; a.wts = (long)cw.Co * cw.Ci;
        a.k = o.k; a.act = GACT_NONE; a.accumulate = o.first_x[sidx] ? 0 : 1;
        if (o.transposed) { a.transposed = 0; a.stride = o.stride; }           // convT -> strided conv
        else if (o.stride == 2) { a.transposed = 1; a.stride = 2; }             // strided conv -> convT
        else { a.transposed = 0; a.stride = 1; }
        if (ptta_launch_gconv_direct(a, s)) return e->fail("data gradient of " + o.wname + " failed", -5);
    }
    if (meta) {
        const GView xv = e->view(o.x[0], W_GRAD, true);
        if ((e->naive ? ptta_launch_gwgrad(xv, gy, e->wg_part, e->gall + e->adapted[o.ad_w].goff, e->gall + e->adapted[o.ad_b].goff, s)
                      : ptta_launch_gwgrad_mfma(xv, gy, e->wg_part, e->gall + e->adapted[o.ad_w].goff, e->gall + e->adapted[o.ad_b].goff, s)))
            return e->fail("weight gradient failed", -5);
    }
    return 0;
}

int run_bn_bwd(nlspn_engine* e, const Op& o, hipStream_t s) {
    const GView x = e->view(o.x[0], W_GRAD, true), y = e->view(o.y, W_GRAD, true);
    const GView g = e->view(o.y, W_GRAD, true, true), gx = e->view(o.x[0], W_GRAD, true, true);
    GView gres; if (o.res >= 0) gres = e->view(o.res, W_GRAD, true, true);
    const int npass = o.xw[0] == W_BOTH ? 2 : 1;
    float* dg = o.ad_g >= 0 ? e->gall + e->adapted[o.ad_g].goff : nullptr;
    float* db = o.ad_beta >= 0 ? e->gall + e->adapted[o.ad_beta].goff : nullptr;
    if (ptta_launch_gbn_backward(x, g, y, gx, gres, npass, o.act, o.res >= 0 ? 1 : 0, o.first_raw ? 0 : 1, o.first_res ? 0 : 1,
                                 bn_gamma(e, o), o.st, e->bn_part, e->bn_bw, dg, db, s))
        return e->fail("batch-norm gradient of " + o.bname + " failed", -5);
    return 0;
}

// consumes e->gdepth (N,1,H,W) and the gradient of `ref` already stored in T[t_ref].g
int backward(nlspn_engine* e, hipStream_t s) {
    if (!e->fwd_valid) return e->fail("backward without a training forward", -3);
    const int N = e->N, H = e->H, W = e->W;
    const long NP = (long)N * H * W;
    hipLaunchKernelGGL(clamp_bwd_kernel, dim3(nb(NP)), dim3(256), 0, s, e->gdepth, e->feats + (size_t)PROP_TIME * NP, e->gy, NP);
    if (hipMemsetAsync(e->goff9, 0, (size_t)NP * 18 * sizeof(float), s) != hipSuccess) return e->fail("memset failed", -5);
    if (hipMemsetAsync(e->gaff9, 0, (size_t)NP * 9 * sizeof(float), s) != hipSuccess) return e->fail("memset failed", -5);
    float* gcur = e->gy; float* gnext = e->gping;
    for (int k = PROP_TIME - 1; k >= 0; --k) {
        float* dst = k == 0 ? e->T[e->t_pred].g : gnext;
        if (hipMemsetAsync(dst, 0, (size_t)NP * sizeof(float), s) != hipSuccess) return e->fail("memset failed", -5);
        if (ptta_launch_nl_prop_bwd(e->feats + (size_t)k * NP, e->T[e->t_sd].p, e->off9, e->aff9, gcur, dst, e->goff9, e->gaff9, N, H, W, s))
            return e->fail("propagation gradient failed", -5);
        float* t = gcur; gcur = dst; gnext = t;
    }
    if (hipMemsetAsync(e->T[e->t_conf].g, 0, (size_t)NP * sizeof(float), s) != hipSuccess) return e->fail("memset failed", -5);
    if (ptta_launch_nl_affinity_bwd(e->view(e->t_oa, W_GRAD, true), e->T[e->t_conf].p, e->S, e->legacy, e->goff9, e->gaff9,
                                    e->view(e->t_oa, W_GRAD, true, true), e->T[e->t_conf].g, s))
        return e->fail("affinity gradient failed", -5);
    for (int i = (int)e->ops.size() - 1; i >= 0; --i) {
        const Op& o = e->ops[i];
        if (!o.bwd) continue;
        const int rc = o.kind == K_CONV ? run_conv_bwd(e, o, s) : run_bn_bwd(e, o, s);
        if (rc) return rc;
    }
    return 0;
}

int upload_hparams(nlspn_engine* e, hipStream_t s) {
    const ptta_hparams& hp = e->hp;
    const float h8[8] = {hp.lr, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.w_sparse_depth, hp.w_smoothness, hp.w_cos};
    if (ptta_launch_set_floats(e->hyper, h8, 8, s)) return e->fail("hyper-parameter upload failed", -5);     // by kernel argument: no sync
    return 0;
}

}  // namespace

nlspn_engine* nlspn_create(int n, int h, int w, const ptta_hparams* hp, int legacy_offset, int* rc) {
    *rc = 0;
    if (n < 1 || h < 16 || w < 16 || !hp) { *rc = -22; return nullptr; }
    if ((h % 16) || (w % 16)) { *rc = -38; return nullptr; }      // decoder crops of nlspnmodel_adapt.py:474-490 are not implemented
    nlspn_engine* e = new nlspn_engine();
    e->N = n; e->H = h; e->W = w; e->hp = *hp; e->legacy = legacy_offset ? 1 : 0;
    const char* impl = getenv("PTTA_CONV_IMPL");
    e->naive = (impl && strcmp(impl, "naive") == 0) ? 1 : 0;          // direct fp32 kernels everywhere (validation)
    e->build();
    if (e->oom || !e->step_dev) { nlspn_destroy(e); *rc = -12; return nullptr; }
    const float one[1] = {4.0f};                                  // affinity_gamma * num = 0.5 * 8 (nlspnmodel_adapt.py:231-233)
    if (hipMemcpy(e->S, one, sizeof(one), hipMemcpyHostToDevice) != hipSuccess) { nlspn_destroy(e); *rc = -5; return nullptr; }
    if (upload_hparams(e, nullptr)) { nlspn_destroy(e); *rc = -5; return nullptr; }
    return e;
}
void nlspn_destroy(nlspn_engine* e) {
    if (!e) return;
    for (void* p : e->allocs) if (p) (void)hipFree(p);
    delete e;
}
const char* nlspn_last_error(nlspn_engine* e) { return e->err.c_str(); }
int nlspn_set_hparams(nlspn_engine* e, const ptta_hparams* hp, hipStream_t s) { e->hp = *hp; return upload_hparams(e, s); }
int nlspn_set_image_norm(nlspn_engine* e, float div, const float* mean, const float* stdv) {
    if (!(div > 0.f)) return e->fail("ptta_set_image_norm: divisor must be positive", -22);
    e->norm_div = div;
    for (int k = 0; k < 3; ++k) { e->norm_mean[k] = mean ? mean[k] : 0.f; e->norm_std[k] = stdv ? stdv[k] : 1.f; if (!(e->norm_std[k] > 0.f)) return e->fail("std must be positive", -22); }
    e->norm_on = !(div == 1.f && e->norm_mean[0] == 0.f && e->norm_mean[1] == 0.f && e->norm_mean[2] == 0.f && e->norm_std[0] == 1.f &&
                   e->norm_std[1] == 1.f && e->norm_std[2] == 1.f);
    return 0;
}
int nlspn_load_weights(nlspn_engine* e, const char* name, const void* tensor, const int64_t* shape, int ndim, hipStream_t s) {
    if (!name || !tensor) return -1;
    return e->load(name, tensor, shape, ndim, s);
}
int nlspn_bind_adapted(nlspn_engine* e, const char* name, float* p, float* m, float* v) {
    auto it = e->aid.find(name ? name : "");
    if (it == e->aid.end()) return e->fail(std::string("not an adapted parameter: ") + (name ? name : "(null)"), -2);
    Adapted& a = e->adapted[it->second];
    a.p = p; a.m = m; a.v = v;
    return 0;
}
int nlspn_adapted_count(nlspn_engine* e) { return (int)e->adapted.size(); }
const char* nlspn_adapted_name(nlspn_engine* e, int index, int64_t* numel) {
    if (index < 0 || index >= (int)e->adapted.size()) return nullptr;
    if (numel) *numel = e->adapted[index].n;
    return e->adapted[index].name.c_str();
}
int nlspn_set_adam_step(nlspn_engine* e, int step, hipStream_t s) {
    return ptta_launch_set_int(e->step_dev, step, s) ? e->fail("set step failed", -5) : 0;
}
int nlspn_get_adam_step(nlspn_engine* e, int* step, hipStream_t s) {
    if (hipMemcpyAsync(step, e->step_dev, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return e->fail("memcpy failed", -5);
    return hipStreamSynchronize(s) == hipSuccess ? 0 : e->fail("sync failed", -5);
}
int64_t nlspn_embedding_rows(nlspn_engine* e) { return e->rows(); }

int nlspn_forward_train(nlspn_engine* e, const float* image, const float* sparse, float* depth, float* emb, float* ref, hipStream_t s) {
    const int rc = forward(e, image, sparse, true, s);
    if (rc) return rc;
    const size_t nb_ = (size_t)e->N * e->H * e->W * sizeof(float), eb = (size_t)e->rows() * 1024 * sizeof(float);
    if (depth && hipMemcpyAsync(depth, e->depth, nb_, hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    if (emb && hipMemcpyAsync(emb, e->T[e->t_emb].p, eb, hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    if (ref && hipMemcpyAsync(ref, e->T[e->t_ref].p, eb, hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    return 0;
}
int nlspn_forward_eval(nlspn_engine* e, const float* image, const float* sparse, float* depth, hipStream_t s) {
    const int rc = forward(e, image, sparse, false, s);
    if (rc) return rc;
    if (depth && hipMemcpyAsync(depth, e->depth, (size_t)e->N * e->H * e->W * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    return 0;
}
int nlspn_step(nlspn_engine* e, const float* image, const float* loss_image, const float* sparse, const float* validity, float* depth_out,
               float* loss_info_out, hipStream_t s) {
    for (auto& ad : e->adapted) if (!ad.m || !ad.v) return e->fail("Adam state of " + ad.name + " not bound", -3);
    if (!loss_image) loss_image = image;
    int rc = forward(e, image, sparse, true, s);
    if (rc) return rc;
    const long NP = (long)e->N * e->H * e->W;
    if (!validity) {
        hipLaunchKernelGGL(validity_kernel, dim3(nb(NP)), dim3(256), 0, s, sparse, e->validity_tmp, NP);
        validity = e->validity_tmp;
    }
    const float* emb = e->T[e->t_emb].p; const float* ref = e->T[e->t_ref].p;
    if (ptta_launch_loss_forward(e->depth, loss_image, sparse, validity, e->hp.max_input_depth, emb, ref, e->rows(), 1024, e->hyper + 5,
                                 e->N, e->H, e->W, e->loss_ws, e->loss_info, s)) return e->fail("loss forward failed", -5);
    if (ptta_launch_loss_backward(e->depth, loss_image, sparse, validity, e->hp.max_input_depth, emb, ref, e->rows(), 1024, e->N, e->H,
                                  e->W, e->loss_ws, e->gdepth, e->T[e->t_ref].g, s)) return e->fail("loss backward failed", -5);
    rc = backward(e, s);
    if (rc) return rc;
    if (ptta_launch_step_inc(e->step_dev, s)) return e->fail("step counter failed", -5);
    for (auto& ad : e->adapted)
        if (ptta_launch_adam(ad.p, ad.m, ad.v, e->gall + ad.goff, ad.n, e->hyper, e->step_dev, s)) return e->fail("adam failed", -5);
    if (depth_out && hipMemcpyAsync(depth_out, e->depth, (size_t)NP * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    if (loss_info_out && hipMemcpyAsync(loss_info_out, e->loss_info, 4 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    e->fwd_valid = false;
    return 0;
}
int nlspn_get_grad(nlspn_engine* e, const char* name, float* dst, int64_t capacity, hipStream_t s) {
    auto it = e->aid.find(name ? name : "");
    if (it == e->aid.end()) return e->fail(std::string("not an adapted parameter: ") + (name ? name : "(null)"), -2);
    const Adapted& a = e->adapted[it->second];
    if (capacity < a.n) return e->fail("ptta_get_grad: destination too small", -22);
    if (hipMemcpyAsync(dst, e->gall + a.goff, (size_t)a.n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    return 0;
}
int nlspn_debug_tensor(nlspn_engine* e, const char* name, float* dst, int64_t capacity, int64_t* numel, hipStream_t s) {
    std::string nm(name ? name : "");
    bool grad = false;
    if (nm.size() > 5 && nm.compare(0, 5, "grad:") == 0) { grad = true; nm = nm.substr(5); }
    const float* src = nullptr; long n = 0;
    const long NP = (long)e->N * e->H * e->W;
    if (nm == "depth") { src = e->depth; n = NP; }
    else if (nm == "off9") { src = e->off9; n = NP * 18; }
    else if (nm == "aff9") { src = e->aff9; n = NP * 9; }
    else if (nm == "goff9") { src = e->goff9; n = NP * 18; }
    else if (nm == "gaff9") { src = e->gaff9; n = NP * 9; }
    else {
        auto it = e->tid.find(nm);
        if (it == e->tid.end()) return e->fail("unknown debug tensor " + nm, -2);
        const Tn& t = e->T[it->second];
        if (t.ld != t.C) return e->fail("debug tensor " + nm + " is a strided slice", -22);
        src = grad ? t.g : t.p; n = (long)(grad ? e->N : t.items) * t.H * t.W * t.C;
        if (!src) return e->fail("debug tensor " + nm + " has no gradient buffer", -2);
    }
    if (numel) *numel = n;
    if (!dst) return 0;
    if (capacity < n) return e->fail("debug tensor: destination too small", -22);
    if (hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    return 0;
}

// ---- the reference's split calls: compute_loss / loss.backward() / optimizer.step() (src/tta_main.py:619-633) ----------
int nlspn_loss_forward(nlspn_engine* e, const float* loss_image, const float* depth, const float* sparse, const float* validity,
                       const float* emb, const float* ref, int64_t rows, float w_sd, float w_sm, float w_cos, float* loss_info_out, hipStream_t s) {
    if (rows > e->rows()) return e->fail("rows exceeds the handle's embedding rows", -22);
    const float w3[3] = {w_sd, w_sm, w_cos};
    if (ptta_launch_set_floats(e->w3_tmp, w3, 3, s)) return e->fail("loss weight upload failed", -5);
    if (ptta_launch_loss_forward(depth, loss_image, sparse, validity, e->hp.max_input_depth, emb, ref, rows, 1024, e->w3_tmp, e->N, e->H,
                                 e->W, e->loss_ws, loss_info_out, s)) return e->fail("loss forward failed", -5);
    return 0;
}
int nlspn_loss_backward(nlspn_engine* e, const float* loss_image, const float* depth, const float* sparse, const float* validity,
                        const float* emb, const float* ref, int64_t rows, float* grad_depth_out, float* grad_ref_out, hipStream_t s) {
    if (ptta_launch_loss_backward(depth, loss_image, sparse, validity, e->hp.max_input_depth, emb, ref, rows, 1024, e->N, e->H, e->W,
                                  e->loss_ws, grad_depth_out, grad_ref_out, s)) return e->fail("loss backward failed", -5);
    return 0;
}
int nlspn_backward(nlspn_engine* e, const float* grad_depth, const float* grad_ref, hipStream_t s) {
    if (!e->fwd_valid) return e->fail("ptta_backward without a preceding ptta_forward_train", -3);
    const size_t nb_ = (size_t)e->N * e->H * e->W * sizeof(float), rb = (size_t)e->rows() * 1024 * sizeof(float);
    if (hipMemcpyAsync(e->gdepth, grad_depth, nb_, hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    if (grad_ref) { if (hipMemcpyAsync(e->T[e->t_ref].g, grad_ref, rb, hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5); }
    else if (hipMemsetAsync(e->T[e->t_ref].g, 0, rb, s) != hipSuccess) return e->fail("memset failed", -5);
    return backward(e, s);
}
int nlspn_adam_step(nlspn_engine* e, hipStream_t s) {
    for (auto& ad : e->adapted) if (!ad.p || !ad.m || !ad.v) return e->fail("Adam state of " + ad.name + " not bound", -3);
    if (ptta_launch_step_inc(e->step_dev, s)) return e->fail("step counter failed", -5);
    for (auto& ad : e->adapted)
        if (ptta_launch_adam(ad.p, ad.m, ad.v, e->gall + ad.goff, ad.n, e->hyper, e->step_dev, s)) return e->fail("adam failed", -5);
    return 0;
}
int nlspn_set_grad(nlspn_engine* e, const char* name, const float* src, int64_t numel, hipStream_t s) {
    auto it = e->aid.find(name ? name : "");
    if (it == e->aid.end()) return e->fail(std::string("not an adapted parameter: ") + (name ? name : "(null)"), -2);
    const Adapted& a = e->adapted[it->second];
    if (numel != a.n) return e->fail("ptta_set_grad: size mismatch", -22);
    if (hipMemcpyAsync(e->gall + a.goff, src, (size_t)a.n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return e->fail("memcpy failed", -5);
    return 0;
}
