// Generic NHWC layer-graph engine of libptta_hip, shared by the backbones whose adapted set crosses the whole network
// (NLSPN: nlspn_api.hip, CostDCNet: costdc_api.hip).
//
// A network is a flat op list built once per handle ("program"):
//   CONV  one or two channel-concatenated NHWC sources (torch.cat never materialises), 3x3 / 1x1, stride 1 / 2 or stride-2
//         transposed, optional fused activation; matrix-core (bf16x3) kernels of gconv_mfma.hip, direct fp32 kernels of
//         gconv.hip for odd channel counts and for PTTA_CONV_IMPL=naive validation;
//   BN    batch-statistics normalisation (+activation, + residual add), adapted or frozen affine parameters, optionally
//         TRACKED (running statistics updated in train mode, used in eval mode: BatchNorm3d / BatchNorm1d of CostDCNet);
//   FUNC  backbone-specific kernels (pooling, resampling, fusion ...) as forward / backward closures.
// The grad pass and the no-grad proxy pass (zero image) of a training step are batched as [real | proxy] with separate
// batch statistics per pass.  The backward sweep is the op list in reverse: every data gradient is again a CONV with
// re-packed weights, BatchNorm gradients for gamma/beta, weight gradients for adapted convolutions, then Adam on device.
// The first writer of each gradient buffer overwrites, later ones accumulate (decided at build time): no gradient memsets.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "../../include/ptta.h"
#include "ptta_common.h"
#include "ptta_kernels.h"

long ptta_gwgrad_mfma_part_floats(long pixels, int Ci, int Co);
int ptta_launch_gbn_running_update(const float* st, int npass, int C, long R, float momentum, float eps, float* rm, float* rv, long long* nbt,
                                   int repeats, hipStream_t s);
int ptta_launch_gbn_eval_affine(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C, float* st, hipStream_t s);

#define NCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_), -100 - (int)e_); } while (0)
#define NRUN(x) do { int r_ = (x); if (r_ != 0) return fail(std::string(#x) + " failed", r_ < 0 ? r_ : -r_); } while (0)

namespace gnet {

enum { W_BOTH = 0, W_GRAD = 1, W_PROXY = 2 };
enum { K_CONV = 0, K_BN = 1, K_FUNC = 2 };
const float BN_EPS = 1e-5f;

struct Tn {
    std::string name;
    float *p = nullptr, *g = nullptr;
    int items = 0, per = 0, H = 0, W = 0, C = 0, ld = 0;      // per = items of ONE pass (frames, or frames x depth planes)
    bool need_grad = false;
};

struct GConvW {            // convolution weights, packed at load time (frozen) or at every forward (adapted)
    float *wf = nullptr, *wb = nullptr, *bias = nullptr;
    int Ci = 0, Co = 0, k = 3, stride = 1, transposed = 0;
    int C0 = 0, C1 = 0;                                   // source split of the input channels (torch.cat order)
    int Co_pad = 0;
    int Ci_real = 0;                                      // > 0: the state_dict tensor has fewer input channels than the zero-padded activation
    int vcol = 0;                                         // 3x1x1 Conv3d stored as a 3x3 filter with only the middle column set
    float* gpad = nullptr;                                // zero-padded copy of the output gradient (Co_pad channels)
    bf16_t *ff_hi = nullptr, *ff_lo = nullptr, *fb_hi = nullptr, *fb_lo = nullptr;     // bf16x3 MFMA fragments (forward / data gradient)
    bf16_t* ff_l2 = nullptr;                              // third operand plane of the forward fragments (GNet::x6)
    bf16_t *fb1_hi = nullptr, *fb1_lo = nullptr;          // data-gradient fragments of the SECOND source when its channel offset is not a multiple of 32
    bool mf = false, mb = false;                          // matrix-core kernel usable for forward / data gradient
    bool loaded = false, has_bias = false;
    long gpad_pix = 0;
};

struct Op {
    int kind = K_CONV;
    bool train_only = false, bwd = true;
    // conv
    int nsrc = 1, x[2] = {-1, -1}, c0[2] = {0, 0}, xw[2] = {W_BOTH, W_BOTH};
    int y = -1, yw = W_BOTH;
    int k = 3, stride = 1, transposed = 0, act = GACT_NONE;
    int rH = 0, rW = 0;                 // > 0: every tensor of this op is re-viewed as images of rH x rW (same memory):
                                        // a 3x1x1 Conv3d over [B][D][H][W][C] is a vertical 3-tap conv over [B][D][H*W][C]
    std::string wname;
    int ad_w = -1, ad_b = -1;           // adapted (bound) weight / bias
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
    bool tracked = false;             // running statistics kept: updated in train mode (momentum 0.1), used in eval mode
    bool act_first = false;           // with a residual: y = relu(act(bn(x)) + res) instead of relu(bn(x) + res)
    int stat_repeats = 1;             // the reference runs this layer `repeats` times per training forward on the same input
    float *rm = nullptr, *rv = nullptr; long long* nbt = nullptr;
    float *rm_own = nullptr, *rv_own = nullptr;   // BatchNorm2d (not tracked): the LOADED running statistics, used by the stage-2 head trainer only
    bool has_running = false;
    bool no_dx = false;               // (CONV) no data gradient into the source although it has a gradient buffer (head trainer: detached input)
    float* st_eval = nullptr;         // tracked: eval-mode affine [mean, inv, scale, shift][C] of the running statistics
    mutable bool st_eval_valid = false;   // written by the last training forward's finalize (or an eval forward); cleared by every load()
    // func: fx = tensors whose gradient buffers the backward closure writes (first_x[k] tells it to overwrite or accumulate)
    int fx[2] = {-1, -1};
    std::function<int(bool, hipStream_t)> ffwd;
    std::function<int(hipStream_t)> fbwd;
};

struct Adapted { std::string name; long n = 0, goff = 0; float *p = nullptr, *m = nullptr, *v = nullptr; int rep = 1; };

__global__ void gnet_pad_channels_kernel(const float* __restrict__ src, int lds_, int C, float* __restrict__ dst, int Cp, long npix);
inline int nb(long total) { long b = (total + 255) / 256; if (b > 16384) b = 16384; if (b < 1) b = 1; return (int)b; }

}  // namespace gnet

struct GNet {
    typedef gnet::Tn Tn; typedef gnet::Op Op; typedef gnet::GConvW GConvW; typedef gnet::Adapted Adapted;
    int N = 1, H = 0, W = 0;            // network batch / size (after dual-corner padding, if any)
    int Nu = 1, Hu = 0, Wu = 0;         // caller's batch / size
    ptta_hparams hp{};
    std::string err;
    std::vector<void*> allocs;
    bool oom = false;
    std::vector<Tn> T;
    std::map<std::string, int> tid;
    std::vector<Op> ops;
    std::map<std::string, GConvW> convs;
    std::map<std::string, std::pair<float*, float*>> frozen_bn;     // BatchNorm gamma/beta that are not adapted
    std::vector<Adapted> adapted;
    std::map<std::string, int> aid;
    float* gall = nullptr; long gall_n = 0;
    float* w3_tmp = nullptr;
    float *hyper = nullptr, *loss_ws = nullptr, *loss_info = nullptr, *validity_tmp = nullptr;
    int* step_dev = nullptr;
    PttaAdamEntry* adam_tab = nullptr; unsigned* adam_ticket = nullptr; bool adam_tab_dirty = true;
    std::vector<PttaAdamEntry> adam_host;      // stays alive: the asynchronous upload reads it
    float *bn_part = nullptr, *bn_bw = nullptr, *wg_part = nullptr;
    float *depth = nullptr, *gdepth = nullptr;     // (Nu,1,Hu,Wu): network output / its gradient
    int t_emb = -1, t_ref = -1;
    int naive = 0;
    // mixed mode (include/ptta.h PTTA_DTYPE_MIXED for the generic engine): fp32 storage everywhere; the matrix-core convolutions of the
    // proxy frames (bit 0) and of the data gradients (bit 1) take one bf16 MFMA per product instead of three (GX3Args::x1_from_B); bit 2: the data
    // gradients with hi activations x (hi + lo) weights, two MFMAs (GX3Args::x1_w2) -- what PTTA_DTYPE_MIXED selects since round 6
    int mixed = 0;
    // bf16x6 forward for the real frames (GX3Args::six_B): the two-way operand split's 2^-17 representation error reaches the depth
    // map as ~2e-5 relative, enough to flip the sign of near-zero gradient entries -- and Adam's first step turns a sign into +-lr
    // (CostDCNet: post-update eval depth 1.6e-3 from the reference with bf16x3).  Per backbone: CostDCNet on, NLSPN off.
    int x6 = 0;
    int norm_on = 0; float norm_div = 1.f, norm_mean[3] = {0, 0, 0}, norm_std[3] = {1, 1, 1};
    bool fwd_valid = false;
    int max_bn_C = 16;
    // stage-2 head trainer (ghead.hip): `train(prepare=True)` of the reference puts every BatchNorm2d of the backbone into eval mode -- while
    // bn_prepare is set, BN ops that are not `tracked` (= the BatchNorm2d's) normalise every pass with their LOADED running statistics
    int bn_prepare = 0;
    PttaStatSync stat_sync;               // SyncBatchNorm exchange (ptta_set_stat_sync); world == 1: off
    void* grad_comm = nullptr;            // RCCL communicator of the gradient all-reduce inside step() (ptta_set_grad_sync_rccl)

    // ---- hipGraph replay of the fused step and of the eval forward (fixed shapes, ~450 / ~200 launches): the caller's frames are staged
    // at fixed addresses, everything else the launches read is handle-owned or bound memory.  The FIRST call of each kind runs eagerly
    // (lazy one-time work must not happen under capture); graphs are dropped whenever a pointer or a baked-in constant changes.
    int use_graph = 1;
    bool eager_step_done = false, eager_eval_done = false;
    hipGraph_t graph[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};          // step: key = (own loss image) | 2 (caller's validity); 4 = eval forward
    hipGraphExec_t gexec[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    hipStream_t cap_stream = nullptr;
    hipEvent_t ev_replay = nullptr;
    float *gi_image = nullptr, *gi_loss = nullptr, *gi_sparse = nullptr, *gi_valid = nullptr;
    void drop_graphs() {
        if (ev_replay) (void)hipEventSynchronize(ev_replay);
        for (int k = 0; k < 5; ++k) {
            if (gexec[k]) { (void)hipGraphExecDestroy(gexec[k]); gexec[k] = nullptr; }
            if (graph[k]) { (void)hipGraphDestroy(graph[k]); graph[k] = nullptr; }
        }
    }
    bool graph_ok() const { return use_graph && !(stat_sync.on() && !stat_sync.comm); }       // a host callback inside the step cannot be captured
    int stage_inputs(const float* image, const float* loss_image, const float* sparse, const float* validity, hipStream_t s);
    int replay(int key, std::function<int(hipStream_t)> body, hipStream_t s);
    int step_body(const float* image, const float* loss_image, const float* sparse, const float* validity, hipStream_t s);

    virtual ~GNet() {
        drop_graphs();
        if (ev_replay) (void)hipEventDestroy(ev_replay);
        if (cap_stream) (void)hipStreamDestroy(cap_stream);
        for (void* p : allocs) if (p) (void)hipFree(p);
    }
    // ---- backbone-specific ------------------------------------------------------------------------------------------
    virtual int forward(const float* image, const float* sparse, bool train, hipStream_t s) = 0;    // fills `depth`
    virtual int backward(hipStream_t s) = 0;                    // consumes gdepth and T[t_ref].g
    virtual long rows() const = 0;
    virtual int emb_dim() const = 0;
    virtual int load_extra(const std::string& name, const float* src, const int64_t* shape, int ndim, hipStream_t s) { (void)src; (void)shape; (void)ndim; (void)s; return fail("unknown state_dict key " + name, -2); }
    // ---- stage-2 head trainer on this engine (ghead.hip; include/ptta.h ptta_head_*) ---------------------------------------
    struct HeadSpec { int x_real = -1, xw_real = 0, x_proxy = -1, xw_proxy = 0, hidden = 0, out = 0; };
    virtual int head_spec(HeadSpec*) { return fail("the stage-2 head trainer is not built for this backbone", -38); }
    // both no-gradient backbone passes up to the heads' input rows, BatchNorm2d from running statistics (bn_prepare is set by the caller)
    virtual int head_features(const float* image, const float* sparse, hipStream_t s) { (void)image; (void)sparse; (void)s; return fail("the stage-2 head trainer is not built for this backbone", -38); }
    struct HeadTgt { std::string name; long n = 0; float* p = nullptr; };
    struct HeadTrain {
        bool built = false, fwd_ok = false, bwd_ok = false;
        int reverse = -1;                       // wiring of the built program (-1: none yet)
        HeadSpec spec;
        std::vector<Op> ops;                    // proj.{0,1,3} on one pass, proj_t.{0,1,3} on the other, pred.{0,1,3} on proj's output
        std::vector<Adapted> adapted;           // the twelve trained tensors, reference order (proj.0.w, proj.0.b, proj.1.w, ... pred.3.b)
        std::map<std::string, int> aid;
        float* gall = nullptr; long gall_n = 0;
        std::vector<HeadTgt> tgt;               // proj_t.{0,1,3}.{weight,bias}: bound, written by the EMA
        PttaAdamEntry *adam_tab = nullptr, *etab = nullptr; unsigned* ticket = nullptr;
        std::vector<PttaAdamEntry> adam_host, etab_host; bool adam_dirty = true, etab_dirty = true; long etab_total = 0;
        float *hyper = nullptr, *tau2 = nullptr, *loss = nullptr, *loss_part = nullptr;
        int* step_dev = nullptr;
        int t_h[3][3] = {{-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};      // [proj, proj_t, pred][hidden, activation, out]
        int t_emb = -1, t_ref = -1;
    } head;
    int head_build(int reverse);
    int head_bind(const char* name, float* p, float* m, float* v);
    int head_set_hparams(float lr, float b1, float b2, float eps, float wd, float tau, int adam_step, hipStream_t s);
    int head_reload(hipStream_t s);
    int head_forward(const float* image, const float* sparse, int reverse, float* emb_out, float* ref_out, hipStream_t s);
    int head_backward(float* loss_out, hipStream_t s);
    int head_adam_step(hipStream_t s);
    int head_get_grad(const char* name, float* dst, int64_t capacity, int* has_grad_host, hipStream_t s);
    int head_sync_packed(bool targets_too, hipStream_t s);
    virtual int debug_extra(const std::string& nm, const float** src, long* n) { (void)src; (void)n; return fail("unknown debug tensor " + nm, -2); }

    int fail(const std::string& m, int code) { err = m; return code; }
    void* dalloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { oom = true; return nullptr; }
        allocs.push_back(p);
        if (hipMemset(p, 0, bytes ? bytes : 16) != hipSuccess) { oom = true; return nullptr; }
        return p;
    }
    float* falloc(size_t n) { return (float*)dalloc(n * sizeof(float)); }

    // items: allocated items (2*per for tensors that hold both passes); per: items of one pass (default N)
    int tensor(const std::string& name, int items, int h, int w, int c, bool need_grad, int per = 0) {
        Tn t; t.name = name; t.items = items; t.per = per > 0 ? per : N; t.H = h; t.W = w; t.C = c; t.ld = c; t.need_grad = need_grad;
        t.p = falloc((size_t)items * h * w * c);
        if (need_grad) t.g = falloc((size_t)t.per * h * w * c);
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
        v.B = (which == gnet::W_BOTH && train) ? 2 * tn.per : tn.per;
        v.p = (grad ? tn.g : tn.p);
        if (which == gnet::W_PROXY && !grad) v.p += (size_t)tn.per * tn.H * tn.W * tn.ld;
        return v;
    }
    static void review(GView& v, int rH, int rW) {              // same memory, images of rH x rW
        if (rH <= 0) return;
        const long pix = (long)v.B * v.H * v.W;
        v.B = (int)(pix / ((long)rH * rW)); v.H = rH; v.W = rW;
    }

    // layers whose real-frame forward keeps the third operand plane (x6): the backbone's list (x6_layers, comma-separated name prefixes,
    // empty = every layer)
    std::string x6_layers;
    bool x6_layer(const std::string& wname) const {
        if (x6_layers.empty()) return true;
        size_t a = 0;
        while (a <= x6_layers.size()) {
            size_t b = x6_layers.find(',', a); if (b == std::string::npos) b = x6_layers.size();
            if (b > a && wname.compare(0, b - a, x6_layers, a, b - a) == 0) return true;
            a = b + 1;
        }
        return false;
    }
    // ---- program construction --------------------------------------------------------------------------------------
    int add_adapted(const std::string& name, long n) {
        Adapted a; a.name = name; a.n = n; a.goff = gall_n; gall_n += n;
        adapted.push_back(a); aid[name] = (int)adapted.size() - 1;
        return (int)adapted.size() - 1;
    }
    Op& conv(const std::string& wname, int x0, int x1, int y, int k, int stride, int transposed, int act, int xw, int yw,
             bool train_only = false, bool bwd = true) {
        using namespace gnet;
        Op o; o.kind = K_CONV; o.wname = wname; o.x[0] = x0; o.x[1] = x1; o.nsrc = x1 >= 0 ? 2 : 1; o.c0[0] = 0; o.c0[1] = x1 >= 0 ? T[x0].C : 0;
        o.y = y; o.k = k; o.stride = stride; o.transposed = transposed; o.act = act; o.xw[0] = o.xw[1] = xw; o.yw = yw;
        o.train_only = train_only; o.bwd = bwd;
        GConvW& cw = convs[wname];
        cw.Ci = T[x0].C + (x1 >= 0 ? T[x1].C : 0); cw.Co = T[y].C; cw.k = k; cw.stride = stride; cw.transposed = transposed;
        cw.C0 = T[x0].C; cw.C1 = x1 >= 0 ? T[x1].C : 0;
        cw.mf = !naive && (stride == 1 && !transposed ? (cw.C0 % 8) == 0 && (cw.C1 % 8) == 0 : (cw.C0 % 16) == 0 && (cw.C1 % 16) == 0);
        cw.mb = !naive && ((cw.Co % 16) == 0 || (stride == 1 && !transposed));
        cw.Co_pad = (cw.Co + 15) / 16 * 16;               // data gradient of a conv with < 16 output channels: gy is zero-padded
        const long pix = (long)T[y].per * T[y].H * T[y].W;
        if (pix > cw.gpad_pix) cw.gpad_pix = pix;
        ops.push_back(o);
        return ops.back();
    }
    // y = act(bn(x)) [+ res, relu]; adapted gamma/beta unless frozen
    Op& bn(const std::string& bname, int x, int y, int res, int act, int w, bool frozen = false, bool train_only = false, bool bwd = true) {
        using namespace gnet;
        Op o; o.kind = K_BN; o.bname = bname; o.x[0] = x; o.y = y; o.res = res; o.act = act; o.xw[0] = w; o.yw = w;
        o.train_only = train_only; o.bwd = bwd;
        const int C = T[x].C;
        if (C > max_bn_C) max_bn_C = C;
        if (frozen) { if (!frozen_bn.count(bname)) frozen_bn[bname] = std::make_pair(falloc(C), falloc(C)); }
        else {      // a backbone may have reserved the entries already (to fix their place in the adapted list)
            o.ad_g = aid.count(bname + ".weight") ? aid[bname + ".weight"] : add_adapted(bname + ".weight", C);
            o.ad_beta = aid.count(bname + ".bias") ? aid[bname + ".bias"] : add_adapted(bname + ".bias", C);
        }
        o.st = falloc((size_t)4 * 2 * C); o.st_eval = falloc((size_t)4 * C);
        o.rm_own = falloc(C); o.rv_own = falloc(C);
        // statistics fused into the producing convolution's epilogue when that is the stride-1 matrix-core kernel
        for (int k = (int)ops.size() - 1; k >= 0; --k) {
            Op& pr = ops[k];
            if (pr.kind != K_CONV || pr.y != x) continue;
            const GConvW& cw = convs[pr.wname];
            if (cw.mf && pr.stride == 1 && !pr.transposed && pr.act == GACT_NONE && pr.yw == w && T[x].ld == T[x].C) {
                o.fused_from = k; pr.stat_to = (int)ops.size();
                o.rH = pr.rH; o.rW = pr.rW;
                GView v; v.B = 2 * T[x].per; v.H = T[x].H; v.W = T[x].W; review(v, o.rH, o.rW);
                o.part = falloc((size_t)2 * ptta_gconv_x3_tiles(v.B, v.H, v.W) * C);
            }
            break;
        }
        ops.push_back(o);
        return ops.back();
    }
    // closures receive the op's index so that they can read ops[i].first_x at run time
    int func(std::function<int(bool, hipStream_t)> f, std::function<int(hipStream_t)> b, int fx0 = -1, int fx1 = -1, bool train_only = false) {
        Op o; o.kind = gnet::K_FUNC; o.ffwd = f; o.fbwd = b; o.bwd = (bool)b; o.train_only = train_only; o.fx[0] = fx0; o.fx[1] = fx1;
        ops.push_back(o);
        return (int)ops.size() - 1;
    }

    // first writer of every gradient buffer overwrites, later ones accumulate; `seeded`: buffers written before the sweep
    void plan_backward(const std::vector<int>& seeded) {
        using namespace gnet;
        std::vector<char> written(T.size(), 0);
        auto first = [&](int t) {
            // slices share their parent's buffer only through disjoint channels: tracking per tensor id is exact
            const bool f = !written[t]; written[t] = 1; return f;
        };
        for (int t : seeded) first(t);
        for (int i = (int)ops.size() - 1; i >= 0; --i) {
            Op& o = ops[i];
            if (!o.bwd) continue;
            if (o.kind == K_CONV) {
                for (int s = 0; s < o.nsrc; ++s) if (T[o.x[s]].need_grad) o.first_x[s] = first(o.x[s]);
            } else if (o.kind == K_BN) {
                o.first_raw = first(o.x[0]);
                if (o.res >= 0) o.first_res = first(o.res);
            } else {
                for (int k = 0; k < 2; ++k) if (o.fx[k] >= 0) o.first_x[k] = first(o.fx[k]);
            }
        }
    }
    void mark_written(int) {}

    // workspace shared by every backbone; call after the program is built
    void alloc_common(long max_wgrad_pixels, int wg_ci, int wg_co) {
        gall = falloc((size_t)(gall_n > 0 ? gall_n : 1));
        hyper = falloc(8); loss_info = falloc(4); w3_tmp = falloc(4);
        loss_ws = falloc((size_t)ptta_loss_ws_floats(Nu, Hu, Wu, rows()));
        step_dev = (int*)dalloc(sizeof(int));
        adam_tab = (PttaAdamEntry*)dalloc(adapted.size() * sizeof(PttaAdamEntry)); adam_ticket = (unsigned*)dalloc(sizeof(unsigned));
        depth = falloc((size_t)Nu * Hu * Wu); gdepth = falloc((size_t)Nu * Hu * Wu); validity_tmp = falloc((size_t)Nu * Hu * Wu);
        bn_part = falloc((size_t)ptta_gbn_part_floats(max_bn_C, 2)); bn_bw = falloc((size_t)3 * max_bn_C);
        { const size_t a = (size_t)ptta_gwgrad_slabs(max_wgrad_pixels) * (9 * wg_ci * wg_co + wg_co),
                       b = (size_t)ptta_gwgrad_mfma_part_floats(max_wgrad_pixels, wg_ci, wg_co); wg_part = falloc(a > b ? a : b); }
        for (auto& kv : convs) {
            GConvW& cw = kv.second;
            const int KK = cw.k * cw.k;
            if (cw.mf) { const size_t n = (size_t)ptta_gfrag_elems(KK, cw.C0, cw.C1, cw.Co); cw.ff_hi = (bf16_t*)dalloc(n * 2); cw.ff_lo = (bf16_t*)dalloc(n * 2);
                         if (x6) cw.ff_l2 = (bf16_t*)dalloc(n * 2); }
            if (cw.mb && !cw.Ci_real && cw.Co_pad != cw.Co) cw.gpad = falloc((size_t)cw.gpad_pix * cw.Co_pad);
            if (cw.mb && !cw.Ci_real) { const size_t n = (size_t)ptta_gfrag_elems(KK, cw.Co, 0, cw.Ci); cw.fb_hi = (bf16_t*)dalloc(n * 2); cw.fb_lo = (bf16_t*)dalloc(n * 2); }
            if (cw.mb && !cw.Ci_real && cw.C1 > 0 && (cw.C0 % 32) != 0) { const size_t n = (size_t)ptta_gfrag_elems(KK, cw.Co, 0, cw.C1); cw.fb1_hi = (bf16_t*)dalloc(n * 2); cw.fb1_lo = (bf16_t*)dalloc(n * 2); }
            const size_t n = (size_t)KK * cw.Ci * cw.Co;
            cw.wf = falloc(n); cw.wb = falloc(n); cw.bias = falloc(cw.Co);
        }
    }

    // ---- weights ---------------------------------------------------------------------------------------------------
    void pack_frags(GConvW& cw, hipStream_t s) {
        if (cw.vcol) {
            // 3x1x1 Conv3d: the matrix-core kernel's VERT form takes the three vertical taps only (canonical taps 1, 4, 7)
            const long ts = (long)cw.Ci * cw.Co;
            if (cw.mf) ptta_gfrag_pack(cw.wf + ts, cw.Co, 3 * ts, 3, cw.C0, cw.C1, 0, cw.C0, cw.Co, cw.ff_hi, cw.ff_lo, s, cw.ff_l2);
            if (cw.mb) ptta_gfrag_pack(cw.wb + ts, cw.Ci, 3 * ts, 3, cw.Co, 0, 0, 0, cw.Ci, cw.fb_hi, cw.fb_lo, s);
            return;
        }
        const int KK = cw.k * cw.k;
        if (cw.mf && cw.Ci_real) ptta_gfrag_pack(cw.wf, cw.Co, (long)cw.Ci_real * cw.Co, KK, cw.Ci_real, 0, 0, 0, cw.Co, cw.ff_hi, cw.ff_lo, s, cw.ff_l2);
        else if (cw.mf) ptta_gfrag_pack(cw.wf, cw.Co, (long)cw.Ci * cw.Co, KK, cw.C0, cw.C1, 0, cw.C0, cw.Co, cw.ff_hi, cw.ff_lo, s, cw.ff_l2);
        if (cw.mb && !cw.Ci_real) ptta_gfrag_pack(cw.wb, cw.Ci, (long)cw.Co * cw.Ci, KK, cw.Co, 0, 0, 0, cw.Ci, cw.fb_hi, cw.fb_lo, s);
        // second source at a channel offset that is not a tile boundary: its own fragment set (columns C0.. of the packed matrix)
        if (cw.fb1_hi) ptta_gfrag_pack(cw.wb + cw.C0, cw.Ci, (long)cw.Co * cw.Ci, KK, cw.Co, 0, 0, 0, cw.C1, cw.fb1_hi, cw.fb1_lo, s);
    }
    // src: Conv2d (Co,Ci,k,k) / ConvTranspose2d (Ci,Co,k,k) / Linear (Co,Ci) weight
    void pack_conv_weight(GConvW& cw, const float* src, hipStream_t s) {
        const int KK = cw.k * cw.k;
        const int Ci = cw.Ci_real ? cw.Ci_real : cw.Ci;
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
        pack_frags(cw, s);
    }
    int load(const char* name_c, const void* tensor_, const int64_t* shape, int ndim, hipStream_t s);
    // adapted convolutions are re-packed from the bound tensors on every forward
    void repack_adapted(hipStream_t s) {
        for (const Op& o : ops)
            if (o.kind == gnet::K_CONV && o.ad_w >= 0) pack_conv_weight(convs[o.wname], adapted[o.ad_w].p, s);
    }

    // ---- execution -------------------------------------------------------------------------------------------------
    const float* bn_gamma(const Op& o) { return o.ad_g >= 0 ? adapted[o.ad_g].p : frozen_bn[o.bname].first; }
    const float* bn_beta(const Op& o) { return o.ad_beta >= 0 ? adapted[o.ad_beta].p : frozen_bn[o.bname].second; }
    int run_conv_fwd(const Op& o, bool train, hipStream_t s);
    int run_bn_fwd(const Op& o, bool train, hipStream_t s);
    int run_conv_bwd(const Op& o, hipStream_t s);
    int run_bn_bwd(const Op& o, hipStream_t s);
    int run_ops_fwd(bool train, hipStream_t s, int limit = -1) {
        int idx = 0;
        for (const Op& o : ops) {
            if (limit >= 0 && idx++ >= limit) break;
            if (o.train_only && !train) continue;
            const int rc = o.kind == gnet::K_CONV ? run_conv_fwd(o, train, s) : (o.kind == gnet::K_BN ? run_bn_fwd(o, train, s) : o.ffwd(train, s));
            if (rc) return rc;
        }
        return 0;
    }
    int run_ops_bwd(hipStream_t s) {
        for (int i = (int)ops.size() - 1; i >= 0; --i) {
            const Op& o = ops[i];
            if (!o.bwd) continue;
            const int rc = o.kind == gnet::K_CONV ? run_conv_bwd(o, s) : (o.kind == gnet::K_BN ? run_bn_bwd(o, s) : o.fbwd(s));
            if (rc) return rc;
        }
        return 0;
    }
    int upload_hparams(hipStream_t s) {
        const float h8[8] = {hp.lr, hp.beta1, hp.beta2, hp.eps, hp.weight_decay, hp.w_sparse_depth, hp.w_smoothness, hp.w_cos};
        if (ptta_launch_set_floats(hyper, h8, 8, s)) return fail("hyper-parameter upload failed", -5);     // by kernel argument: no sync
        return 0;
    }

    // ---- the C-ABI entry points (ptta_api.hip forwards to these) ---------------------------------------------------------
    int set_hparams(const ptta_hparams* h, hipStream_t s) { if (h->max_input_depth != hp.max_input_depth) drop_graphs(); hp = *h; return upload_hparams(s); }
    int set_image_norm(float div, const float* mean, const float* stdv);
    int bind_adapted(const char* name, float* p, float* m, float* v);
    int adapted_count() const { return (int)adapted.size(); }
    int adapted_repeat(int index) const { return (index < 0 || index >= (int)adapted.size()) ? 0 : adapted[index].rep; }
    const char* adapted_name(int index, int64_t* numel) const {
        if (index < 0 || index >= (int)adapted.size()) return nullptr;
        if (numel) *numel = adapted[index].n;
        return adapted[index].name.c_str();
    }
    int set_adam_step(int step, hipStream_t s) { return ptta_launch_set_int(step_dev, step, s) ? fail("set step failed", -5) : 0; }
    int get_adam_step(int* step, hipStream_t s) {
        if (hipMemcpyAsync(step, step_dev, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return fail("memcpy failed", -5);
        return hipStreamSynchronize(s) == hipSuccess ? 0 : fail("sync failed", -5);       // returns a host value: has to wait
    }
    int forward_train(const float* image, const float* sparse, float* depth_out, float* emb, float* ref, hipStream_t s);
    int forward_eval(const float* image, const float* sparse, float* depth_out, hipStream_t s);
    int step(const float* image, const float* loss_image, const float* sparse, const float* validity, float* depth_out, float* loss_info_out, hipStream_t s);
    int get_grad(const char* name, float* dst, int64_t capacity, hipStream_t s);
    int set_grad(const char* name, const float* src, int64_t numel, hipStream_t s);
    int debug_tensor(const char* name, float* dst, int64_t capacity, int64_t* numel, hipStream_t s);
    int loss_forward(const float* loss_image, const float* depth_, const float* sparse, const float* validity, const float* emb, const float* ref,
                     int64_t rows_, float w_sd, float w_sm, float w_cos, float* loss_info_out, hipStream_t s);
    int loss_backward(const float* loss_image, const float* depth_, const float* sparse, const float* validity, const float* emb, const float* ref,
                      int64_t rows_, float* grad_depth_out, float* grad_ref_out, hipStream_t s);
    int backward_from(const float* grad_depth, const float* grad_ref, hipStream_t s);
    int adam_step(hipStream_t s);
    int upload_adam_table(hipStream_t s);
};
