// CostDCNet engine of libptta_hip (BASELINE config 5): the ProxyTTA step of
// CostDCNet._rgbd_meta_contrast (external_src/costdcnet/CostDCNet_adapt.py:207-256 = CD) behind
// CostDCNetModel_Adapt (src/costdcnet_model_adapt.py = AD: dual-corner padding :134-210, adapt_parameters('meta_bn')
// :357-378 -> conv1_rgb_meta + every BatchNorm2d gamma/beta of Encoder2D = 32 tensors / 5,200 values).
//
// Program on the generic layer-graph engine (gnet.h):
//   Encoder2D (models/encoder2d.py:53-102) on cat(image, sparse): conv-BN-ReLU, six ResBlocks (relu(x + relu(bn(conv)))),
//     1x1 conv; BatchNorm2d always normalises with batch statistics (running statistics dropped by 'meta_bn');
//   conv1_rgb_meta Conv2d(16,16,3): the adapted convolution;
//   sparse 3-D encoder (models/encoder3d.py:33-103, MinkowskiEngine): frozen, input independent of the image -> evaluated
//     ONCE per forward for both passes (costdc_kernels.hip), its BatchNorm running statistics updated twice like the
//     reference's two calls;
//   fusion (CD:390-406) -> P3D UNet3D (models/unet3d.py:7-131): 1x3x3 convs over frames x planes images, 3x1x1 convs as
//     vertical 3-tap convs over [plane][y*x] images, BatchNorm3d (tracked) + ELU, MaxPool3d, nearest upsampling + concat
//     as a two-source convolution; both passes go through the WHOLE UNet because the reference updates every
//     BatchNorm3d's running statistics with the proxy pass too;
//   1x1x1 classifier -> per-plane pixel shuffle + softmax + expected plane (CD:408-424) x z_step;
//   heads: proj / pred on the proxy pass's bottleneck rows, proj_t on the real pass's (CD:243-251).
// Backward = the op list in reverse (data gradients through the frozen UNet3D and Encoder2D, BatchNorm2d gamma/beta
// gradients, weight gradient of the meta conv), then Adam on device.
#include "gnet.h"
#include "nlspn.h"
#include "costdc.h"

using namespace gnet;

struct costdc_engine : GNet {
    float max_depth = 8.f, z_step = 8.f / 15.f;
    int pt = 0, pr = 0, dual = 0;
    int h4 = 0, w4 = 0;
    int t_in = -1, t_feat2d = -1, t_vol = -1, t_cost = -1, t_feat = -1, t_rows = -1, t_rows_p = -1;
    int fD = 2, fh = 0, fw = 0;                       // bottleneck volume (planes, height, width)
    float *img_pad = nullptr, *sp_pad = nullptr, *sp_clamp = nullptr, *feat3d = nullptr, *maskw = nullptr, *pred_net = nullptr, *g_net = nullptr;
    // sparse encoder
    CdSparse sp;
    float* sbuf[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    struct SConv { float* w = nullptr; int K = 0, Ci = 0, Co = 0; bool loaded = false; };
    struct SBn { float *g = nullptr, *b = nullptr, *rm = nullptr, *rv = nullptr; long long* nbt = nullptr; int C = 0;
                 int ad_g = -1, ad_b = -1; float *f = nullptr, *y = nullptr, *st = nullptr; };      // sync_adapt: adapted + saved for the backward
    // the adapted set of the reference's DDP run (PTTA_SYNCBN_ADAPT): convert_syncbn() runs before adapt_parameters('meta_bn')
    // (src/tta_main.py:326,339), so every BatchNorm of the model is adapted and has lost its running statistics (AD:364-372)
    int sync_adapt = 0;
    std::map<std::string, std::pair<int, int>> sbn_ad;
    float* sgrad[4] = {nullptr, nullptr, nullptr, nullptr};
    float* sp_bw = nullptr;
    // The sparse encoder needs only the clamped sparse depth: it runs on a stream of its own BESIDE Encoder2D (its launches are small and
    // latency-bound, Encoder2D's are the step's largest) and joins at the fusion.  Not under SyncBatchNorm: the ranks must issue their
    // exchanges in one order.  PTTA_SPARSE_ASYNC=0 keeps it in line (A/B).
    hipStream_t sp_stream = nullptr;
    hipEvent_t ev_sp_fork = nullptr, ev_sp_done = nullptr;
    bool sp_async = true, sp_inflight = false;
    ~costdc_engine() {
        if (sp_stream) { (void)hipStreamSynchronize(sp_stream); (void)hipStreamDestroy(sp_stream); (void)hipEventDestroy(ev_sp_fork); (void)hipEventDestroy(ev_sp_done); }
    }
    std::map<std::string, SConv> sconv;
    std::map<std::string, SBn> sbn;

    long rows() const override { return (long)N * fh * fw; }
    int emb_dim() const override { return 512; }

    // one P3D block (unet3d.py:66-84) on a volume of D planes of h x w: returns the output tensor
    int p3d(const std::string& name, int x0, int x1, int cout, int D, int h, int w) {
        const int per = N * D;
        const int r1 = tensor(name + ".r1", 2 * per, h, w, cout, true, per), a1 = tensor(name + ".a1", 2 * per, h, w, cout, true, per);
        const int r2 = tensor(name + ".r2", 2 * per, h, w, cout, true, per), a2 = tensor(name + ".a2", 2 * per, h, w, cout, true, per);
        conv(name + ".conv1", x0, x1, r1, 3, 1, 0, GACT_NONE, W_BOTH, W_BOTH);
        bn(name + ".bn1", r1, a1, -1, GACT_ELU, W_BOTH, !sync_adapt).tracked = !sync_adapt;
        { Op& c2 = conv(name + ".conv2", a1, -1, r2, 3, 1, 0, GACT_NONE, W_BOTH, W_BOTH); c2.rH = D; c2.rW = h * w; }
        convs[name + ".conv2"].vcol = 1;
        bn(name + ".bn2", r2, a2, -1, GACT_ELU, W_BOTH, !sync_adapt).tracked = !sync_adapt;
        return a2;
    }
    int double_conv(const std::string& pre, int x0, int x1, int mid, int cout, int D, int h, int w) {
        const int m = p3d(pre + ".double_conv.0", x0, x1, mid, D, h, w);
        return p3d(pre + ".double_conv.1", m, -1, cout, D, h, w);
    }

    void build() {
        const int N2 = 2 * N;
        h4 = H / 4; w4 = W / 4;
        // adapt_parameters('meta_bn') order: parameters with 'meta' in the name first (AD:358-364), then BatchNorm2d affine
        // parameters in module order (AD:365-378)
        const int ad_mw = add_adapted("conv1_rgb_meta.weight", 16L * 16 * 9), ad_mb = add_adapted("conv1_rgb_meta.bias", 16);
        t_in = tensor("in2d", N2, H, W, naive ? 4 : 16, false);       // zero-padded to 16 channels for the matrix-core kernel
        const int c1r = tensor("enc2d.c1.r", N2, H, W, 64, true), c1 = tensor("enc2d.c1", N2, H, W, 64, true);
        conv("enc2d.conv1", t_in, -1, c1r, 3, 1, 0, GACT_NONE, W_BOTH, W_BOTH, false, false);
        if (!naive) convs["enc2d.conv1"].Ci_real = 4;
        bn("enc2d.norm1", c1r, c1, -1, GACT_RELU, W_BOTH);
        int cur = c1, hh = H, ww = W;
        const int planes[3] = {64, 96, 128}, strides[3] = {1, 2, 2};
        for (int li = 0; li < 3; ++li)
            for (int b = 0; b < 2; ++b) {
                const int st = b == 0 ? strides[li] : 1, C = planes[li], ho = hh / st, wo = ww / st;
                char pre_[64]; snprintf(pre_, sizeof(pre_), "enc2d.layer%d.%d", li + 1, b);
                const std::string P(pre_);
                const int r1 = tensor(P + ".r1", N2, ho, wo, C, true), a1 = tensor(P + ".a1", N2, ho, wo, C, true);
                const int r2 = tensor(P + ".r2", N2, ho, wo, C, true), out = tensor(P + ".out", N2, ho, wo, C, true);
                conv(P + ".conv1", cur, -1, r1, 3, st, 0, GACT_NONE, W_BOTH, W_BOTH);
                bn(P + ".norm1", r1, a1, -1, GACT_RELU, W_BOTH);
                conv(P + ".conv2", a1, -1, r2, 3, 1, 0, GACT_NONE, W_BOTH, W_BOTH);
                if (st != 1) {
                    const int rd = tensor(P + ".rd", N2, ho, wo, C, true), d = tensor(P + ".d", N2, ho, wo, C, true);
                    // module (= adapted-parameter) order inside the block: norm1, norm2, norm3 (encoder2d.py:33-37)
                    bn(P + ".norm2", r2, out, d, GACT_RELU, W_BOTH).act_first = true;
                    Op bn2 = ops.back(); ops.pop_back();
                    conv(P + ".downsample.0", cur, -1, rd, 1, st, 0, GACT_NONE, W_BOTH, W_BOTH);
                    bn(P + ".norm3", rd, d, -1, GACT_NONE, W_BOTH);
                    ops.push_back(bn2);
                    if (bn2.fused_from >= 0) ops[bn2.fused_from].stat_to = (int)ops.size() - 1;
                } else {
                    bn(P + ".norm2", r2, out, cur, GACT_RELU, W_BOTH).act_first = true;
                }
                cur = out; hh = ho; ww = wo;
            }
        const int f16 = tensor("enc2d.out", N2, h4, w4, 16, true);
        conv("enc2d.conv2", cur, -1, f16, 1, 1, 0, GACT_NONE, W_BOTH, W_BOTH);
        t_feat2d = tensor("feat2d", N2, h4, w4, 16, true);
        { Op& m = conv("conv1_rgb_meta", f16, -1, t_feat2d, 3, 1, 0, GACT_NONE, W_BOTH, W_BOTH); m.ad_w = ad_mw; m.ad_b = ad_mb; }
        if (sync_adapt) {
            // convert_sync_batchnorm makes one new module per visited attribute: ResBlock.norm3 and its alias downsample[1] become two
            // SyncBatchNorm modules sharing one Parameter, which adapt_parameters then lists twice -> two Adam updates per step
            for (const char* b : {"enc2d.layer2.0.norm3", "enc2d.layer3.0.norm3"})
                for (const char* l : {".weight", ".bias"}) adapted[aid[std::string(b) + l]].rep = 2;
            // module order: enc2d, enc3d (the BatchNorm1d inside each MinkowskiBatchNorm is called `bn`), unet3d, proj, proj_t, pred
            const char* sn[9] = {"bn0", "block1.0.norm1", "block1.0.norm2", "block2.0.norm1", "block2.0.norm2", "block2.0.downsample.1",
                                 "block3.0.norm1", "block3.0.norm2", "block3.0.downsample.1"};
            const int sc_[9] = {32, 32, 32, 48, 48, 48, 64, 64, 64};
            for (int k = 0; k < 9; ++k) {
                const std::string b = std::string("enc3d.") + sn[k];
                const int ig = add_adapted(b + ".bn.weight", sc_[k]), ib = add_adapted(b + ".bn.bias", sc_[k]);
                sbn_ad[b] = std::make_pair(ig, ib);
            }
        }
        // fusion (CD:390-406); the sparse encoder runs inside its forward closure
        t_vol = tensor("vol", N2 * 16, h4, w4, 32, true, N * 16);
        func([this](bool train, hipStream_t s) { return fusion_fwd(train, s); },
             [this](hipStream_t s) {
                 if (cd_launch_fusion_bwd(T[t_vol].g, maskw, T[t_feat2d].g, N, h4, w4, s)) return fail("fusion gradient failed", -5);
                 return sync_adapt ? sparse_backward(s) : 0;
             },
             t_feat2d);
        // UNet3D (unet3d.py:7-47), f_maps = [32, 48, 64, 80]
        const int f[4] = {32, 48, 64, 80};
        int D[4], hs[4], ws[4];
        D[0] = 16; hs[0] = h4; ws[0] = w4;
        for (int i = 1; i < 4; ++i) { D[i] = D[i - 1] / 2; hs[i] = hs[i - 1] / 2; ws[i] = ws[i - 1] / 2; }
        int xl[4];
        xl[0] = double_conv("unet3d.inc", t_vol, -1, f[0], f[0], D[0], hs[0], ws[0]);
        for (int i = 1; i < 4; ++i) {           // Down: MaxPool3d(2) + DoubleConv(in, out, mid = in)
            const int src = xl[i - 1], C = f[i - 1], per = N * D[i];
            char nm[64]; snprintf(nm, sizeof(nm), "unet3d.down%d", i);
            const int pooled = tensor(std::string(nm) + ".pool", 2 * per, hs[i], ws[i], C, true, per);
            const int Hi = hs[i - 1], Wi = ws[i - 1];
            const int oi = func(nullptr, nullptr, src);
            ops[oi].ffwd = [this, src, pooled, Hi, Wi, C](bool train, hipStream_t s) {
                const long items = (long)(train ? 2 : 1) * T[pooled].per;
                return cd_launch_pool_fwd(T[src].p, T[pooled].p, items, Hi, Wi, C, s) ? fail("max-pool failed", -5) : 0;
            };
            ops[oi].fbwd = [this, src, pooled, Hi, Wi, C, oi](hipStream_t s) {
                return cd_launch_pool_bwd(T[src].p, T[pooled].g, T[src].g, (long)T[src].per, Hi, Wi, C, ops[oi].first_x[0] ? 0 : 1, s) ? fail("max-pool gradient failed", -5) : 0;
            };
            ops[oi].bwd = true;
            xl[i] = double_conv(std::string(nm) + ".maxpool_conv.1", pooled, -1, f[i - 1], f[i], D[i], hs[i], ws[i]);
        }
        t_feat = xl[3]; fD = D[3]; fh = hs[3]; fw = ws[3];
        int x = xl[3], xi = 3;
        for (int i = 2; i >= 0; --i) {          // Up: nearest interpolate to the skip's size, cat([skip, up]), DoubleConv(in, out, mid = out)
            const int skip = xl[i], C = T[x].C, per = N * D[i];
            char nm[64]; snprintf(nm, sizeof(nm), "unet3d.up%d", 4 - i);
            const int up = tensor(std::string(nm) + ".up", 2 * per, hs[i], ws[i], C, true, per);
            const int Di = D[xi], Hi = hs[xi], Wi = ws[xi], Do = D[i], Ho = hs[i], Wo = ws[i], xs = x;
            const int oi = func(nullptr, nullptr, xs);
            ops[oi].ffwd = [this, xs, up, Di, Hi, Wi, Do, Ho, Wo, C](bool train, hipStream_t s) {
                return cd_launch_up_fwd(T[xs].p, T[up].p, (long)(train ? 2 : 1) * N, Di, Hi, Wi, Do, Ho, Wo, C, s) ? fail("upsampling failed", -5) : 0;
            };
            ops[oi].fbwd = [this, xs, up, Di, Hi, Wi, Do, Ho, Wo, C, oi](hipStream_t s) {
                return cd_launch_up_bwd(T[up].g, T[xs].g, (long)N, Di, Hi, Wi, Do, Ho, Wo, C, ops[oi].first_x[0] ? 0 : 1, s) ? fail("upsampling gradient failed", -5) : 0;
            };
            ops[oi].bwd = true;
            x = double_conv(std::string(nm) + ".conv", skip, up, f[i], f[i], D[i], hs[i], ws[i]);
            xi = i;
        }
        t_cost = tensor("cost", N * 16, h4, w4, 16, true, N * 16);
        conv("unet3d.classif0", x, -1, t_cost, 1, 1, 0, GACT_NONE, W_GRAD, W_GRAD);
        idx_regress = (int)ops.size();
        func([this](bool, hipStream_t s) { return regress_fwd(s); }, [this](hipStream_t s) { return regress_bwd(s); }, t_cost);
        // heads (CD:243-251, :463-465): emb = pred(proj(rows of the proxy pass)), ref = proj_t(rows of the real pass)
        t_rows = tensor("rows", N, fh, fw, 80 * fD, true);
        t_rows_p = tensor("rows_proxy", N, fh, fw, 80 * fD, false);
        {
            const int oi = func(nullptr, nullptr, t_feat, -1, true);
            idx_rows = oi;
            ops[oi].ffwd = [this](bool, hipStream_t s) {
                const float* fz = T[t_feat].p + (size_t)T[t_feat].per * fh * fw * 80;
                if (cd_launch_rows_fwd(T[t_feat].p, T[t_rows].p, N, fD, fh, fw, 80, s) || cd_launch_rows_fwd(fz, T[t_rows_p].p, N, fD, fh, fw, 80, s))
                    return fail("head rows failed", -5);
                return 0;
            };
            ops[oi].fbwd = [this, oi](hipStream_t s) {
                return cd_launch_rows_bwd(T[t_rows].g, T[t_feat].g, N, fD, fh, fw, 80, ops[oi].first_x[0] ? 0 : 1, s) ? fail("head rows gradient failed", -5) : 0;
            };
            ops[oi].bwd = true;
        }
        auto mlp = [&](const char* name, int xin, bool bwd) {
            const int r = tensor(std::string(name) + ".h", N, fh, fw, 512, bwd), a = tensor(std::string(name) + ".a", N, fh, fw, 512, bwd);
            const int o = tensor(std::string(name) + ".out", N, fh, fw, 512, bwd);
            conv(std::string(name) + ".0", xin, -1, r, 1, 1, 0, GACT_NONE, W_GRAD, W_GRAD, true, bwd);
            bn(std::string(name) + ".1", r, a, -1, GACT_RELU, W_GRAD, !sync_adapt, true, bwd).tracked = !sync_adapt;
            conv(std::string(name) + ".3", a, -1, o, 1, 1, 0, GACT_NONE, W_GRAD, W_GRAD, true, bwd);
            return o;
        };
        if (sync_adapt)            // module order proj, proj_t, pred; proj / pred only feed the detached embedding: never given a gradient
            for (const char* b : {"proj.1", "proj_t.1", "pred.1"})
                for (const char* l : {".weight", ".bias"}) {
                    const int i = add_adapted(std::string(b) + l, 512);
                    if (strcmp(b, "proj_t.1") != 0) adapted[i].rep = 0;
                }
        const int pz = mlp("proj", t_rows_p, false);
        t_emb = mlp("pred", pz, false); tid["emb"] = t_emb;
        t_ref = mlp("proj_t", t_rows, true); tid["ref"] = t_ref;
        plan_backward({t_ref});
        // ---- remaining workspace ----
        const long P = (long)H * W;
        if (dual) { img_pad = falloc((size_t)N * 3 * P); sp_pad = falloc((size_t)N * P); }
        sp_clamp = falloc((size_t)N * P); pred_net = falloc((size_t)N * P); g_net = falloc((size_t)N * P);
        feat3d = falloc((size_t)N * 16 * h4 * w4 * 16); maskw = falloc((size_t)N * 16 * h4 * w4);
        build_sparse();
        alloc_common((long)N * h4 * w4, 16, 16);
    }

    // ---- sparse 3-D encoder (models/encoder3d.py) ---------------------------------------------------------------------------
    void build_sparse() {
        sp.N = N; sp.H = H; sp.W = W;
        const long cap = (long)N * H * W;                    // at most one voxel per pixel at level 0
        for (int l = 0; l < 3; ++l) {
            sp.vol[l] = (int*)dalloc((size_t)N * 16 * (H >> l) * (W >> l) * sizeof(int));
            sp.coords[l] = (int4*)dalloc((size_t)cap * sizeof(int4));
        }
        sp.cnt = (int*)dalloc(4 * sizeof(int));
        sp.rowcnt = (int*)dalloc((size_t)N * 16 * H * sizeof(int)); sp.rowoff = (int*)dalloc((size_t)N * 16 * H * sizeof(int));
        sp.feat_in = falloc((size_t)cap);
        sp.bn_part = falloc((size_t)256 * 2 * 64); sp.bn_st = falloc(4 * 64);
        for (int k = 0; k < 5; ++k) sbuf[k] = falloc((size_t)cap * 64);
        auto sc = [&](const std::string& n, int K, int Ci, int Co) { SConv c; c.K = K; c.Ci = Ci; c.Co = Co; c.w = falloc((size_t)K * Ci * Co); sconv[n] = c; };
        auto sb = [&](const std::string& n, int C) {
            SBn b; b.C = C; b.g = falloc(C); b.b = falloc(C);
            if (sync_adapt) {                    // adapted, and everything the backward needs is kept: input, output, [scale, shift, mean, inv]
                b.ad_g = sbn_ad[n].first; b.ad_b = sbn_ad[n].second;
                b.f = falloc((size_t)cap * C); b.y = falloc((size_t)cap * C); b.st = falloc((size_t)4 * C);
            }
            sbn[n] = b;
        };
        if (sync_adapt) { for (int k = 0; k < 4; ++k) sgrad[k] = falloc((size_t)cap * 64); sp_bw = falloc(3 * 64); }
        sc("enc3d.conv1", 27, 1, 32); sb("enc3d.bn0", 32);
        int inpl = 32; const int pl[3] = {32, 48, 64};
        for (int b = 0; b < 3; ++b) {
            const std::string P = "enc3d.block" + std::to_string(b + 1) + ".0";
            sc(P + ".conv1", 27, inpl, pl[b]); sb(P + ".norm1", pl[b]);
            sc(P + ".conv2", 27, pl[b], pl[b]); sb(P + ".norm2", pl[b]);
            if (b > 0) { sc(P + ".downsample.0", 1, inpl, pl[b]); sb(P + ".downsample.1", pl[b]); }
            inpl = pl[b];
        }
        sc("enc3d.conv2", 1, 64, 16);
    }
    int sconv_run(const std::string& n, const float* fin, int lin, int lout, float* fout, hipStream_t s) {
        const SConv& c = sconv[n];
        if (!c.loaded) return fail("weights of " + n + " not loaded (ptta_load_weights)", -3);
        return cd_launch_sparse_conv(sp, fin, lin, lout, c.w, c.K == 27 ? 3 : 1, c.Ci, c.Co, fout, s) ? fail("sparse conv " + n + " failed", -5) : 0;
    }
    int sbn_run(const std::string& n, const float* f, const float* res, int level, bool train, int relu, float* out, hipStream_t s) {
        SBn& b = sbn[n];
        if (sync_adapt)      // no running statistics: batch statistics in train AND eval mode (synchronised over the ranks in train mode only)
            return cd_launch_sparse_bn(sp, f, res, level, b.C, adapted[b.ad_g].p, adapted[b.ad_b].p, nullptr, nullptr, nullptr, 1, 1, relu, out, s,
                                       train ? &stat_sync : nullptr, b.st) ? fail("sparse batch-norm " + n + " failed", -5) : 0;
        if (!train && (!b.rm || !b.rv)) return fail("running statistics of " + n + " not loaded", -3);
        // train: the reference evaluates the sparse encoder once per pass on the same input (CD:216, :237): two updates
        return cd_launch_sparse_bn(sp, f, res, level, b.C, b.g, b.b, b.rm, b.rv, b.nbt, train ? 1 : 0, 2, relu, out, s, &stat_sync) ? fail("sparse batch-norm " + n + " failed", -5) : 0;
    }
    // sync_adapt: the same network with every BatchNorm's input and output kept, then its backward (BatchNorm gamma / beta gradients only)
    int sparse_encoder_adapt(bool train, hipStream_t s) {
        NRUN(sconv_run("enc3d.conv1", sp.feat_in, 0, 0, sbn["enc3d.bn0"].f, s));
        NRUN(sbn_run("enc3d.bn0", sbn["enc3d.bn0"].f, nullptr, 0, train, 1, sbn["enc3d.bn0"].y, s));
        const float* x = sbn["enc3d.bn0"].y;
        for (int blk = 1; blk <= 3; ++blk) {
            const std::string P = "enc3d.block" + std::to_string(blk) + ".0";
            const int lin = blk == 1 ? 0 : blk - 2, lout = blk - 1;
            SBn &n1 = sbn[P + ".norm1"], &n2 = sbn[P + ".norm2"];
            NRUN(sconv_run(P + ".conv1", x, lin, lout, n1.f, s));
            NRUN(sbn_run(P + ".norm1", n1.f, nullptr, lout, train, 1, n1.y, s));
            NRUN(sconv_run(P + ".conv2", n1.y, lout, lout, n2.f, s));
            const float* res = x;
            if (blk > 1) {
                SBn& ds = sbn[P + ".downsample.1"];
                NRUN(sconv_run(P + ".downsample.0", x, lin, lout, ds.f, s));
                NRUN(sbn_run(P + ".downsample.1", ds.f, nullptr, lout, train, 0, ds.y, s));
                res = ds.y;
            }
            NRUN(sbn_run(P + ".norm2", n2.f, res, lout, train, 1, n2.y, s));
            x = n2.y;
        }
        NRUN(sconv_run("enc3d.conv2", x, 2, 2, sbuf[0], s));
        if (cd_launch_densify(sp, sbuf[0], 16, feat3d, s)) return fail("densify failed", -5);
        return 0;
    }
    int sconv_bwd(const std::string& n, const float* gy, int lin, int lout, float* gx, int acc, hipStream_t s) {
        const SConv& c = sconv[n];
        return cd_launch_sparse_conv_bwd(sp, gy, lin, lout, c.w, c.K == 27 ? 3 : 1, c.Ci, c.Co, gx, acc, s) ? fail("sparse conv gradient " + n + " failed", -5) : 0;
    }
    int sbn_bwd(const std::string& n, const float* g, int level, int relu, float* gx, float* gres, hipStream_t s) {
        SBn& b = sbn[n];
        return cd_launch_sparse_bn_bwd(sp, b.f, b.y, g, level, b.C, adapted[b.ad_g].p, b.st, relu, gall + adapted[b.ad_g].goff, gall + adapted[b.ad_b].goff,
                                       gx, gres, 0, sp_bw, s, &stat_sync) ? fail("sparse batch-norm gradient " + n + " failed", -5) : 0;
    }
    int sparse_backward(hipStream_t s) {
        float *G0 = sgrad[0], *G1 = sgrad[1], *G2 = sgrad[2], *G3 = sgrad[3];
        // the real pass's half of the fused volume's gradient, channels 16..31 (fusion concatenates [feat2d * mask | feat3d])
        if (cd_launch_densify_bwd(sp, T[t_vol].g, 32, 16, 16, G0, s)) return fail("densify gradient failed", -5);
        NRUN(sconv_bwd("enc3d.conv2", G0, 2, 2, G1, 0, s));                              // G1 = d block3 output
        for (int blk = 3; blk >= 1; --blk) {
            const std::string P = "enc3d.block" + std::to_string(blk) + ".0";
            const int lin = blk == 1 ? 0 : blk - 2, lout = blk - 1;
            NRUN(sbn_bwd(P + ".norm2", G1, lout, 1, G0, G2, s));                         // G0 = d conv2 out, G2 = d residual branch
            NRUN(sconv_bwd(P + ".conv2", G0, lout, lout, G3, 0, s));
            NRUN(sbn_bwd(P + ".norm1", G3, lout, 1, G0, nullptr, s));
            if (blk > 1) {
                NRUN(sconv_bwd(P + ".conv1", G0, lin, lout, G1, 0, s));                  // G1 = d block input
                NRUN(sbn_bwd(P + ".downsample.1", G2, lout, 0, G0, nullptr, s));
                NRUN(sconv_bwd(P + ".downsample.0", G0, lin, lout, G1, 1, s));
            } else {
                NRUN(sconv_bwd(P + ".conv1", G0, lin, lout, G2, 1, s));                  // identity residual: G2 already holds its share
            }
        }
        return sbn_bwd("enc3d.bn0", G2, 0, 1, G0, nullptr, s);
    }
    int sparse_encoder(bool train, hipStream_t s) {
        if (cd_sparse_levels_build(sp, sp_clamp, z_step, s)) return fail("depth2MDP failed", -5);
        if (sync_adapt) return sparse_encoder_adapt(train, s);
        float *a = sbuf[0], *b = sbuf[1], *c = sbuf[2], *d = sbuf[3], *e = sbuf[4];
        NRUN(sconv_run("enc3d.conv1", sp.feat_in, 0, 0, a, s));
        NRUN(sbn_run("enc3d.bn0", a, nullptr, 0, train, 1, b, s));                       // out_p1 = b
        // block1: stride 1, no downsample
        NRUN(sconv_run("enc3d.block1.0.conv1", b, 0, 0, a, s));
        NRUN(sbn_run("enc3d.block1.0.norm1", a, nullptr, 0, train, 1, c, s));
        NRUN(sconv_run("enc3d.block1.0.conv2", c, 0, 0, a, s));
        NRUN(sbn_run("enc3d.block1.0.norm2", a, b, 0, train, 1, d, s));                  // out_p2 = d (level 0)
        float* x = d; float* t0 = a; float* t1 = b; float* t2 = c; float* t3 = e;
        for (int blk = 2; blk <= 3; ++blk) {                                            // stride (1,2,2): level blk-2 -> blk-1
            const std::string P = "enc3d.block" + std::to_string(blk) + ".0";
            const int lin = blk - 2, lout = blk - 1;
            NRUN(sconv_run(P + ".conv1", x, lin, lout, t0, s));
            NRUN(sbn_run(P + ".norm1", t0, nullptr, lout, train, 1, t1, s));
            NRUN(sconv_run(P + ".conv2", t1, lout, lout, t0, s));
            NRUN(sconv_run(P + ".downsample.0", x, lin, lout, t2, s));
            NRUN(sbn_run(P + ".downsample.1", t2, nullptr, lout, train, 0, t3, s));      // residual = t3
            NRUN(sbn_run(P + ".norm2", t0, t3, lout, train, 1, t1, s));                  // relu(bn(conv2) + residual)
            float* nx = t1; t1 = x; x = nx;
        }
        NRUN(sconv_run("enc3d.conv2", x, 2, 2, t0, s));
        if (cd_launch_densify(sp, t0, 16, feat3d, s)) return fail("densify failed", -5);
        return 0;
    }
    int fusion_fwd(bool train, hipStream_t s) {
        if (sp_inflight) { NCHK(hipStreamWaitEvent(s, ev_sp_done, 0)); sp_inflight = false; }
        else { const int rc = sparse_encoder(train, s); if (rc) return rc; }
        return cd_launch_fusion_fwd(T[t_feat2d].p, feat3d, T[t_vol].p, maskw, N, train ? 2 : 1, h4, w4, s) ? fail("fusion failed", -5) : 0;
    }
    int regress_fwd(hipStream_t s) {
        if (cd_launch_regress_fwd(T[t_cost].p, dual ? pred_net : depth, N, h4, w4, z_step, s)) return fail("depth regression failed", -5);
        if (dual && cd_launch_crop_avg(pred_net, depth, Nu, Hu, Wu, H, W, pt, pr, s)) return fail("crop failed", -5);
        return 0;
    }
    int regress_bwd(hipStream_t s) {
        const float* g = gdepth;
        if (dual) { if (cd_launch_scatter_dual_grad(gdepth, g_net, Nu, Hu, Wu, H, W, pt, pr, s)) return fail("pad gradient failed", -5); g = g_net; }
        return cd_launch_regress_bwd(T[t_cost].p, g, T[t_cost].g, N, h4, w4, z_step, s) ? fail("depth regression gradient failed", -5) : 0;
    }

    // ---- weights that are not dense convolutions / affine BatchNorm parameters ---------------------------------------------------
    int load_extra(const std::string& name, const float* src, const int64_t* shape, int ndim, hipStream_t s) override {
        auto ends = [&](const char* suf) { const size_t l = strlen(suf); return name.size() >= l && name.compare(name.size() - l, l, suf) == 0; };
        const size_t dot = name.rfind('.');
        const std::string base = dot == std::string::npos ? name : name.substr(0, dot), leaf = dot == std::string::npos ? "" : name.substr(dot + 1);
        if (name.rfind("enc3d.", 0) == 0) {
            if (leaf == "kernel") {
                auto it = sconv.find(base);
                if (it == sconv.end()) return fail("unknown state_dict key " + name, -2);
                SConv& c = it->second;
                long n = 1; for (int i = 0; i < ndim; ++i) n *= shape[i];
                if (n != (long)c.K * c.Ci * c.Co) return fail("shape mismatch for " + name, -22);
                NCHK(hipMemcpyAsync(c.w, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));       // (K, Cin, Cout) as stored
                c.loaded = true;
                return 0;
            }
            // MinkowskiBatchNorm wraps an nn.BatchNorm1d called `bn`: enc3d.<...>.bn.{weight,bias,running_*,num_batches_tracked}
            const std::string owner = base.size() > 3 && base.compare(base.size() - 3, 3, ".bn") == 0 ? base.substr(0, base.size() - 3) : base;
            auto it = sbn.find(owner);
            if (it == sbn.end()) return fail("unknown state_dict key " + name, -2);
            SBn& b = it->second;
            if (leaf == "weight") { NCHK(hipMemcpyAsync(b.g, src, (size_t)b.C * sizeof(float), hipMemcpyDeviceToDevice, s)); return 0; }
            if (leaf == "bias") { NCHK(hipMemcpyAsync(b.b, src, (size_t)b.C * sizeof(float), hipMemcpyDeviceToDevice, s)); return 0; }
            if (sync_adapt && leaf != "weight" && leaf != "bias") return 0;         // running statistics dropped with the SyncBatchNorm conversion
            if (leaf == "running_mean") { b.rm = (float*)src; return 0; }           // bound, updated in place
            if (leaf == "running_var") { b.rv = (float*)src; return 0; }
            if (leaf == "num_batches_tracked") { b.nbt = (long long*)src; return 0; }
            return fail("unknown state_dict key " + name, -2);
        }
        // ResBlock.norm3 is registered a second time inside `downsample` (same tensors): the norm3 keys are authoritative
        if (name.rfind("enc2d.", 0) == 0 && name.find(".downsample.1.") != std::string::npos) return 0;
        // BatchNorm2d running statistics are dropped by adapt_parameters('meta_bn') (AD:370-372)
        if (name.rfind("enc2d.", 0) == 0 && (ends("running_mean") || ends("running_var") || ends("num_batches_tracked"))) return 0;
        if (ends("running_mean") || ends("running_var") || ends("num_batches_tracked")) {
            if (sync_adapt) return 0;
            for (Op& o : ops)
                if (o.kind == K_BN && o.tracked && o.bname == base) {
                    if (leaf == "running_mean") o.rm = (float*)src; else if (leaf == "running_var") o.rv = (float*)src; else o.nbt = (long long*)src;
                    return 0;
                }
        }
        return fail("unknown state_dict key " + name, -2);
    }

    int idx_regress = -1, idx_rows = -1;
    // stage-2 head trainer (ghead.hip): rows = the UNet3D bottleneck of the real pass / of the zero-image pass (CD:268-290); MLP(160, 512, 512)
    int head_spec(HeadSpec* hs) override {
        if (sync_adapt) return fail("the stage-2 head trainer runs on a handle without PTTA_SYNCBN_ADAPT", -38);
        if (dual) return fail("stage 2 runs on sizes divisible by 16", -38);
        hs->x_real = t_rows; hs->xw_real = W_GRAD; hs->x_proxy = t_rows_p; hs->xw_proxy = W_GRAD; hs->hidden = 512; hs->out = 512;
        return 0;
    }
    // both passes through Encoder2D (BatchNorm2d from running statistics: bn_prepare), the sparse encoder and the WHOLE UNet3D (train mode: its
    // BatchNorm3d running statistics move as in the reference, which evaluates the full unet3d for the feature, CD:275,288), then the rows
    int head_features(const float* image, const float* sparse, hipStream_t s) override {
        NRUN(forward_ops(image, sparse, true, s, idx_regress));
        fwd_valid = false;
        return ops[idx_rows].ffwd(true, s);
    }
    int forward(const float* image, const float* sparse, bool train, hipStream_t s) override { return forward_ops(image, sparse, train, s, -1); }
    int forward_ops(const float* image, const float* sparse, bool train, hipStream_t s, int limit) {
        for (auto& ad : adapted) if (!ad.p) return fail("adapted parameter " + ad.name + " not bound (ptta_bind_adapted)", -3);
        const float* img = image; const float* spp = sparse;
        if (dual) {
            // the fused image normalisation happens while padding (in-frame pixels only: the padding stays zero, as in the reference)
            if (cd_launch_pad_dual(image, img_pad, Nu, 3, Hu, Wu, H, W, pt, pr, s, norm_on, norm_div, norm_mean, norm_std) ||
                cd_launch_pad_dual(sparse, sp_pad, Nu, 1, Hu, Wu, H, W, pt, pr, s))
                return fail("padding failed", -5);
            img = img_pad; spp = sp_pad;
        }
        // clamp (src/external_model_adapt.py:108) and cat([image | zeros, sparse]) staging; the reference normalises the image
        // before it pads, so padded pixels must stay zero: normalisation is applied to the caller's frame region only
        if (cd_launch_clamp(spp, sp_clamp, (long)N * H * W, hp.max_input_depth, s)) return fail("clamp failed", -5);
        if (cd_launch_stage(img, sp_clamp, T[t_in].p, N, train ? 2 : 1, H, W, T[t_in].C, dual ? 0 : norm_on, norm_div, norm_mean, norm_std, s)) return fail("input staging failed", -5);
        sp_inflight = false;
        if (sp_async && !stat_sync.on()) {
            if (!sp_stream) {
                NCHK(hipStreamCreateWithFlags(&sp_stream, hipStreamNonBlocking));
                NCHK(hipEventCreateWithFlags(&ev_sp_fork, hipEventDisableTiming));
                NCHK(hipEventCreateWithFlags(&ev_sp_done, hipEventDisableTiming));
            }
            NCHK(hipEventRecord(ev_sp_fork, s));                      // the clamped sparse depth is ready
            NCHK(hipStreamWaitEvent(sp_stream, ev_sp_fork, 0));
            const int rcs = sparse_encoder(train, sp_stream);
            if (rcs) return rcs;
            NCHK(hipEventRecord(ev_sp_done, sp_stream));
            sp_inflight = true;
        }
        repack_adapted(s);
        const int rc = run_ops_fwd(train, s, limit);
        if (rc) return rc;
        fwd_valid = train;
        return 0;
    }
    int backward(hipStream_t s) override {
        if (!fwd_valid) return fail("backward without a training forward", -3);
        return run_ops_bwd(s);
    }
};

GNet* costdc_create(int n, int h, int w, const ptta_hparams* hp, float max_depth, int flags, int* rc) {
    *rc = 0;
    if (n < 1 || h < 32 || w < 32 || !hp || !(max_depth > 0.f)) { *rc = -22; return nullptr; }
    costdc_engine* e = new costdc_engine();
    e->Nu = n; e->Hu = h; e->Wu = w;
    e->pt = (16 - h % 16) % 16; e->pr = (16 - w % 16) % 16; e->dual = (e->pt || e->pr) ? 1 : 0;
    e->N = e->dual ? 2 * n : n; e->H = h + e->pt; e->W = w + e->pr;
    if ((e->H / 4) < 8 || (e->W / 4) < 8) { delete e; *rc = -22; return nullptr; }      // three 2x poolings of the 1/4-resolution volume
    e->hp = *hp; e->max_depth = max_depth; e->z_step = (float)((double)max_depth / 15.0);
    const PttaCreateEnv env = ptta_create_env();
    e->naive = env.naive;
    e->x6 = e->naive ? 0 : 1;                // the third operand plane (parity of the post-update depth: DESIGN.md)
    // ... on the layers that decide it (measured layer by layer, tools/costdc_report.py): Encoder2D, the meta convolution
    // and the first two levels of the UNet3D give the post-update depth of bf16x6 everywhere (5.5e-6 / 6.0e-4 at 320x400 / 480x640); any
    // smaller set leaves it at 0.8 - 1.3e-3
    // (the DDP adapted set adapts the UNet3D's own BatchNorm too: every layer keeps the third plane there)
    e->x6_layers = (flags & 1) ? "" : "enc2d,conv1_rgb_meta,unet3d.inc,unet3d.down1";
    // hipGraph replay of the step / eval forward: built and bit-identical (tests), but measured 0.3 - 2 % SLOWER than kernel-by-kernel
    // launches on this engine (the host keeps ahead of the GPU either way: DESIGN_LOG.md section 9) -> opt-in: ptta_set_option(h, "graph", 1) / PTTA_GRAPH=1
    e->use_graph = env.graph == 1 ? 1 : 0;
    e->sync_adapt = (flags & 1) ? 1 : 0;
    e->build();
    if (e->oom || !e->step_dev) { delete e; *rc = -12; return nullptr; }
    if (e->upload_hparams(nullptr)) { delete e; *rc = -5; return nullptr; }
    return e;
}
