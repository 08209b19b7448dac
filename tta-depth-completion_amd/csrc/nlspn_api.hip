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
#include "gnet.h"
#include "nlspn.h"

using namespace gnet;

namespace {

const int PROP_TIME = 18;

__global__ void clamp_dup_kernel(const float* __restrict__ src, float* __restrict__ dst, long per, int copies, float maxd) {
    const long total = per * copies;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float v = src[i % per];
        if (maxd >= 0.f) v = fminf(fmaxf(v, 0.f), maxd);
        dst[i] = v;
    }
}
// `_concat` (nlspnmodel_adapt.py:474-490): dst[b][y][x][c] = src[b][y][x][c] for y < Hd, x < Wd (trailing rows / columns dropped)
__global__ void crop_fwd_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int Hs, int Ws, int Hd, int Wd, int C) {
    const long total = (long)B * Hd * Wd * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C); long t = idx / C;
        const int x = (int)(t % Wd); t /= Wd;
        const int y = (int)(t % Hd); const int b = (int)(t / Hd);
        dst[idx] = src[(((long)b * Hs + y) * Ws + x) * C + c];
    }
}
// its gradient: the cropped border received nothing
__global__ void crop_bwd_kernel(const float* __restrict__ gdst, float* __restrict__ gsrc, int B, int Hs, int Ws, int Hd, int Wd, int C, int accumulate) {
    const long total = (long)B * Hs * Ws * C;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C); long t = idx / C;
        const int x = (int)(t % Ws); t /= Ws;
        const int y = (int)(t % Hs); const int b = (int)(t / Hs);
        const float g = (y < Hd && x < Wd) ? gdst[(((long)b * Hd + y) * Wd + x) * C + c] : 0.f;
        gsrc[idx] = accumulate ? gsrc[idx] + g : g;
    }
}
__global__ void relu_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = fmaxf(src[i], 0.f);
}
// gradient of clamp(min=0): passes where y >= 0
__global__ void clamp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = y[i] >= 0.f ? g[i] : 0.f;
}

}  // namespace

struct nlspn_engine : GNet {
    float* S = nullptr;
    // inputs / propagation
    int t_sd16 = -1;
    int t_img = -1, t_sd = -1, t_pred = -1, t_oa = -1, t_conf = -1, t_fe6 = -1;
    float *off9 = nullptr, *aff9 = nullptr, *goff9 = nullptr, *gaff9 = nullptr, *feats = nullptr, *gy = nullptr, *gping = nullptr;
    int legacy = 0, heads_adapted = 0;
    int enc_ops_end = 0;               // ops [0, enc_ops_end) = conv1 .. conv6: what the stage-2 head trainer runs of the backbone

    static int half(int v) { return (v + 1) / 2; }                 // 3x3 stride 2 padding 1: ceil
    long rows() const override { return (long)N * half(half(half(half(H)))) * half(half(half(half(W)))); }
    int emb_dim() const override { return 1024; }

    void build() {
        // encoder sizes: every stride-2 3x3 convolution (padding 1) halves with ceil; fe2 is at full resolution (layer1 has
        // stride 1), fe3 1/2, fe4 1/4, fe5 1/8, fe6 1/16
        const int H2 = half(H), W2 = half(W), H4 = half(H2), W4 = half(W2), H8 = half(H4), W8 = half(W4), H16 = half(H8), W16 = half(W8);
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
                const int ho = stride == 2 ? half(hh) : hh, wo = stride == 2 ? half(ww) : ww, C = planes[st];
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
        enc_ops_end = (int)ops.size();
        // shared decoder (:413-424): transposed convs over [decoder | encoder skip]
        // a transposed convolution doubles its input; when an encoder map has an odd size the decoder map is one row / column
        // larger and `_concat` crops it AFTER BatchNorm saw the whole map (nlspnmodel_adapt.py:474-490): the layer is built at
        // its full size and a crop op feeds the next concatenation (its backward zero-fills the cropped border)
        auto dec = [&](const char* name, int a, int b, int hin, int win, int h, int w, int cout) {
            const int hf = 2 * hin, wf = 2 * win;
            const int r = tensor(std::string(name) + ".r", N, hf, wf, cout, true), o = tensor(std::string(name) + ".o", N, hf, wf, cout, true);
            conv(std::string(name) + ".0", a, b, r, 3, 2, 1, GACT_NONE, W_GRAD, W_GRAD);
            bn(std::string(name) + ".1", r, o, -1, GACT_LRELU, W_GRAD);
            if (hf == h && wf == w) return o;
            const int oc = tensor(std::string(name) + ".crop", N, h, w, cout, true);
            const int oi = func(nullptr, nullptr, o);
            ops[oi].ffwd = [this, o, oc](bool, hipStream_t s) {
                hipLaunchKernelGGL(crop_fwd_kernel, dim3(nb((long)T[oc].per * T[oc].H * T[oc].W * T[oc].C)), dim3(256), 0, s, (const float*)T[o].p, T[oc].p,
                                   T[oc].per, T[o].H, T[o].W, T[oc].H, T[oc].W, T[oc].C);
                return hipGetLastError() == hipSuccess ? 0 : fail("crop failed", -5);
            };
            ops[oi].fbwd = [this, o, oc, oi](hipStream_t s) {
                hipLaunchKernelGGL(crop_bwd_kernel, dim3(nb((long)T[o].per * T[o].H * T[o].W * T[o].C)), dim3(256), 0, s, (const float*)T[oc].g, T[o].g,
                                   T[o].per, T[o].H, T[o].W, T[oc].H, T[oc].W, T[o].C, ops[oi].first_x[0] ? 0 : 1);
                return hipGetLastError() == hipSuccess ? 0 : fail("crop gradient failed", -5);
            };
            ops[oi].bwd = true;
            return oc;
        };
        const int fd5 = dec("dec5", fe[6], -1, H16, W16, H8, W8, 256);
        const int fd4 = dec("dec4", fd5, fe[5], H8, W8, H4, W4, 128);
        const int fd3 = dec("dec3", fd4, fe[4], H4, W4, H2, W2, 64);
        const int fd2 = dec("dec2", fd3, fe[3], H2, W2, H, W, 64);
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
            // BatchNorm1d of the heads: train mode on the TTA path, running statistics kept (meta_bn only drops BatchNorm2d's)
            bn(std::string(name) + ".1", r, a, -1, GACT_RELU, W_GRAD, !heads_adapted, true, bwd).tracked = true;
            conv(std::string(name) + ".3", a, -1, o, 1, 1, 0, GACT_NONE, W_GRAD, W_GRAD, true, bwd);
            return o;
        };
        // module order proj, proj_t, pred (:1340-1342): the order of their BatchNorm parameters in the 94-tensor adapted list
        const int pz = mlp("proj", fe[6], W_PROXY, 512, 1024, false);
        t_ref = mlp("proj_t", fe[6], W_GRAD, 512, 1024, true);
        tid["ref"] = t_ref;
        t_emb = mlp("pred", pz, W_GRAD, 1024, 1024, false);
        tid["emb"] = t_emb;

        plan_backward({t_pred, t_oa, t_conf, t_ref});           // written by the propagation / loss gradients
        // ---- remaining workspace ----
        const long P = (long)H * W;
        off9 = falloc((size_t)N * P * 18); aff9 = falloc((size_t)N * P * 9);
        goff9 = falloc((size_t)N * P * 18); gaff9 = falloc((size_t)N * P * 9);
        feats = falloc((size_t)(PROP_TIME + 1) * N * P);
        gy = falloc((size_t)N * P); gping = falloc((size_t)N * P);
        S = falloc(1);
        alloc_common((long)N * P, 48, 48);
    }

    // ---- weights that are not convolutions / BatchNorm affine ----------------------------------------------------------
    int load_extra(const std::string& name, const float* src, const int64_t* shape, int ndim, hipStream_t s) override {
        (void)shape; (void)ndim;
        auto ends = [&](const char* suf) { const size_t l = strlen(suf); return name.size() >= l && name.compare(name.size() - l, l, suf) == 0; };
        if (ends("num_batches_tracked") || ends("running_mean") || ends("running_var")) {
            const size_t dot = name.rfind('.');
            const std::string base = name.substr(0, dot), leaf = name.substr(dot + 1);
            for (Op& o : ops)            // the heads' BatchNorm1d buffers are BOUND and updated in place by every training forward
                if (o.kind == K_BN && o.tracked && o.bname == base) {
                    if (leaf == "running_mean") o.rm = (float*)src; else if (leaf == "running_var") o.rv = (float*)src; else o.nbt = (long long*)src;
                    return 0;
                }
            return 0;                    // BatchNorm2d: dropped by adapt_parameters('meta_bn')
        }
        if (name == "prop_layer.aff_scale_const") { NCHK(hipMemcpyAsync(S, src, sizeof(float), hipMemcpyDeviceToDevice, s)); return 0; }
        if (name == "prop_layer.w" || name == "prop_layer.b" || name == "prop_layer.w_conf") return 0;   // constants ones / zero (nlspnmodel_adapt.py:239-247)
        return fail("unknown state_dict key " + name, -2);
    }
    int debug_extra(const std::string& nm, const float** src, long* n) override {
        const long NP = (long)N * H * W;
        if (nm == "off9") { *src = off9; *n = NP * 18; }
        else if (nm == "aff9") { *src = aff9; *n = NP * 9; }
        else if (nm == "goff9") { *src = goff9; *n = NP * 18; }
        else if (nm == "gaff9") { *src = gaff9; *n = NP * 9; }
        else return fail("unknown debug tensor " + nm, -2);
        return 0;
    }
    int forward(const float* image, const float* sparse, bool train, hipStream_t s) override;
    int backward(hipStream_t s) override;
    int stage_inputs_nhwc(const float* image, const float* sparse, bool train, hipStream_t s);
    // stage-2 head trainer (ghead.hip): rows = fe6 of the real pass / of the zero-image pass (nlspnmodel_adapt.py:1028-1046); MLP(512, 1024, 1024)
    int head_spec(HeadSpec* hs) override {
        if (heads_adapted) return fail("the stage-2 head trainer runs on a handle whose heads are not in the adapted list (no PTTA_SYNCBN_ADAPT)", -38);
        hs->x_real = t_fe6; hs->xw_real = W_GRAD; hs->x_proxy = t_fe6; hs->xw_proxy = W_PROXY; hs->hidden = 1024; hs->out = 1024;
        return 0;
    }
    int head_features(const float* image, const float* sparse, hipStream_t s) override {
        for (auto& ad : adapted) if (!ad.p) return fail("adapted parameter " + ad.name + " not bound (ptta_bind_adapted)", -3);
        NRUN(stage_inputs_nhwc(image, sparse, true, s));
        repack_adapted(s);
        return run_ops_fwd(true, s, enc_ops_end);
    }
};

int nlspn_engine::forward(const float* image, const float* sparse, bool train, hipStream_t s) {
    for (auto& ad : adapted) if (!ad.p) return fail("adapted parameter " + ad.name + " not bound (ptta_bind_adapted)", -3);
    const long P = (long)H * W;
    NRUN(stage_inputs_nhwc(image, sparse, train, s));
    repack_adapted(s);                  // the adapted conv is re-packed from the bound tensor on every forward
    const int rc = run_ops_fwd(train, s);
    if (rc) return rc;
    // propagation (nlspnmodel_adapt.py:340-373) and the final clamp (:900)
    const GView oa = view(t_oa, W_GRAD, train);
    if (ptta_launch_nl_affinity_fwd(oa, T[t_conf].p, S, legacy, off9, aff9, s)) return fail("affinity launch failed", -5);
    // feats[k], k < PROP_TIME: the map sweep k samples, with the sparse input imposed; feats[PROP_TIME]: the raw last output
    if (ptta_launch_nl_pin(T[t_pred].p, T[t_sd].p, feats, (long)N * P, s)) return fail("pin launch failed", -5);
    for (int k = 0; k < PROP_TIME; ++k)
        if (ptta_launch_nl_prop_fwd(feats + (size_t)k * N * P, T[t_sd].p, off9, aff9, feats + (size_t)(k + 1) * N * P, k + 1 < PROP_TIME ? 1 : 0, N, H, W, s))
            return fail("propagation launch failed", -5);
    hipLaunchKernelGGL(relu_copy_kernel, dim3(nb((long)N * P)), dim3(256), 0, s, feats + (size_t)PROP_TIME * N * P, depth, (long)N * P);
    if (hipGetLastError() != hipSuccess) return fail("launch failed", -5);
    fwd_valid = train;
    return 0;
}

// inputs: image -> NHWC (normalised on the fly; proxy half = zero image), sparse depth clamped (external_model_adapt.py:108)
int nlspn_engine::stage_inputs_nhwc(const float* image, const float* sparse, bool train, hipStream_t s) {
    const long P = (long)H * W;
    const int Be = train ? 2 * N : N;
    GView iv = view(t_img, W_BOTH, train);
    if (ptta_launch_gnchw_to_nhwc(image, N, 3, iv, N, norm_on, norm_div, norm_mean, norm_std, s)) return fail("image staging failed", -5);
    hipLaunchKernelGGL(clamp_dup_kernel, dim3(nb(P * Be)), dim3(256), 0, s, sparse, T[t_sd].p, (long)N * P, Be / N, hp.max_input_depth);
    if (t_sd16 >= 0) {
        GView sv = view(t_sd16, W_BOTH, train);
        if (ptta_launch_gnchw_to_nhwc(T[t_sd].p, Be, 1, sv, Be, 0, 1.f, nullptr, nullptr, s)) return fail("sparse staging failed", -5);
    }
    return 0;
}

// consumes gdepth (N,1,H,W) and the gradient of `ref` already stored in T[t_ref].g
int nlspn_engine::backward(hipStream_t s) {
    if (!fwd_valid) return fail("backward without a training forward", -3);
    const long NP = (long)N * H * W;
    hipLaunchKernelGGL(clamp_bwd_kernel, dim3(nb(NP)), dim3(256), 0, s, gdepth, feats + (size_t)PROP_TIME * NP, gy, NP);
    if (hipMemsetAsync(goff9, 0, (size_t)NP * 18 * sizeof(float), s) != hipSuccess) return fail("memset failed", -5);
    if (hipMemsetAsync(gaff9, 0, (size_t)NP * 9 * sizeof(float), s) != hipSuccess) return fail("memset failed", -5);
    float* gcur = gy; float* gnext = gping;
    for (int k = PROP_TIME - 1; k >= 0; --k) {
        float* dst = k == 0 ? T[t_pred].g : gnext;
        if (hipMemsetAsync(dst, 0, (size_t)NP * sizeof(float), s) != hipSuccess) return fail("memset failed", -5);
        if (ptta_launch_nl_prop_bwd(feats + (size_t)k * NP, T[t_sd].p, off9, aff9, gcur, dst, goff9, gaff9, N, H, W, s))
            return fail("propagation gradient failed", -5);
        float* t = gcur; gcur = dst; gnext = t;
    }
    if (hipMemsetAsync(T[t_conf].g, 0, (size_t)NP * sizeof(float), s) != hipSuccess) return fail("memset failed", -5);
    if (ptta_launch_nl_affinity_bwd(view(t_oa, W_GRAD, true), T[t_conf].p, S, legacy, goff9, gaff9, view(t_oa, W_GRAD, true, true), T[t_conf].g, s))
        return fail("affinity gradient failed", -5);
    return run_ops_bwd(s);
}

GNet* nlspn_create(int n, int h, int w, const ptta_hparams* hp, int legacy_offset, int* rc) {
    *rc = 0;
    if (n < 1 || h < 16 || w < 16 || !hp) { *rc = -22; return nullptr; }
    nlspn_engine* e = new nlspn_engine();
    e->N = e->Nu = n; e->H = e->Hu = h; e->W = e->Wu = w; e->hp = *hp; e->legacy = (legacy_offset & 1) ? 1 : 0; e->heads_adapted = (legacy_offset & 2) ? 1 : 0;
    const PttaCreateEnv env = ptta_create_env();
    e->naive = env.naive;          // direct fp32 kernels everywhere (validation)
    e->x6 = 0;                     // (third operand plane: off -- matrix-core bound, parity holds without)
    // hipGraph replay of the step / eval forward: built and bit-identical (tests), but measured 0.3 - 2 % SLOWER than kernel-by-kernel
    // launches on this engine (the host keeps ahead of the GPU either way: DESIGN_LOG.md section 9) -> opt-in: ptta_set_option(h, "graph", 1) / PTTA_GRAPH=1
    e->use_graph = env.graph == 1 ? 1 : 0;
    e->build();
    if (e->oom || !e->step_dev) { delete e; *rc = -12; return nullptr; }
    const float one[1] = {4.0f};                                  // affinity_gamma * num = 0.5 * 8 (nlspnmodel_adapt.py:231-233)
    if (hipMemcpy(e->S, one, sizeof(one), hipMemcpyHostToDevice) != hipSuccess) { delete e; *rc = -5; return nullptr; }
    if (e->upload_hparams(nullptr)) { delete e; *rc = -5; return nullptr; }
    return e;
}
