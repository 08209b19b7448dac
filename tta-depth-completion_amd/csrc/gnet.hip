// Generic layer-graph engine (gnet.h): op execution, weight loading and the C-ABI entry points that are identical for
// every backbone built on it (NLSPN, CostDCNet).
#include "gnet.h"

using namespace gnet;

namespace gnet {
__global__ void gnet_pad_channels_kernel(const float* __restrict__ src, int lds_, int C, float* __restrict__ dst, int Cp, long npix) {
    const long total = npix * Cp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp); const long p = i / Cp;
        dst[i] = c < C ? src[p * lds_ + c] : 0.f;
    }
}
__global__ void gnet_validity_kernel(const float* __restrict__ sp, float* __restrict__ v, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { const float s = sp[i]; v[i] = s > 0.f ? 1.f : s; }
}
}  // namespace gnet

// ---- weights -----------------------------------------------------------------------------------------------------------
int GNet::load(const char* name_c, const void* tensor_, const int64_t* shape, int ndim, hipStream_t s) {
    const std::string name(name_c);
    const float* src = (const float*)tensor_;
    for (Op& o : ops) o.st_eval_valid = false;                        // any load may change an affine parameter or a running statistic
    drop_graphs();                                                    // ... or re-bind a running-statistics pointer
    if (aid.count(name)) return 0;                                   // adapted: the bound tensor is authoritative
    const size_t dot = name.rfind('.');
    if (dot == std::string::npos) return load_extra(name, src, shape, ndim, s);
    const std::string base = name.substr(0, dot), leaf = name.substr(dot + 1);
    auto fb = frozen_bn.find(base);
    if (fb != frozen_bn.end() && (leaf == "weight" || leaf == "bias")) {
        float* dst = leaf == "weight" ? fb->second.first : fb->second.second;
        NCHK(hipMemcpyAsync(dst, src, (size_t)shape[0] * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    }
    if (leaf == "running_mean" || leaf == "running_var") {
        // BatchNorm2d of the backbone: the TTA step never reads them (adapt_parameters('meta_bn') drops them), the stage-2 head trainer does
        for (Op& o : ops)
            if (o.kind == K_BN && !o.tracked && o.bname == base && o.rm_own && ndim == 1 && shape[0] == T[o.x[0]].C) {
                NCHK(hipMemcpyAsync(leaf == "running_mean" ? o.rm_own : o.rv_own, src, (size_t)shape[0] * sizeof(float), hipMemcpyDeviceToDevice, s));
                o.has_running = true;
            }
    }
    auto it = convs.find(base);
    if (it == convs.end()) return load_extra(name, src, shape, ndim, s);
    GConvW& cw = it->second;
    if (leaf == "bias") {
        if (ndim != 1 || shape[0] != cw.Co) return fail("shape mismatch for " + name, -22);
        NCHK(hipMemcpyAsync(cw.bias, src, (size_t)cw.Co * sizeof(float), hipMemcpyDeviceToDevice, s));
        cw.has_bias = true;
        return 0;
    }
    if (leaf != "weight") return load_extra(name, src, shape, ndim, s);
    const int Ci = cw.Ci_real ? cw.Ci_real : cw.Ci;
    if (cw.vcol) {                                 // Conv3d (Co, Ci, 3, 1, 1): vertical taps of a 3x3 filter over [D][H*W]
        if (ndim != 5 || shape[0] != cw.Co || shape[1] != Ci || shape[2] != 3 || shape[3] != 1 || shape[4] != 1) return fail("shape mismatch for " + name, -22);
        for (int ky = 0; ky < 3; ++ky) {
            ptta_gpack(src + ky, cw.wf + (size_t)(ky * 3 + 1) * Ci * cw.Co, 1, Ci, cw.Co, 3, (long)Ci * 3, 0, s);
            ptta_gpack(src + (2 - ky), cw.wb + (size_t)(ky * 3 + 1) * cw.Co * Ci, 1, cw.Co, Ci, (long)Ci * 3, 3, 0, s);
        }
        pack_frags(cw, s);
        cw.loaded = true;
        return 0;
    }
    if (cw.k == 1 && ndim == 2) {                  // nn.Linear (N, K)
        if (shape[0] != cw.Co || shape[1] != Ci) return fail("shape mismatch for " + name, -22);
    } else if (ndim == 5) {                        // Conv3d (Co, Ci, 1, k, k): one 2-D filter per depth plane
        if (shape[0] != cw.Co || shape[1] != Ci || shape[2] != 1 || shape[3] != cw.k || shape[4] != cw.k) return fail("shape mismatch for " + name, -22);
    } else if (ndim != 4 || shape[2] != cw.k || shape[3] != cw.k ||
               (cw.transposed ? (shape[0] != Ci || shape[1] != cw.Co) : (shape[0] != cw.Co || shape[1] != Ci)))
        return fail("shape mismatch for " + name, -22);
    pack_conv_weight(cw, src, s);
    cw.loaded = true;
    return 0;
}

// ---- execution ---------------------------------------------------------------------------------------------------------
int GNet::run_conv_fwd(const Op& o, bool train, hipStream_t s) {
    const bool ad = o.ad_w >= 0;
    const GConvW& cw = convs[o.wname];
    if (!ad && !cw.loaded) return fail("weights of " + o.wname + " not loaded (ptta_load_weights)", -3);
    const float* bias = ad ? (o.ad_b >= 0 ? adapted[o.ad_b].p : nullptr) : (cw.has_bias ? cw.bias : nullptr);
    if (cw.mf) {
        GX3Args a;
        GView x0 = view(o.x[0], o.xw[0], train), y = view(o.y, o.yw, train);
        review(x0, o.rH, o.rW); review(y, o.rH, o.rW);
        a.x0 = x0.p; a.C0 = x0.C; a.ld0 = x0.ld;
        if (o.nsrc == 2) { const GView x1 = view(o.x[1], o.xw[1], train); a.x1 = x1.p; a.C1 = x1.C; a.ld1 = x1.ld; }
        a.B = y.B; a.H = y.H; a.W = y.W;
        a.whi = (const uint4*)cw.ff_hi; a.wlo = (const uint4*)cw.ff_lo;
        a.nchunks = (a.C0 + 31) / 32 + (a.C1 + 31) / 32; a.nf0 = 0; a.nnf = (cw.Co + 31) / 32;
        a.y = y.p; a.ldy = y.ld; a.Cy = y.C; a.bias = bias; a.act = o.act; a.vert = cw.vcol;
        if (cw.ff_l2 && o.yw != W_PROXY && x6_layer(o.wname)) {               // bf16x6 for the real frames (they come first in a [real | proxy] batch)
            a.wl2 = (const uint4*)cw.ff_l2; a.six_B = (o.yw == W_BOTH && train) ? y.B / 2 : y.B;
        }
        if ((mixed & 1) && train && o.yw != W_GRAD) a.x1_from_B = o.yw == W_BOTH ? y.B / 2 : 0;     // the proxy frames' products: one MFMA
        if (o.stat_to >= 0 && (train || !ops[o.stat_to].tracked) && !(bn_prepare && !ops[o.stat_to].tracked)) {
            a.stat_part = ops[o.stat_to].part; a.stat_C = y.C; a.stat_npass = (o.yw == W_BOTH && train) ? 2 : 1;
        }
        int rc;
        if (o.stride == 1 && !o.transposed) rc = ptta_launch_gconv_x3(a, o.k, s);
        else rc = ptta_launch_gconv_x3_strided(a, o.k, o.transposed ? 2 : 1, x0.H, x0.W, s);
        if (rc) return fail("conv " + o.wname + " (matrix-core) launch failed", -5);
        return 0;
    }
    for (int sidx = 0; sidx < o.nsrc; ++sidx) {
        GConvArgs a;
        a.x = view(o.x[sidx], o.xw[sidx], train); a.y = view(o.y, o.yw, train);
        review(a.x, o.rH, o.rW); review(a.y, o.rH, o.rW);
        a.w = cw.wf + (size_t)o.c0[sidx] * cw.Co; a.wld = cw.Co; a.wts = (long)cw.Ci * cw.Co;
        a.bias = sidx == 0 ? bias : nullptr;
        a.k = o.k; a.stride = o.stride; a.transposed = o.transposed;
        a.accumulate = sidx > 0; a.act = sidx == o.nsrc - 1 ? o.act : GACT_NONE;
        const int rc = ptta_launch_gconv_direct(a, s);
        if (rc) return fail("conv " + o.wname + " launch failed", -5);
    }
    return 0;
}

int GNet::run_bn_fwd(const Op& o, bool train, hipStream_t s) {
    GView x = view(o.x[0], o.xw[0], train), y = view(o.y, o.yw, train);
    GView res; if (o.res >= 0) res = view(o.res, o.xw[0], train);
    const int npass = (o.xw[0] == W_BOTH && train) ? 2 : 1;
    if (bn_prepare && !o.tracked) {                         // stage-2 head trainer: a BatchNorm2d in eval mode, every pass from the loaded running statistics
        if (!o.has_running) return fail("running statistics of " + o.bname + " not loaded (the stage-2 head trainer evaluates BatchNorm2d from them)", -3);
        if (ptta_launch_gbn_eval_affine(bn_gamma(o), bn_beta(o), o.rm_own, o.rv_own, BN_EPS, x.C, o.st_eval, s)) return fail("batch-norm " + o.bname + " (prepare) launch failed", -5);
        o.st_eval_valid = false;
        if (ptta_launch_gbn_apply(x, res, y, 1, o.act, o.st_eval, 1, s, o.act_first ? 1 : 0)) return fail("batch-norm " + o.bname + " (prepare) launch failed", -5);
        return 0;
    }
    if (o.tracked && !train) {                              // eval mode: running statistics
        if (!o.rm || !o.rv) return fail("running statistics of " + o.bname + " not loaded", -3);
        // the affine of the running statistics is left behind by the training forward's finalize; computed here only after a load()
        if (!o.st_eval_valid) {
            if (ptta_launch_gbn_eval_affine(bn_gamma(o), bn_beta(o), o.rm, o.rv, BN_EPS, x.C, o.st_eval, s)) return fail("batch-norm " + o.bname + " (eval) launch failed", -5);
            o.st_eval_valid = true;
        }
        if (ptta_launch_gbn_apply(x, res, y, 1, o.act, o.st_eval, 1, s, o.act_first ? 1 : 0))
            return fail("batch-norm " + o.bname + " (eval) launch failed", -5);
        return 0;
    }
    GView xt = x; review(xt, o.rH, o.rW);
    const int fused = o.fused_from >= 0 ? ptta_gconv_x3_tiles(xt.B / npass, xt.H, xt.W) : 0;
    const bool upd = o.tracked && train && o.rm && o.rv;      // running statistics + eval affine out of the finalize launch
    if (ptta_launch_gbn_forward(x, res, y, npass, o.act, BN_EPS, bn_gamma(o), bn_beta(o), fused ? o.part : bn_part, o.st, s, fused, o.act_first ? 1 : 0, &stat_sync,
                                upd ? o.rm : nullptr, upd ? o.rv : nullptr, upd ? o.nbt : nullptr, 0.1f, o.stat_repeats, upd ? o.st_eval : nullptr))
        return fail("batch-norm " + o.bname + " launch failed", -5);
    if (upd) o.st_eval_valid = true;
    return 0;
}

int GNet::run_conv_bwd(const Op& o, hipStream_t s) {
    const bool ad = o.ad_w >= 0;
    const GConvW& cw = convs[o.wname];
    GView gy = view(o.y, W_GRAD, true, true);
    GView yv = view(o.y, W_GRAD, true);
    if (o.act != GACT_NONE && ptta_launch_gact_bwd(gy, yv, o.act, s)) return fail("activation gradient failed", -5);
    review(gy, o.rH, o.rW);
    bool padded = false;
    for (int sidx = 0; sidx < o.nsrc; ++sidx) {
        if (!T[o.x[sidx]].need_grad || o.no_dx) continue;
        GView gx = view(o.x[sidx], W_GRAD, true, true);
        review(gx, o.rH, o.rW);
        const bool own_frags = sidx == 1 && cw.fb1_hi != nullptr;
        if (cw.mb && ((o.c0[sidx] % 32) == 0 || own_frags)) {
            GX3Args a;
            a.x0 = gy.p; a.C0 = gy.C; a.ld0 = gy.ld;
            if (cw.gpad) {                                   // < 16 gradient channels: matrix-core kernel on a zero-padded copy
                if (sidx == 0 || !padded) {
                    hipLaunchKernelGGL(gnet_pad_channels_kernel, dim3(nb((long)gy.B * gy.H * gy.W * cw.Co_pad)), dim3(256), 0, s, gy.p, gy.ld, gy.C,
                                       cw.gpad, cw.Co_pad, (long)gy.B * gy.H * gy.W);
                    padded = true;
                }
                a.x0 = cw.gpad; a.C0 = cw.Co_pad; a.ld0 = cw.Co_pad;
            }
            a.whi = (const uint4*)(own_frags ? cw.fb1_hi : cw.fb_hi); a.wlo = (const uint4*)(own_frags ? cw.fb1_lo : cw.fb_lo);
            a.nchunks = (a.C0 + 31) / 32; a.nf0 = own_frags ? 0 : o.c0[sidx] / 32; a.nnf = (gx.C + 31) / 32;
            a.y = gx.p; a.ldy = gx.ld; a.Cy = gx.C; a.accumulate = o.first_x[sidx] ? 0 : 1;
            a.B = gx.B; a.H = gx.H; a.W = gx.W;                  // output geometry = the source's
            a.vert = cw.vcol;
            if (mixed & 6) { a.x1_from_B = 0; a.x1_w2 = (mixed & 4) ? 1 : 0; }      // data gradients: hi activations; bit 1: x bf16-rounded weights, bit 2: x (hi + lo) weights
            int rc;
            if (o.stride == 1 && !o.transposed) rc = ptta_launch_gconv_x3(a, o.k, s);
            else rc = ptta_launch_gconv_x3_strided(a, o.k, o.transposed ? 1 : 2, gy.H, gy.W, s);     // convT -> strided conv, strided conv -> convT
            if (rc) return fail("data gradient of " + o.wname + " (matrix-core) failed", -5);
            continue;
        }
        GConvArgs a;
        a.x = gy; a.y = gx;
        a.w = cw.wb + o.c0[sidx]; a.wld = cw.Ci; a.wts = (long)cw.Co * cw.Ci;
        a.k = o.k; a.act = GACT_NONE; a.accumulate = o.first_x[sidx] ? 0 : 1;
        if (o.transposed) { a.transposed = 0; a.stride = o.stride; }           // convT -> strided conv
        else if (o.stride == 2) { a.transposed = 1; a.stride = 2; }             // strided conv -> convT
        else { a.transposed = 0; a.stride = 1; }
        if (ptta_launch_gconv_direct(a, s)) return fail("data gradient of " + o.wname + " failed", -5);
    }
    if (ad) {
        const GView xv = view(o.x[0], o.xw[0] == W_PROXY ? W_PROXY : W_GRAD, true);
        float* gw = gall + adapted[o.ad_w].goff;
        float* gb = o.ad_b >= 0 ? gall + adapted[o.ad_b].goff : nullptr;
        if (o.k == 1) {                                  // nn.Linear / 1x1 convolution (the heads): dW[co][ci] = sum_rows gy[r][co] x[r][ci]
            if (ptta_launch_glinear_wgrad(xv, gy, gw, gb, s)) return fail("weight gradient of " + o.wname + " failed", -5);
            return 0;
        }
        if ((naive ? ptta_launch_gwgrad(xv, gy, wg_part, gw, gb, s) : ptta_launch_gwgrad_mfma(xv, gy, wg_part, gw, gb, s)))
            return fail("weight gradient failed", -5);
    }
    return 0;
}

int GNet::run_bn_bwd(const Op& o, hipStream_t s) {
    const GView x = view(o.x[0], W_GRAD, true), y = view(o.y, W_GRAD, true);
    const GView g = view(o.y, W_GRAD, true, true), gx = view(o.x[0], W_GRAD, true, true);
    GView gres; if (o.res >= 0) gres = view(o.res, W_GRAD, true, true);
    const int npass = o.xw[0] == W_BOTH ? 2 : 1;
    float* dg = o.ad_g >= 0 ? gall + adapted[o.ad_g].goff : nullptr;
    float* db = o.ad_beta >= 0 ? gall + adapted[o.ad_beta].goff : nullptr;
    if (ptta_launch_gbn_backward(x, g, y, gx, gres, npass, o.act, o.res >= 0 ? 1 : 0, o.first_raw ? 0 : 1, o.first_res ? 0 : 1,
                                 bn_gamma(o), o.st, bn_part, bn_bw, dg, db, s, o.act_first ? 1 : 0, &stat_sync))
        return fail("batch-norm gradient of " + o.bname + " failed", -5);
    return 0;
}

// ---- entry points ------------------------------------------------------------------------------------------------------
int GNet::set_image_norm(float div, const float* mean, const float* stdv) {
    if (!(div > 0.f)) return fail("ptta_set_image_norm: divisor must be positive", -22);
    norm_div = div;
    for (int k = 0; k < 3; ++k) { norm_mean[k] = mean ? mean[k] : 0.f; norm_std[k] = stdv ? stdv[k] : 1.f; if (!(norm_std[k] > 0.f)) return fail("std must be positive", -22); }
    drop_graphs();                                 // the constants are kernel arguments of the captured launches
    norm_on = !(div == 1.f && norm_mean[0] == 0.f && norm_mean[1] == 0.f && norm_mean[2] == 0.f && norm_std[0] == 1.f &&
                norm_std[1] == 1.f && norm_std[2] == 1.f);
    return 0;
}
int GNet::bind_adapted(const char* name, float* p, float* m, float* v) {
    auto it = aid.find(name ? name : "");
    if (it == aid.end()) return fail(std::string("not an adapted parameter: ") + (name ? name : "(null)"), -2);
    Adapted& a = adapted[it->second];
    if (a.p != p || a.m != m || a.v != v) drop_graphs();
    a.p = p; a.m = m; a.v = v;
    adam_tab_dirty = true;
    return 0;
}
int GNet::forward_train(const float* image, const float* sparse, float* depth_out, float* emb, float* ref, hipStream_t s) {
    const int rc = forward(image, sparse, true, s);
    if (rc) return rc;
    const size_t nb_ = (size_t)Nu * Hu * Wu * sizeof(float), eb = (size_t)rows() * emb_dim() * sizeof(float);
    if (depth_out && hipMemcpyAsync(depth_out, depth, nb_, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    if (emb && hipMemcpyAsync(emb, T[t_emb].p, eb, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    if (ref && hipMemcpyAsync(ref, T[t_ref].p, eb, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    return 0;
}
int GNet::stage_inputs(const float* image, const float* loss_image, const float* sparse, const float* validity, hipStream_t s) {
    const size_t ib = (size_t)Nu * 3 * Hu * Wu * sizeof(float), pb = (size_t)Nu * Hu * Wu * sizeof(float);
    if (!gi_image) {                                     // once, outside any capture
        gi_image = falloc((size_t)Nu * 3 * Hu * Wu); gi_loss = falloc((size_t)Nu * 3 * Hu * Wu);
        gi_sparse = falloc((size_t)Nu * Hu * Wu); gi_valid = falloc((size_t)Nu * Hu * Wu);
        if (oom) return fail("out of device memory (graph input staging)", -12);
    }
    NCHK(hipMemcpyAsync(gi_image, image, ib, hipMemcpyDeviceToDevice, s));
    NCHK(hipMemcpyAsync(gi_sparse, sparse, pb, hipMemcpyDeviceToDevice, s));
    if (loss_image && loss_image != image) NCHK(hipMemcpyAsync(gi_loss, loss_image, ib, hipMemcpyDeviceToDevice, s));
    if (validity) NCHK(hipMemcpyAsync(gi_valid, validity, pb, hipMemcpyDeviceToDevice, s));
    return 0;
}
int GNet::replay(int key, std::function<int(hipStream_t)> body, hipStream_t s) {
    if (!gexec[key]) {
        if (!cap_stream) NCHK(hipStreamCreateWithFlags(&cap_stream, hipStreamNonBlocking));
        NCHK(hipStreamBeginCapture(cap_stream, hipStreamCaptureModeThreadLocal));
        const int rc = body(cap_stream);
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(cap_stream, &g);
        if (rc != 0) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (e != hipSuccess || !g) return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e), -100 - (int)e);
        graph[key] = g;
        NCHK(hipGraphInstantiate(&gexec[key], g, nullptr, nullptr, 0));
    }
    NCHK(hipGraphLaunch(gexec[key], s));
    if (!ev_replay) NCHK(hipEventCreateWithFlags(&ev_replay, hipEventDisableTiming));
    NCHK(hipEventRecord(ev_replay, s));
    return 0;
}
int GNet::forward_eval(const float* image, const float* sparse, float* depth_out, hipStream_t s) {
    int rc;
    if (graph_ok() && eager_eval_done) {
        rc = stage_inputs(image, nullptr, sparse, nullptr, s);
        if (rc) return rc;
        rc = replay(4, [this](hipStream_t cs) { return forward(gi_image, gi_sparse, false, cs); }, s);
        fwd_valid = false;
    } else {
        rc = forward(image, sparse, false, s);
        eager_eval_done = true;
    }
    if (rc) return rc;
    if (depth_out && hipMemcpyAsync(depth_out, depth, (size_t)Nu * Hu * Wu * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    return 0;
}
int GNet::upload_adam_table(hipStream_t s) {           // after ptta_bind_adapted only: pointer table of every adapted tensor
    adam_host.resize(adapted.size());
    long off = 0;
    for (size_t k = 0; k < adapted.size(); ++k) { const Adapted& ad = adapted[k]; adam_host[k] = PttaAdamEntry{ad.p, ad.m, ad.v, gall + ad.goff, ad.n, off, ad.rep}; off += ad.n; }
    NCHK(hipMemcpyAsync(adam_tab, adam_host.data(), adam_host.size() * sizeof(PttaAdamEntry), hipMemcpyHostToDevice, s));
    adam_tab_dirty = false;
    return 0;
}
int GNet::adam_step(hipStream_t s) {
    for (auto& ad : adapted) if (!ad.p || !ad.m || !ad.v) return fail("Adam state of " + ad.name + " not bound", -3);
    if (adam_tab_dirty) { const int rc = upload_adam_table(s); if (rc) return rc; }
    if (ptta_launch_adam_multi(adam_tab, (int)adapted.size(), gall_n, hyper, step_dev, adam_ticket, s)) return fail("adam failed", -5);
    return 0;
}
int GNet::step(const float* image, const float* loss_image, const float* sparse, const float* validity, float* depth_out, float* loss_info_out, hipStream_t s) {
    for (auto& ad : adapted) if (!ad.p || !ad.m || !ad.v) return fail("Adam state of " + ad.name + " not bound", -3);
    if (!loss_image) loss_image = image;
    int rc;
    if (graph_ok() && eager_step_done) {
        const int key = (loss_image != image ? 1 : 0) | (validity ? 2 : 0);
        rc = stage_inputs(image, loss_image, sparse, validity, s);
        if (rc) return rc;
        if (adam_tab_dirty) { rc = upload_adam_table(s); if (rc) return rc; }          // a host -> device copy: never under capture
        rc = replay(key, [this, key](hipStream_t cs) { return step_body(gi_image, (key & 1) ? gi_loss : gi_image, gi_sparse, (key & 2) ? gi_valid : nullptr, cs); }, s);
    } else {
        rc = step_body(image, loss_image, sparse, validity, s);
        eager_step_done = true;
    }
    if (rc) return rc;
    const long NP = (long)Nu * Hu * Wu;
    if (depth_out && hipMemcpyAsync(depth_out, depth, (size_t)NP * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    if (loss_info_out && hipMemcpyAsync(loss_info_out, loss_info, 4 * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    fwd_valid = false;
    return 0;
}
int GNet::step_body(const float* image, const float* loss_image, const float* sparse, const float* validity, hipStream_t s) {
    int rc = forward(image, sparse, true, s);
    if (rc) return rc;
    const float* emb = T[t_emb].p; const float* ref = T[t_ref].p;
    // validity == NULL: evaluated inside the loss kernels; the finalisation runs inside the two gradient kernels
    if (ptta_launch_loss_forward(depth, loss_image, sparse, validity, hp.max_input_depth, emb, ref, rows(), emb_dim(), hyper + 5,
                                 Nu, Hu, Wu, loss_ws, loss_info, s, 1)) return fail("loss forward failed", -5);
    if (ptta_launch_loss_backward(depth, loss_image, sparse, validity, hp.max_input_depth, emb, ref, rows(), emb_dim(), Nu, Hu,
                                  Wu, loss_ws, gdepth, T[t_ref].g, s, hyper + 5, loss_info)) return fail("loss backward failed", -5);
    rc = backward(s);
    if (rc) return rc;
    if (grad_comm && ptta_rccl_allreduce_mean_f32(grad_comm, gall, gall_n, (ptta_stream)s)) return fail(std::string("gradient all-reduce: ") + ptta_rccl_last_error(), -5);
    return adam_step(s);
}
int GNet::get_grad(const char* name, float* dst, int64_t capacity, hipStream_t s) {
    auto it = aid.find(name ? name : "");
    if (it == aid.end()) return fail(std::string("not an adapted parameter: ") + (name ? name : "(null)"), -2);
    const Adapted& a = adapted[it->second];
    if (capacity < a.n) return fail("ptta_get_grad: destination too small", -22);
    if (hipMemcpyAsync(dst, gall + a.goff, (size_t)a.n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    return 0;
}
int GNet::set_grad(const char* name, const float* src, int64_t numel, hipStream_t s) {
    auto it = aid.find(name ? name : "");
    if (it == aid.end()) return fail(std::string("not an adapted parameter: ") + (name ? name : "(null)"), -2);
    const Adapted& a = adapted[it->second];
    if (numel != a.n) return fail("ptta_set_grad: size mismatch", -22);
    if (hipMemcpyAsync(gall + a.goff, src, (size_t)a.n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    return 0;
}
int GNet::debug_tensor(const char* name, float* dst, int64_t capacity, int64_t* numel, hipStream_t s) {
    std::string nm(name ? name : "");
    bool grad = false;
    if (nm.size() > 5 && nm.compare(0, 5, "grad:") == 0) { grad = true; nm = nm.substr(5); }
    const float* src = nullptr; long n = 0;
    if (nm == "depth") { src = depth; n = (long)Nu * Hu * Wu; }
    else {
        auto it = tid.find(nm);
        if (it == tid.end()) { const int rc = debug_extra(nm, &src, &n); if (rc) return rc; }
        else {
            const Tn& t = T[it->second];
            if (t.ld != t.C) return fail("debug tensor " + nm + " is a strided slice", -22);
            src = grad ? t.g : t.p; n = (long)(grad ? t.per : t.items) * t.H * t.W * t.C;
            if (!src) return fail("debug tensor " + nm + " has no gradient buffer", -2);
        }
    }
    if (numel) *numel = n;
    if (!dst) return 0;
    if (capacity < n) return fail("debug tensor: destination too small", -22);
    if (hipMemcpyAsync(dst, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    return 0;
}

// ---- the reference's split calls: compute_loss / loss.backward() / optimizer.step() (src/tta_main.py:619-633) ----------
int GNet::loss_forward(const float* loss_image, const float* depth_, const float* sparse, const float* validity, const float* emb, const float* ref,
                       int64_t rows_, float w_sd, float w_sm, float w_cos, float* loss_info_out, hipStream_t s) {
    if (rows_ > rows()) return fail("rows exceeds the handle's embedding rows", -22);
    const float w3[3] = {w_sd, w_sm, w_cos};
    if (ptta_launch_set_floats(w3_tmp, w3, 3, s)) return fail("loss weight upload failed", -5);
    if (ptta_launch_loss_forward(depth_, loss_image, sparse, validity, hp.max_input_depth, emb, ref, rows_, emb_dim(), w3_tmp, Nu, Hu, Wu,
                                 loss_ws, loss_info_out, s)) return fail("loss forward failed", -5);
    return 0;
}
int GNet::loss_backward(const float* loss_image, const float* depth_, const float* sparse, const float* validity, const float* emb, const float* ref,
                        int64_t rows_, float* grad_depth_out, float* grad_ref_out, hipStream_t s) {
    if (ptta_launch_loss_backward(depth_, loss_image, sparse, validity, hp.max_input_depth, emb, ref, rows_, emb_dim(), Nu, Hu, Wu,
                                  loss_ws, grad_depth_out, grad_ref_out, s)) return fail("loss backward failed", -5);
    return 0;
}
int GNet::backward_from(const float* grad_depth, const float* grad_ref, hipStream_t s) {
    if (!fwd_valid) return fail("ptta_backward without a preceding ptta_forward_train", -3);
    const size_t nb_ = (size_t)Nu * Hu * Wu * sizeof(float), rb = (size_t)rows() * emb_dim() * sizeof(float);
    if (hipMemcpyAsync(gdepth, grad_depth, nb_, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5);
    if (grad_ref) { if (hipMemcpyAsync(T[t_ref].g, grad_ref, rb, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail("memcpy failed", -5); }
    else if (hipMemsetAsync(T[t_ref].g, 0, rb, s) != hipSuccess) return fail("memset failed", -5);
    return backward(s);
}
