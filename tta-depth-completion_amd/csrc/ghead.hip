// Stage-2 head trainer on the generic engine (SURVEY.md 8f-4 for the NLSPN and CostDCNet backbones; include/ptta.h ptta_head_*).
//
// One step of src/head_main.py:464-480 with loss_type 'head_selfsup_seq_ema[_reverse]':
//   NLSPN     external_src/NLSPN/src/model/nlspnmodel_adapt.py:1014-1060 (`_rgbd_meta_contrast_prepare`), EMA :1314-1316
//   CostDCNet external_src/costdcnet/CostDCNet_adapt.py:258-303, EMA :426-428
//     _update_head()      proj_t <- tau proj_t + (1 - tau) proj over parameters(), before the heads run
//     both backbone passes under no_grad; `train(prepare=True)` (src/nlspn_model_adapt.py:360-368, src/costdcnet_model_adapt.py:418-430) put every
//                         BatchNorm2d that is not a head's into eval mode: the backbone normalises with its LOADED running statistics
//     not reverse:        emb = pred(proj(rows(real).detach())),  ref = proj_t(rows(zero image)).detach()
//     reverse:            emb = pred(proj(rows(zero image).detach())), ref = proj_t(rows(real)).detach()
//                         (unlike MSG_CHN the reference branch is the EMA target, and proj trains in both directions)
//     heads' BatchNorm1d  train mode (batch statistics, running statistics updated) -- proj_t's too on a single process (the isinstance test of
//                         train_prepare names BatchNorm2d / SyncBatchNorm only)
//     prepare_loss        mean(2 - 2 <normalize(emb), normalize(ref)>)     src/external_model_adapt.py:524-541
//     Adam                over prepare_parameters('head_selfsup_ema') = the twelve proj.* / pred.* tensors
//
// The heads are a second op list of the engine ("head program": 1x1 CONV - BN (tracked) - 1x1 CONV, three times) built on first use; while it
// runs it is swapped into the engine's program slots, so forward / backward are the engine's own op runners.  The weight gradient of a 1x1
// convolution is the one kernel of this file.
#include "gnet.h"

using namespace gnet;

// dW[co][ci] = sum_r gy[r][co] * x[r][ci], db[co] = sum_r gy[r][co]; rows in order, fp32 (R = N * H/16 * W/16 rows: 1,672 at 352x1216).
// A block owns a 64 x 64 tile of (co, ci); 256 threads x (4 x 4) accumulators; 16-row slabs of both operands staged through LDS.
__global__ __launch_bounds__(256) void glinear_wgrad_kernel(const float* __restrict__ x, int ldx, int I, const float* __restrict__ gy, int ldg, int O, long R,
                                                            float* __restrict__ gw, float* __restrict__ gb) {
    __shared__ float Xs[16][64 + 4], Gs[16][64 + 4];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;                  // ci = ci0 + 4 tx + j, co = co0 + 4 ty + i
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    float acc[4][4] = {{0.f}}, bacc[4] = {0.f, 0.f, 0.f, 0.f};
    for (long r0 = 0; r0 < R; r0 += 16) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                       // 16 rows x 64 columns of each operand: 4 elements per thread
            const int e = t + 256 * k, rr = e >> 6, cc = e & 63;
            const long r = r0 + rr;
            Xs[rr][cc] = (r < R && ci0 + cc < I) ? x[r * ldx + ci0 + cc] : 0.f;
            Gs[rr][cc] = (r < R && co0 + cc < O) ? gy[r * ldg + co0 + cc] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            float xv[4], gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[j] = Xs[rr][4 * tx + j]; gv[j] = Gs[rr][4 * ty + j]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bacc[i] += gv[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(gv[i], xv[j], acc[i][j]);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int co = co0 + 4 * ty + i;
        if (co >= O) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int ci = ci0 + 4 * tx + j; if (ci < I) gw[(long)co * I + ci] = acc[i][j]; }
        if (gb && blockIdx.x == 0 && tx == 0) gb[co] = bacc[i];
    }
}
int ptta_launch_glinear_wgrad(const GView& x, const GView& gy, float* gw, float* gb, hipStream_t s) {
    const long R = (long)x.B * x.H * x.W;
    if (R != (long)gy.B * gy.H * gy.W || R < 1 || !gw) return -22;
    hipLaunchKernelGGL(glinear_wgrad_kernel, dim3((x.C + 63) / 64, (gy.C + 63) / 64), dim3(256), 0, s, x.p, x.ld, x.C, gy.p, gy.ld, gy.C, R, gw, gb);
    PTTA_CHECK_LAUNCH();
    return 0;
}

namespace {
const char* const HM[3] = {"proj", "proj_t", "pred"};
// the engine's program slots <-> the head program
struct ProgSwap {
    GNet* g; GNet::HeadTrain* h;
    ProgSwap(GNet* g_, GNet::HeadTrain* h_) : g(g_), h(h_) { sw(); }
    ~ProgSwap() { sw(); }
    void sw() { std::swap(g->ops, h->ops); std::swap(g->adapted, h->adapted); std::swap(g->aid, h->aid); std::swap(g->gall, h->gall); std::swap(g->gall_n, h->gall_n); }
};
}  // namespace

int GNet::head_build(int reverse) {
    HeadTrain& h = head;
    if (!h.built) {
        if (stat_sync.world > 1) return fail("the stage-2 head trainer has no SyncBatchNorm exchange", -38);
        NRUN(head_spec(&h.spec));
        const Tn xr = T[h.spec.x_real];
        const int per = xr.per, Hh = xr.H, Ww = xr.W;
        {
            ProgSwap sw(this, &h);                 // build into empty program slots: conv() / bn() append to `ops`, bn() adds to `adapted`
            for (int m = 0; m < 3; ++m) {
                const bool train = m != 1;           // proj_t: forward only, frozen affine (the EMA writes its bound tensors)
                const std::string nm(HM[m]);
                const int xin = m == 2 ? h.t_h[0][2] : h.spec.x_real, xw = m == 2 ? W_GRAD : h.spec.xw_real;
                const int Cout = h.spec.out;
                h.t_h[m][0] = tensor("head/" + nm + ".h", per, Hh, Ww, h.spec.hidden, train, per);
                h.t_h[m][1] = tensor("head/" + nm + ".a", per, Hh, Ww, h.spec.hidden, train, per);
                h.t_h[m][2] = tensor("head/" + nm + ".out", per, Hh, Ww, Cout, train, per);
                {
                    Op& o = conv(nm + ".0", xin, -1, h.t_h[m][0], 1, 1, 0, GACT_NONE, xw, W_GRAD, true, train);
                    if (train) { o.ad_w = add_adapted(nm + ".0.weight", (long)T[xin].C * h.spec.hidden); o.ad_b = add_adapted(nm + ".0.bias", h.spec.hidden); o.no_dx = m == 0; }
                }
                bn(nm + ".1", h.t_h[m][0], h.t_h[m][1], -1, GACT_RELU, W_GRAD, !train, true, train).tracked = true;
                {
                    Op& o = conv(nm + ".3", h.t_h[m][1], -1, h.t_h[m][2], 1, 1, 0, GACT_NONE, W_GRAD, W_GRAD, true, train);
                    if (train) { o.ad_w = add_adapted(nm + ".3.weight", (long)h.spec.hidden * Cout); o.ad_b = add_adapted(nm + ".3.bias", Cout); }
                }
            }
            h.t_emb = h.t_h[2][2]; h.t_ref = h.t_h[1][2];
            plan_backward({h.t_emb});
        }
        if (h.adapted.size() != 12) return fail("head program: expected twelve trained tensors", -5);
        h.gall = falloc((size_t)h.gall_n);
        for (int k = 0; k < 6; ++k) { HeadTgt t; t.name = std::string("proj_t") + h.adapted[k].name.substr(4); t.n = h.adapted[k].n; h.tgt.push_back(t); }
        h.adam_tab = (PttaAdamEntry*)dalloc(12 * sizeof(PttaAdamEntry)); h.etab = (PttaAdamEntry*)dalloc(6 * sizeof(PttaAdamEntry));
        h.ticket = (unsigned*)dalloc(sizeof(unsigned)); h.step_dev = (int*)dalloc(sizeof(int));
        h.hyper = falloc(8); h.tau2 = falloc(2); h.loss = falloc(1); h.loss_part = falloc(1024);
        if (oom) return fail("out of device memory (head trainer workspace)", -12);
        const float hy[5] = {1e-3f, 0.9f, 0.999f, 1e-8f, 0.f};
        const float t2[2] = {0.999f, (float)(1.0 - 0.999)};
        if (ptta_launch_set_floats(h.hyper, hy, 5, nullptr) || ptta_launch_set_floats(h.tau2, t2, 2, nullptr)) return fail("head hyper-parameter upload failed", -5);
        NCHK(hipStreamSynchronize(nullptr));        // creation-time only
        h.built = true;
    }
    if (reverse >= 0 && reverse != h.reverse) {
        // which pass feeds which head: ops 0 (proj.0) and 3 (proj_t.0) read the backbone's rows
        const int xa = reverse ? h.spec.x_proxy : h.spec.x_real, wa = reverse ? h.spec.xw_proxy : h.spec.xw_real;
        const int xb = reverse ? h.spec.x_real : h.spec.x_proxy, wb = reverse ? h.spec.xw_real : h.spec.xw_proxy;
        h.ops[0].x[0] = xa; h.ops[0].xw[0] = wa; h.ops[3].x[0] = xb; h.ops[3].xw[0] = wb;
        h.reverse = reverse;
    }
    return 0;
}

int GNet::head_bind(const char* name_, float* p, float* m, float* v) {
    if (!name_ || !p) return fail("ptta_head_bind: missing argument", -22);
    NRUN(head_build(-1));
    HeadTrain& h = head;
    const std::string name(name_);
    auto it = h.aid.find(name);
    if (it != h.aid.end()) {
        if (!m || !v) return fail("Adam moments of " + name + " are required", -22);
        Adapted& a = h.adapted[it->second];
        a.p = p; a.m = m; a.v = v; h.adam_dirty = h.etab_dirty = true; h.fwd_ok = h.bwd_ok = false;
        return 0;
    }
    for (auto& t : h.tgt) if (t.name == name) { t.p = p; h.etab_dirty = true; return 0; }
    return fail("not a head parameter: " + name, -2);
}

int GNet::head_set_hparams(float lr, float b1, float b2, float eps, float wd, float tau, int adam_step, hipStream_t s) {
    NRUN(head_build(-1));
    const float hy[5] = {lr, b1, b2, eps, wd};
    const float t2[2] = {tau, (float)(1.0 - (double)tau)};
    if (ptta_launch_set_floats(head.hyper, hy, 5, s) || ptta_launch_set_floats(head.tau2, t2, 2, s)) return fail("head hyper-parameter upload failed", -5);
    if (adam_step >= 0 && ptta_launch_set_int(head.step_dev, adam_step, s)) return fail("set step failed", -5);
    return 0;
}

// the engine's own packed copies of the head tensors (what ptta_load_weights derived for the TTA program) from the bound parameters
int GNet::head_sync_packed(bool targets_too, hipStream_t s) {
    HeadTrain& h = head;
    auto put = [&](const std::string& name, const float* p, long n) -> int {
        if (!p) return 0;
        const size_t dot = name.rfind('.');
        const std::string base = name.substr(0, dot), leaf = name.substr(dot + 1);
        auto cv = convs.find(base);
        if (cv != convs.end()) {
            if (leaf == "weight") { pack_conv_weight(cv->second, p, s); cv->second.loaded = true; }
            else { NCHK(hipMemcpyAsync(cv->second.bias, p, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s)); cv->second.has_bias = true; }
            return 0;
        }
        auto fb = frozen_bn.find(base);
        if (fb != frozen_bn.end()) NCHK(hipMemcpyAsync(leaf == "weight" ? fb->second.first : fb->second.second, p, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s));
        return 0;
    };
    for (auto& a : h.adapted) NRUN(put(a.name, a.p, a.n));
    if (targets_too) for (auto& t : h.tgt) NRUN(put(t.name, t.p, t.n));
    for (Op& o : ops) o.st_eval_valid = false;
    return 0;
}

int GNet::head_reload(hipStream_t s) {
    NRUN(head_build(-1));
    drop_graphs();
    return head_sync_packed(true, s);
}

int GNet::head_forward(const float* image, const float* sparse, int reverse, float* emb_out, float* ref_out, hipStream_t s) {
    NRUN(head_build(reverse ? 1 : 0));
    HeadTrain& h = head;
    for (auto& a : h.adapted) if (!a.p || !a.m || !a.v) return fail("head parameter not bound: " + a.name, -3);
    int nt = 0; for (auto& t : h.tgt) nt += t.p ? 1 : 0;
    if (nt != 0 && nt != 6) return fail("bind all six proj_t parameters or none", -3);
    h.bwd_ok = false;
    if (nt == 6) {             // _update_head(): BEFORE the heads run
        if (h.etab_dirty) {
            h.etab_host.resize(6); long off = 0;
            for (int k = 0; k < 6; ++k) { h.etab_host[k] = PttaAdamEntry{h.tgt[k].p, nullptr, nullptr, h.adapted[k].p, h.tgt[k].n, off}; off += h.tgt[k].n; }
            h.etab_total = off;
            NCHK(hipMemcpyAsync(h.etab, h.etab_host.data(), 6 * sizeof(PttaAdamEntry), hipMemcpyHostToDevice, s));
            h.etab_dirty = false;
        }
        if (ptta_launch_ema_multi(h.etab, 6, h.etab_total, h.tau2, s)) return fail("EMA launch failed", -5);
        for (auto& t : h.tgt) {       // proj_t's packed copies follow its bound tensors
            const size_t dot = t.name.rfind('.');
            const std::string base = t.name.substr(0, dot), leaf = t.name.substr(dot + 1);
            auto cv = convs.find(base);
            if (cv != convs.end()) {
                if (leaf == "weight") pack_conv_weight(cv->second, t.p, s);
                else NCHK(hipMemcpyAsync(cv->second.bias, t.p, (size_t)t.n * sizeof(float), hipMemcpyDeviceToDevice, s));
            } else {
                auto fb = frozen_bn.find(base);
                if (fb != frozen_bn.end()) NCHK(hipMemcpyAsync(leaf == "weight" ? fb->second.first : fb->second.second, t.p, (size_t)t.n * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
        }
    }
    // the heads' BatchNorm1d running statistics are the TTA program's bound buffers
    for (Op& ho : h.ops) {
        if (ho.kind != K_BN) continue;
        for (const Op& o : ops) if (o.kind == K_BN && o.bname == ho.bname) { ho.rm = o.rm; ho.rv = o.rv; ho.nbt = o.nbt; }
    }
    drop_graphs();
    bn_prepare = 1;
    const int rcf = head_features(image, sparse, s);
    bn_prepare = 0;
    fwd_valid = false;                   // not a forward ptta_backward may follow
    if (rcf) return rcf;
    {
        ProgSwap sw(this, &h);
        repack_adapted(s);
        const int rc = run_ops_fwd(true, s);
        if (rc) return rc;
    }
    const size_t eb = (size_t)rows() * h.spec.out * sizeof(float);
    if (emb_out) NCHK(hipMemcpyAsync(emb_out, T[h.t_emb].p, eb, hipMemcpyDeviceToDevice, s));
    if (ref_out) NCHK(hipMemcpyAsync(ref_out, T[h.t_ref].p, eb, hipMemcpyDeviceToDevice, s));
    h.fwd_ok = true;
    return 0;
}

int GNet::head_backward(float* loss_out, hipStream_t s) {
    HeadTrain& h = head;
    if (!h.built || !h.fwd_ok) return fail("ptta_head_backward needs the activations of the last ptta_head_forward", -3);
    if (ptta_launch_prepare_loss(T[h.t_emb].p, T[h.t_ref].p, rows(), h.spec.out, T[h.t_emb].g, h.loss_part, h.loss, s)) return fail("prepare loss failed", -5);
    if (loss_out) NCHK(hipMemcpyAsync(loss_out, h.loss, sizeof(float), hipMemcpyDeviceToDevice, s));
    {
        ProgSwap sw(this, &h);
        const int rc = run_ops_bwd(s);
        if (rc) return rc;
    }
    h.bwd_ok = true;
    return 0;
}

int GNet::head_adam_step(hipStream_t s) {
    HeadTrain& h = head;
    if (!h.built || !h.bwd_ok) return fail("ptta_head_adam_step needs the gradients of ptta_head_backward", -3);
    if (h.adam_dirty) {
        h.adam_host.resize(12); long off = 0;
        for (int k = 0; k < 12; ++k) { const Adapted& a = h.adapted[k]; h.adam_host[k] = PttaAdamEntry{a.p, a.m, a.v, h.gall + a.goff, a.n, off, 1}; off += a.n; }
        NCHK(hipMemcpyAsync(h.adam_tab, h.adam_host.data(), 12 * sizeof(PttaAdamEntry), hipMemcpyHostToDevice, s));
        h.adam_dirty = false;
    }
    if (ptta_launch_adam_multi(h.adam_tab, 12, h.gall_n, h.hyper, h.step_dev, h.ticket, s)) return fail("adam failed", -5);
    drop_graphs();
    NRUN(head_sync_packed(false, s));                  // TTA calls on this handle see the trained heads
    h.bwd_ok = false;
    return 0;
}

int GNet::head_get_grad(const char* name, float* dst, int64_t capacity, int* has_grad_host, hipStream_t s) {
    HeadTrain& h = head;
    if (!h.built) return fail("no head parameter bound", -3);
    auto it = h.aid.find(name ? name : "");
    if (it == h.aid.end()) return fail(std::string("not a head parameter: ") + (name ? name : "(null)"), -2);
    const Adapted& a = h.adapted[it->second];
    if (has_grad_host) *has_grad_host = 1;             // proj and pred train in both directions on these backbones
    if (dst) {
        if (capacity < a.n) return fail("capacity too small", -22);
        NCHK(hipMemcpyAsync(dst, h.gall + a.goff, (size_t)a.n * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return 0;
}
