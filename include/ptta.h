/* libptta_hip — C ABI of the MI355X-native ProxyTTA test-time-adaptation step (MSG_CHN, NLSPN and CostDCNet backbones).
 *
 * This is the drop-in boundary for the hot path of seobbro/TTA-depth-completion.  The reference
 * has no FFI for this path (it is Python calling ATen); each entry point below names the Python
 * call(s) it replaces, file:line relative to the reference root.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host; tensors are the reference's
 *     own layouts: images (N,3,H,W), depth maps (N,1,H,W), embeddings (rows,512), all fp32,
 *     contiguous.  Weights are passed as the reference's state_dict tensors (fp32, NCHW) and are
 *     re-packed internally.
 *   - all work is enqueued on the hipStream_t argument (pass torch.cuda.current_stream()); nothing
 *     is allocated after ptta_create and no call on the path synchronises (host constants travel as
 *     kernel arguments).  The only calls that wait for the stream are the ones that RETURN a host
 *     value (ptta_get_adam_step, ptta_profile_read), the ones that destroy a captured graph
 *     (option "graph" = 1 only; they wait for its last replay: ptta_load_weights, ptta_bind_adapted, ptta_set_image_norm,
 *     ptta_set_graph(0), a changed max_input_depth in ptta_set_hparams), ptta_step_pipelined when it
 *     is handed a frame it was NOT told about by the previous call (it waits for its own prefix
 *     stream once before computing that frame's prefix in line; an announced frame costs no wait),
 *     and any other entry point called while a prefix is in flight (it waits for the prefix stream).
 *   - return value 0 = ok, <0 = error; ptta_last_error() returns a message.  Nothing is printed.
 *   - one handle per (process, GPU); a handle is not thread-safe.
 */
#ifndef PTTA_H
#define PTTA_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ptta_ctx* ptta_handle;
typedef void* ptta_stream;              /* hipStream_t */

/* PTTA_BACKBONE_NLSPN: ExternalModel_Adapt(model_name='nlspn') -- src/nlspn_model_adapt.py:13-128 ->
 * NLSPNModel_Adapt._rgbd_meta_contrast (external_src/NLSPN/src/model/nlspnmodel_adapt.py:850-944), adapter settings of
 * src/nlspn_model_adapt.py:56-68, adapt_parameters('meta_bn') (:322-337).  meta_mode must be PTTA_META_1LAYER (optionally | PTTA_NLSPN_LEGACY_OFFSET), dtype
 * PTTA_DTYPE_F32, any height and width >= 16 (sizes that are not multiples of 16 follow the reference's ceil-halving encoder
 * maps and decoder crops, nlspnmodel_adapt.py:474-490).  Differences from the MSG_CHN handle: embeddings are (rows, 1024)
 * with rows = n*(H/16)*(W/16); ptta_adapted_count() is 88 (conv1_rgb_meta + every BatchNorm2d weight/bias, in the
 * reference's order) and every one of them must be bound; ptta_load_weights ignores BatchNorm running statistics
 * (dropped by 'meta_bn') and the values of adapted tensors (the bound tensors are read instead);
 * ptta_backward ignores its two output pointers (read the 88 gradients with ptta_get_grad) and ptta_adam_step takes
 * NULL gradients (it uses the internal ones); after ptta_set_graph(h, 1) (or PTTA_GRAPH=1) ptta_step and ptta_forward_eval replay
 * captured hipGraphs from their second call on (default off for these handles: measured no faster than plain launches);
 * ptta_profile returns -38. */
/* PTTA_BACKBONE_COSTDCNET: ExternalModel_Adapt(model_name='costdcnet') -- src/costdcnet_model_adapt.py:31-114 ->
 * CostDCNet._rgbd_meta_contrast (external_src/costdcnet/CostDCNet_adapt.py:207-256), res = 16 planes, up_scale = 4
 * (src/costdcnet_model_adapt.py:47-52), adapt_parameters('meta_bn') (:357-378).  meta_mode PTTA_META_1LAYER, dtype
 * PTTA_DTYPE_F32, hp.max_predict_depth > 0 (the far plane).  Shapes not divisible by 16 run the reference's dual-corner
 * padding (:134-210).  Embeddings are (rows, 512) with rows = n'*(H'/32)*(W'/32) over the padded batch / size;
 * ptta_adapted_count() is 32 (conv1_rgb_meta + every BatchNorm2d weight/bias of Encoder2D, the reference's order);
 * ptta_load_weights BINDS the running statistics of BatchNorm3d / BatchNorm1d / the sparse encoder's BatchNorm
 * (updated in place by training forwards, read by the eval forward) and ignores those of BatchNorm2d (dropped by
 * 'meta_bn'); the other remarks of PTTA_BACKBONE_NLSPN apply.
 * meta_mode | PTTA_SYNCBN_ADAPT: the adapted set of the reference's DDP run (src/tta_main.py:326 convert_syncbn() BEFORE :339
 * adapt_parameters): EVERY BatchNorm of the model -- Encoder2D's, the BatchNorm1d inside the sparse encoder's nine MinkowskiBatchNorm
 * (`enc3d.<layer>.bn.weight/bias`), UNet3D's 28 BatchNorm3d, the heads' three BatchNorm1d -- is adapted and has lost its running
 * statistics (batch statistics in train AND eval mode; ptta_load_weights ignores every running_* key).  ptta_adapted_count() is 112
 * unique tensors in the reference's order; the reference's list has 116 entries because enc2d.layer{2,3}.0.norm3.{weight,bias} sit
 * behind two SyncBatchNorm modules (norm3 and downsample[1]) and are listed -- and updated by Adam -- twice per step:
 * ptta_adapted_repeat() is 2 for them, 0 for proj.1 / pred.1 (listed, never given a gradient), 1 otherwise. */
enum { PTTA_BACKBONE_MSG_CHN = 0, PTTA_BACKBONE_NLSPN = 1, PTTA_BACKBONE_COSTDCNET = 2 };
/* Storage / arithmetic of the MSG_CHN step.
 * PTTA_DTYPE_F32: every 32-channel map fp32, every product fp32-faithful on the bf16 matrix cores (bf16x3 split, fp32 accumulate).
 * PTTA_DTYPE_MIXED (BASELINE config 2; replaces the all-bf16 mode of rounds 1-4, which missed the tolerance by 9x and is gone): the REAL
 *   frames' forward -- the tensors the scored depth map is made of -- exactly as PTTA_DTYPE_F32; the tensors the reference computes under
 *   no_grad or detaches (the zero-image proxy pass, network_exp_msg_chn_adapt.py:509-532, the embedding branch :551-554) and the data
 *   gradients of loss.backward() (src/tta_main.py:632), which reach the scored depth only through an lr-sized Adam move, are NARROW: bf16
 *   storage, one bf16 MFMA per product, fp32 accumulate -- except the WEIGHT operand of the data gradients, which stays bf16 hi + lo (two
 *   MFMAs, option "bwd_w2"): a rounded weight is a systematic error of the gradient's direction and showed over 100+ steps.  depth_train is bit-identical to PTTA_DTYPE_F32; measured budget per tensor
 *   class: profiles/r05_precision_budget.txt; bounds per mode: tests/test_gpu_mixed.py.  PTTA_ARITH=exact / PTTA_CONV_IMPL=naive are
 *   PTTA_DTYPE_F32-only (ptta_create returns -38 with PTTA_DTYPE_MIXED). */
/* PTTA_BACKBONE_NLSPN with PTTA_DTYPE_MIXED: the generic engine keeps fp32 storage and switches the matrix-core convolutions of the proxy
 * frames to one bf16 MFMA per product and those of the data gradients to two (hi activations x hi + lo weights; PTTA_MIXED_BWD_ROUNDED_W:
 * one, round 5's form -- scored depth 2.0e-4 after config 3's three steps and growing 6e-5 per step with it).
 * PTTA_BACKBONE_COSTDCNET refuses it (-38): its scored depth leaves the tolerance (profiles/r06_nlspn_costdcnet_mixed.txt). */
enum { PTTA_DTYPE_F32 = 0, PTTA_DTYPE_MIXED = 1,
       /* OR-ed into PTTA_DTYPE_MIXED (precision budget, tools/accuracy_report.py --keep): keep one of the three narrow classes at fp32 storage
        * and bf16x3 arithmetic -- the proxy chain, the data gradients, the heads */
       PTTA_MIXED_KEEP_PROXY = 0x100, PTTA_MIXED_KEEP_BACKWARD = 0x200, PTTA_MIXED_KEEP_HEADS = 0x400,
       /* generic engine, comparison only: the data gradients on bf16-ROUNDED weights, one MFMA per product (round 5's mixed mode) */
       PTTA_MIXED_BWD_ROUNDED_W = 0x800 };
enum { PTTA_META_1LAYER = 0, PTTA_META_2LAYERS = 1 };   /* conv1_rgb_meta = Conv2d(32,32,3) | Res_Conv(32,128) */
/* NLSPN only, OR-ed into meta_mode: ExternalModel_Adapt(..., offset=True) -> args.legacy (src/nlspn_model_adapt.py:62):
 * the confidence gathers add each tap's own (dy, dx) to the learned offset (nlspnmodel_adapt.py:297-302).
 * src/tta_main.py:309-317 always constructs the model this way. */
enum { PTTA_NLSPN_LEGACY_OFFSET = 0x100,
       /* NLSPN, OR-ed into meta_mode: the adapted list of the reference's DDP run.  tta_main.py:326 converts every BatchNorm to
        * SyncBatchNorm BEFORE adapt_parameters('meta_bn') (:339), whose isinstance test (src/nlspn_model_adapt.py:329-331) then
        * also matches the heads' three BatchNorm1d: 94 adapted tensors (proj.1, proj_t.1, pred.1 weight/bias appended in
        * module order) instead of 88.  proj / pred feed the detached embedding: their gradients stay zero. */
       PTTA_NLSPN_SYNCBN_ADAPT = 0x200,
       PTTA_SYNCBN_ADAPT = 0x200 };        /* the same switch for PTTA_BACKBONE_COSTDCNET (see there) */

/* Hyper-parameters of the step.  Reference: src/tta.py:10-160 flags learning_rates,
 * optimizer_betas, optimizer_epsilon, w_weight_decay, w_loss_sparse_depth, w_loss_smoothness,
 * w_loss_cos, max_input_depth (forwarded positionally to src/tta_main.py:23-100). */
typedef struct {
    float lr, beta1, beta2, eps, weight_decay;
    float w_sparse_depth, w_smoothness, w_cos;
    float max_input_depth;              /* < 0: no clamp (max_input_depth=None) */
    float max_predict_depth;            /* CostDCNet only: ExternalModel_Adapt(max_predict_depth) = far plane of the cost volume */
} ptta_hparams;

/* ExternalModel_Adapt(model_name='msg_chn', ...) + _prepare_head(prepare_mode)
 * (src/external_model_adapt.py:46-80, :561; network_exp_msg_chn_adapt.py:1022-1087).
 * n, height, width: batch and frame size of every later call.  Shapes not divisible by 16 run
 * the reference's dual-corner padding (src/msg_chn_model_adapt.py:60-123) internally. */
int ptta_create(ptta_handle* out, int backbone, int meta_mode, int n, int height, int width, int dtype,
                const ptta_hparams* hp_host);
void ptta_destroy(ptta_handle h);
const char* ptta_last_error(ptta_handle h);

/* optimizer.param_groups[i]['lr'] = ... (src/tta_main.py:507-512) and the loss weights. */
int ptta_set_hparams(ptta_handle h, const ptta_hparams* hp_host, ptta_stream s);

/* restore_model -> load_state_dict(checkpoint['net']) (src/msg_chn_model_adapt.py:482-501), one
 * call per state_dict key.  Frozen tensors are copied and re-packed; BatchNorm buffers
 * (*.running_mean / *.running_var / *.num_batches_tracked) are BOUND: the pointer is kept and
 * updated in place by every training-mode forward, as nn.BatchNorm1d does.  Unknown keys: error. */
int ptta_load_weights(ptta_handle h, const char* name, const void* tensor, const int64_t* shape, int ndim,
                      ptta_stream s);

/* adapt_parameters(mode='meta') + torch.optim.Adam state (src/msg_chn_model_adapt.py:392-396,
 * src/tta_main.py:339-346).  The caller keeps ownership of the parameter and of the Adam moments
 * exp_avg / exp_avg_sq (fp32, same shape); the library reads the parameter on every forward and
 * ptta_step / ptta_adam_step update all three in place.  name: one of ptta_adapted_name(). */
int ptta_bind_adapted(ptta_handle h, const char* name, float* param, float* exp_avg, float* exp_avg_sq);
int ptta_set_adam_step(ptta_handle h, int step, ptta_stream s);      /* optimizer.state[p]['step'] */
int ptta_get_adam_step(ptta_handle h, int* step_host, ptta_stream s); /* synchronises s */

/* model.train(); model.forward(image, sparse_depth, loss_type='adapt_meta_selfsup_seq_ema_reverse')
 * (src/tta_main.py:610 -> external_model_adapt.py:82-114 -> msg_chn_model_adapt.py:54-125 ->
 * network_exp_msg_chn_adapt.py:463-557).  Outputs: depth (N,1,H,W), emb/ref (rows,512) with
 * rows = ptta_embedding_rows(h).  Activations needed by ptta_backward stay in the handle. */
int ptta_forward_train(ptta_handle h, const float* image, const float* sparse_depth,
                       float* depth_out, float* emb_out, float* ref_out, ptta_stream s);
int64_t ptta_embedding_rows(ptta_handle h);

/* model.eval(); with no_grad: model.forward(...) (src/tta_main.py:729-736). */
int ptta_forward_eval(ptta_handle h, const float* image, const float* sparse_depth, float* depth_out, ptta_stream s);

/* model.compute_loss(..., loss_type='adapt') (src/external_model_adapt.py:119-203, :371-441).
 * loss_info_out[4] (device) = {loss, loss_smooth, loss_sparse_depth, loss_cos}.  emb/ref may be
 * NULL (loss_cos = 0).  The backward half writes dL/d(depth) (N,1,H,W) and dL/d(ref) (rows,512). */
int ptta_loss_forward(ptta_handle h, const float* loss_image, const float* depth, const float* sparse_depth,
                      const float* validity, const float* emb, const float* ref, int64_t rows,
                      float w_sparse_depth, float w_smoothness, float w_cos, float* loss_info_out, ptta_stream s);
int ptta_loss_backward(ptta_handle h, const float* loss_image, const float* depth, const float* sparse_depth,
                       const float* validity, const float* emb, const float* ref, int64_t rows,
                       float* grad_depth_out, float* grad_ref_out, ptta_stream s);

/* loss.backward() restricted to the adapted parameters (src/tta_main.py:632): consumes
 * dL/d(depth) and dL/d(ref) of the last ptta_forward_train and writes the gradients of
 * conv1_rgb_meta.{weight (32,32,3,3), bias (32)}. */
int ptta_backward(ptta_handle h, const float* grad_depth, const float* grad_ref,
                  float* grad_meta_weight_out, float* grad_meta_bias_out, ptta_stream s);

/* Gradient of one adapted parameter after ptta_backward / ptta_step, by state_dict key (the
 * 2layers meta layer has seven adapted tensors: two conv weights, one conv bias, two BatchNorm
 * gamma/beta pairs).  ptta_adapted_count/name enumerate them in state_dict order. */
int ptta_get_grad(ptta_handle h, const char* name, float* dst, int64_t capacity, ptta_stream s);
/* Overwrite the stored gradient of one adapted parameter (e.g. with the mean over ranks that DistributedDataParallel
 * would have produced, src/msg_chn_model_adapt.py:476-480) before ptta_adam_step(h, NULL, NULL, s). */
int ptta_set_grad(ptta_handle h, const char* name, const float* src, int64_t numel, ptta_stream s);
int ptta_adapted_count(ptta_handle h);
const char* ptta_adapted_name(ptta_handle h, int index, int64_t* numel_host);
/* How many times the reference's parameter list names this tensor = the Adam updates it receives per optimizer.step()
 * (torch.optim.Adam walks the list; a tensor listed twice is stepped twice with the same gradient).  1 except under PTTA_SYNCBN_ADAPT. */
int ptta_adapted_repeat(ptta_handle h, int index);

/* ptta_step for a STREAM of frames (MSG_CHN handles; others forward to ptta_step): besides the step on (image, sparse) it starts, on a stream
 * of its own, the part of the NEXT frame's forward that does not depend on the adapted parameters -- clamp / pooling of the sparse depth, the
 * frozen RGB encoder, the depth-only head of the stage-1 encoder (everything upstream of conv1_rgb_meta) -- so that it runs beside this
 * frame's step instead of at the head of the next one.  Results are those of ptta_step, call by call (the prefix's outputs exist twice).
 * Frames are named by TOKENS the caller chooses: non-zero, and a NEW one whenever the frame CONTENT is new (a counter).  next_image /
 * next_sparse / next_token: the frame the FOLLOWING call will pass as (image, sparse, frame_token); it is copied into the handle when it is
 * announced, so the caller's buffers only have to hold it until this call's work on the prefix stream has run.  The prepared prefix is used
 * iff the following call's frame_token equals the announced next_token -- never by pointer identity: refilling one staging buffer with another
 * frame is safe as long as the new content gets a new token (tests/test_gpu_staging_augment.py).  next_token == frame_token: another step
 * on the same frame (inner_iter > 1), its prefix is kept.  A token of 0 = unnamed: frame_token 0 never matches (prefix computed in line),
 * next_token 0 / NULL pointers prepare nothing.  The reference's loop knows its next frame from the data loader (src/tta_main.py:519-523).
 * ptta_forward_eval / ptta_forward_train between two calls are fine (they run in, and invalidate, the buffer set of the last processed frame).
 * Host synchronisation: none when the frame was announced; an UNANNOUNCED frame (first call, token mismatch) waits for the prefix stream
 * on the host once before computing its prefix in line. */
int ptta_step_pipelined(ptta_handle h, const float* image, const float* loss_image, const float* sparse_depth, const float* validity_map,
                        uint64_t frame_token, const float* next_image, const float* next_sparse_depth, uint64_t next_token,
                        float* depth_out, float* loss_info_out, ptta_stream s);
/* The stream the next frame's prefix runs on: when that frame is still arriving (an asynchronous H2D copy), make THIS stream wait for the
 * copy's event before the call that announces the frame -- not the caller's stream, which would delay the current step. */
int ptta_pipeline_stream(ptta_handle h, ptta_stream* stream_out);
/* The scored eval forward (src/tta_main.py:729-736) of the frame the last ptta_step_pipelined call adapted, without recomputing that
 * frame's parameter-independent prefix (it is still in the handle).  When that call ran as a plain ptta_step (profiling, validation kernels,
 * SyncBatchNorm / gradient exchange, padded sizes) this is a full ptta_forward_eval of its frame, read from the caller's
 * buffers of that call (which must still hold it).  -3 when no such frame is held (e.g. after ptta_load_weights or a ptta_forward_eval). */
int ptta_forward_eval_last(ptta_handle h, float* depth_out, ptta_stream s);

/* optimizer.step() for the bound parameters with explicit gradients (src/tta_main.py:633). */
int ptta_adam_step(ptta_handle h, const float* grad_meta_weight, const float* grad_meta_bias, ptta_stream s);

/* One whole TTA step = src/tta_main.py:610-633: forward_train + compute_loss + zero_grad +
 * backward + optimizer.step, in one enqueue.  `image` feeds the network, `loss_image` the
 * smoothness weights (tta_main.py:610 vs :620; may be the same pointer).  validity NULL =>
 * where(sparse>0,1,sparse) (tta_main.py:583-586).  Optional outputs may be NULL. */
int ptta_step(ptta_handle h, const float* image, const float* loss_image, const float* sparse_depth,
              const float* validity, float* depth_out, float* loss_info_out, ptta_stream s);

/* OutlierRemoval(kernel_size, threshold).remove_outliers(sparse_depth, validity_map) (src/net_utils.py:750-811),
 * the on-device filter src/tta_main.py:590 applies before every forward.  Handle-free; scratch = 4 KiB of
 * device memory (1024 floats).  Outputs may not alias the inputs. */
int ptta_outlier_removal(const float* sparse_depth, const float* validity, float* sparse_out, float* validity_out,
                         int n, int height, int width, int kernel_size, float threshold, float* scratch, ptta_stream s);

/* Validation metrics of src/tta_main.py:779-798 (eval_utils.py:117-174) reduced on device:
 * metrics_out[4] (device) = {MAE mm, RMSE mm, iMAE 1/km, iRMSE 1/km} over pixels with ground_truth > 0 and
 * min <= ground_truth <= max.  scratch: 16 KiB of device memory.  Handle-free. */
int ptta_eval_metrics(const float* depth, const float* ground_truth, int64_t numel, float min_evaluate_depth,
                      float max_evaluate_depth, void* scratch, float* metrics_out, ptta_stream s);

/* The reference's native extension `DCN` (external_src/NLSPN/src/model/deformconv/src/vision.cpp:7-12):
 *   modulated_deform_conv_forward(input, weight, bias, offset, mask, kh,kw, sh,sw, ph,pw, dh,dw, group,
 *                                 deformable_group, im2col_step) -> output           (modulated_deform_conv.h:10-26)
 *   modulated_deform_conv_backward(..., grad_output, ...) -> [grad_input, grad_offset, grad_mask, grad_weight, grad_bias]
 *                                                                                     (modulated_deform_conv.h:46-63)
 * NCHW fp32 contiguous, offset (B, 2*K*dg, Ho, Wo), mask (B, K*dg, Ho, Wo).  im2col_step has no meaning here (no
 * column buffer).  bias may be NULL; any grad_* pointer may be NULL.  Handle-free. */
int ptta_mdconv_forward(const float* input, const float* weight, const float* bias, const float* offset, const float* mask,
                        float* output, int b, int c, int h, int w, int c_out, int kh, int kw, int sh, int sw, int ph, int pw,
                        int dh, int dw, int group, int deformable_group, ptta_stream s);
int ptta_mdconv_backward(const float* input, const float* weight, const float* bias, const float* offset, const float* mask,
                         const float* grad_output, float* grad_input, float* grad_offset, float* grad_mask, float* grad_weight,
                         float* grad_bias, int b, int c, int h, int w, int c_out, int kh, int kw, int sh, int sw, int ph, int pw,
                         int dh, int dw, int group, int deformable_group, ptta_stream s);

/* Photometric normalisation of the network input, fused into the first convolution's loads (and into
 * the dual-corner padding): replaces Transforms.normalize_images (src/transforms.py:668-710) as called at
 * src/tta_main.py:454-464,652 -- image -> (image / divisor - mean[c]) / std[c].  [0,1] is (255, 0, 1);
 * [-1,1] is (255, .5, .5); standard normalisation is (255, mean, std).  After this call `image` in
 * ptta_forward_train / ptta_forward_eval / ptta_step is the RAW image; the loss keeps using the raw image
 * unless a separate loss_image is passed, which is what the reference does (tta_main.py:610,620).
 * mean / std may be NULL (0 / 1).  (1, 0, 1) switches it off. */
int ptta_set_image_norm(ptta_handle h, float divisor, const float* mean, const float* stdv);

/* ---- Stage-2 head trainer ------------------------------------------------------------------------------
 * One step of src/head_main.py:464-480 for MSG_CHN (fp32 handle, sizes divisible by 16) with loss_type
 * 'head_selfsup_seq_ema_reverse' (reverse = 1) or 'head_selfsup_seq_ema' (reverse = 0), i.e.
 * network_exp_msg_chn_adapt.py:610-699 (`_rgbd_meta_contrast_head`, mode without 'adapt'):
 *   _update_head()           proj_t <- tau proj_t + (1 - tau) proj over parameters()                    (:701-703)
 *   both backbone passes     no gradient, train-mode BatchNorm, stop at depth_encoder3                      (:626-676)
 *   reverse:     emb = pred(proj(feat_zero).detach()), ref = proj(feat).detach()      -> pred trains      (:691-694)
 *   not reverse: emb = pred(proj(feat)),               ref = proj(feat_zero).detach() -> proj, pred train (:681-684)
 *   prepare_loss             mean(2 - 2 <normalize(emb), normalize(ref)>)          (src/external_model_adapt.py:524-541)
 *   Adam                     over prepare_parameters('head_selfsup_ema') (src/msg_chn_model_adapt.py:297-304); parameters
 *                            without a gradient are skipped like torch.optim.Adam skips .grad == None
 * NLSPN and CostDCNet handles (csrc/ghead.hip; nlspnmodel_adapt.py:1014-1060, CostDCNet_adapt.py:258-303): the same calls.  There the reference branch
 * goes through the EMA target -- not reverse: emb = pred(proj(rows(real).detach())), ref = proj_t(rows(zero image)).detach(); reverse: the passes
 * swapped -- so proj AND pred train in both directions (ptta_head_get_grad: has_grad = 1 for all twelve), proj_t's BatchNorm1d runs in train mode
 * like the other two, and the backbone's BatchNorm2d layers normalise with their LOADED running statistics (`train(prepare=True)`,
 * src/nlspn_model_adapt.py:360-368, src/costdcnet_model_adapt.py:418-430) -- ptta_load_weights keeps them for this purpose; BatchNorm3d and the
 * sparse encoder's BatchNorm1d stay in train mode as in the reference.  Not offered (-38) with PTTA_SYNCBN_ADAPT, a statistics exchange bound,
 * or (CostDCNet) sizes that need the dual-corner padding.
 * ptta_head_bind: name in {proj,pred}.{0,3}.{weight,bias}, {proj,pred}.1.{weight,bias} (param + both Adam moments, device fp32,
 * caller-owned, updated in place) or proj_t.* (param only; bind all six or none).  After ptta_head_adam_step the handle's
 * packed copies of the updated head weights are re-derived, so TTA calls on the same handle see them; after an EXTERNAL
 * update of bound parameters call ptta_head_reload.  adam_step < 0 in ptta_head_set_hparams keeps the device-side count.
 * ptta_head_step = forward + backward + adam_step.  loss_out: 1 device float.  Everything enqueues only. */
int ptta_head_bind(ptta_handle h, const char* name, float* param, float* exp_avg, float* exp_avg_sq);
int ptta_head_set_hparams(ptta_handle h, float lr, float beta1, float beta2, float eps, float weight_decay, float tau,
                          int adam_step, ptta_stream s);
int ptta_head_reload(ptta_handle h, ptta_stream s);
int ptta_head_forward(ptta_handle h, const float* image, const float* sparse_depth, int reverse, float* emb_out, float* ref_out,
                      ptta_stream s);
int ptta_head_backward(ptta_handle h, float* loss_out, ptta_stream s);
int ptta_head_adam_step(ptta_handle h, ptta_stream s);
int ptta_head_step(ptta_handle h, const float* image, const float* sparse_depth, int reverse, float* loss_out, ptta_stream s);
/* gradient of the last ptta_head_backward; *has_grad_host = 0 (and dst untouched) for parameters outside the graph */
int ptta_head_get_grad(ptta_handle h, const char* name, float* dst, int64_t capacity, int* has_grad_host, ptta_stream s);

/* Geometric augmentation on device: Transforms.crop + horizontal_flip + vertical_flip
 * (src/transforms.py:337-407, 955-1034) in one pass over an N x C x H x W fp32 tensor:
 *   dst[b,c,y,x] = src[b, c, start_y[b] + (vflip[b] ? ch-1-y : y), start_x[b] + (hflip[b] ? cw-1-x : x)]
 * (the reference crops first and flips the cropped sample).  start_y / start_x (int32, device, per sample; NULL = 0) and
 * hflip / vflip (uint8, device, per sample; NULL = no flip) are the caller's random draws; starts are clamped into the
 * sample.  dst is N x C x crop_height x crop_width and must not alias src.  Handle-free, enqueues only. */
int ptta_crop_flip(const float* src, float* dst, int n, int channels, int height, int width, int crop_height, int crop_width,
                   const int32_t* start_y, const int32_t* start_x, const uint8_t* hflip, const uint8_t* vflip, ptta_stream s);

/* The other augmentations the adapt scripts enable (bash/adapt/adapt_msgchn_vkitti.sh:34-41; applied in every step,
 * src/tta_main.py:595-605).  The reference implements them with torchvision.transforms.functional 0.10.1 (third party, absent
 * from the reference tree: parity unpinned); each call restates torchvision's tensor algorithm.  All handle-free, enqueue only,
 * per-sample decisions as device arrays (uint8 flags; a sample whose flag is 0 is copied), dst must not alias src.
 *   ptta_rotate         Transforms.rotate (src/transforms.py:1036-1070): functional.rotate(image, angle[b] degrees, interpolation,
 *                       expand=False): affine grid about the image centre + grid_sample(align_corners=False, zeros);
 *                       bilinear = 0 nearest (depth, validity, ground truth), 1 bilinear (image)  (src/tta_main.py:471-473)
 *   ptta_resize_crop    Transforms.resize_and_crop (:1222-1283): functional.resize to (resize_height[b], resize_width[b]) then the crop
 *                       [start_y[b] : +height, start_x[b] : +width] back to the input size; scale_depth = resize_scaling_depth
 *   ptta_photometric    brightness -> contrast -> saturation (:714-838, :236-311) on uint8-valued 3-channel images (the reference casts
 *                       float images to uint8 first); any do_* pointer may be NULL (transform not configured); scratch: 256*n doubles */
int ptta_rotate(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_rotate, const float* angle_deg,
                int bilinear, ptta_stream s);
int ptta_resize_crop(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_resize,
                     const int32_t* resize_height, const int32_t* resize_width, const int32_t* start_y, const int32_t* start_x,
                     int bilinear, int scale_depth, ptta_stream s);
int ptta_photometric(const float* src, float* dst, int n, int height, int width, const uint8_t* do_brightness, const float* f_brightness,
                     const uint8_t* do_contrast, const float* f_contrast, const uint8_t* do_saturation, const float* f_saturation,
                     double* scratch, ptta_stream s);

/* The augmentations of Transforms that NO adapt script enables (src/transforms.py:279-305 gamma / hue, :322-332 noise, :508-625 crop-and-pad /
 * resize-and-pad, :630-655 patch removal).  gamma / hue / pad / resize are torchvision 0.10.1 calls in the reference (parity unpinned,
 * restated); add_noise, remove_random_patches and the crop / pad index arithmetic are the reference's own torch code (pinned:
 * tests/golden/transforms_extra.npz).  Handle-free, enqueue only, per-sample decisions as device arrays, dst must not alias src.
 *   ptta_photometric_full  brightness -> contrast -> gamma -> hue -> saturation (:236-311).  uint8 path when brightness / contrast / hue /
 *                          saturation is configured (non-NULL do_*; :102-106); with gamma alone the images stay float and torchvision's
 *                          float branch applies: (x ** gamma).clamp(0, 1).  scratch: 256*n doubles (not needed for gamma alone)
 *   ptta_add_noise         Transforms.add_noise (:839-876): dst = src + spread * noise (uniform = 0) or src + spread * (noise - 0.5)
 *                          (uniform = 1) where do_noise[b]; `noise` = the caller's torch.randn / torch.rand field, same shape as src
 *   ptta_remove_patches    Transforms.remove_random_patches (:878-924) behind random_nonzero's selection (:926-953): selected = n x H x W
 *                          bytes, 1 at the chosen nonzero pixels; every pixel within the (patch_height[b], patch_width[b]) patch of a
 *                          chosen pixel is zeroed in all channels
 *   ptta_crop_pad          Transforms.crop_and_pad (:1072-1135): image[start_y:end_y, start_x:end_x] padded by pad_top / pad_left (and what
 *                          remains below / right) back to height x width; padding_mode 0 constant (fill), 1 edge, 2 reflect, 3 symmetric
 *   ptta_resize_pad        Transforms.resize_and_pad (:1137-1220): functional.resize to (resize_height[b], resize_width[b]) <= (height,
 *                          width), then padded back to height x width as ptta_crop_pad */
int ptta_photometric_full(const float* src, float* dst, int n, int height, int width, const uint8_t* do_brightness, const float* f_brightness,
                          const uint8_t* do_contrast, const float* f_contrast, const uint8_t* do_gamma, const float* f_gamma,
                          const uint8_t* do_hue, const float* f_hue, const uint8_t* do_saturation, const float* f_saturation,
                          double* scratch, ptta_stream s);
int ptta_add_noise(const float* src, const float* noise, float* dst, int n, int channels, int height, int width, const uint8_t* do_noise,
                   float spread, int uniform, ptta_stream s);
int ptta_remove_patches(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_remove,
                        const uint8_t* selected, const int32_t* patch_height, const int32_t* patch_width, ptta_stream s);
int ptta_crop_pad(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_crop_pad,
                  const int32_t* start_y, const int32_t* start_x, const int32_t* end_y, const int32_t* end_x,
                  const int32_t* pad_top, const int32_t* pad_left, int padding_mode, float fill, ptta_stream s);
int ptta_resize_pad(const float* src, float* dst, int n, int channels, int height, int width, const uint8_t* do_resize_pad,
                    const int32_t* resize_height, const int32_t* resize_width, const int32_t* pad_top, const int32_t* pad_left,
                    int bilinear, int padding_mode, float fill, ptta_stream s);

/* ptta_step enqueues its kernels directly; with ptta_set_option(h, "graph", 1) it replays a captured hipGraph of the whole step.  Graphs are re-captured after any re-binding. */
/* model.convert_syncbn() (src/tta_main.py:326 -> SyncBatchNorm.convert_sync_batchnorm, src/msg_chn_model_adapt.py:547-556)
 * for the one-process-per-GPU run with shared adapted parameters: every training-mode BatchNorm then normalises with
 * the statistics of the GLOBAL batch.  The library collapses a BatchNorm's partial sums into `exchange_buf` (device,
 * float64, caller-owned, `capacity` elements >= 2 * 2 * widest BatchNorm), calls `fn(user, exchange_buf, count, stream)`
 * -- which must SUM the first `count` elements over the ranks in place, ordered on `stream` (torch.distributed.all_reduce
 * on RCCL does exactly that) -- and finalises with world_size x the local row count (equal local batches).  Gradients of
 * adapted BatchNorm parameters come out already averaged over the ranks.  world_size 1 switches the exchange off (and
 * restores graph replay / the second stream).  MSG_CHN handles replay no hipGraph while the CALLBACK form is on. */
typedef int (*ptta_allreduce_fn)(void* user, double* exchange_buf, long long count, ptta_stream s);
int ptta_set_stat_sync(ptta_handle h, ptta_allreduce_fn fn, void* user, double* exchange_buf, int64_t capacity, int world_size);

/* The same exchange on an RCCL communicator owned by the library (the reference's backend: dist.init_process_group('nccl'),
 * src/tta_main.py:101-111): rank 0 calls ptta_rccl_unique_id, the caller broadcasts the 128 bytes, every rank calls
 * ptta_rccl_comm_create (ncclCommInitRank) and ptta_set_stat_sync_rccl.  The library enqueues ncclAllReduce itself on the
 * step's stream: no host callback, and the collectives are captured into the step's hipGraph (MSG_CHN handles keep replaying).
 * comm == NULL switches the exchange off.  ptta_rccl_allreduce_mean_f32 is the ONE gradient collective of a shared-parameter step
 * (the adapted gradients only, instead of DDP's all-reduce of every gradient: src/msg_chn_model_adapt.py:476-480).
 * librccl is resolved at run time (the process's own copy first); -38 when it cannot be loaded; ptta_rccl_last_error() for RCCL errors. */
int ptta_rccl_unique_id(void* id128_host);
int ptta_rccl_comm_create(const void* id128_host, int rank, int world_size, void** comm_out_host);
int ptta_rccl_comm_destroy(void* comm);
int ptta_rccl_allreduce_mean_f32(void* comm, float* buf, int64_t count, ptta_stream s);
const char* ptta_rccl_last_error(void);
int ptta_set_stat_sync_rccl(ptta_handle h, void* comm, double* exchange_buf, int64_t capacity, int world_size);
/* DistributedDataParallel's gradient averaging inside the fused ptta_step (src/tta_main.py:354,631-633): between backward and
 * Adam the adapted gradients are averaged over the communicator with one ncclAllReduce on the step's stream.  NULL: off. */
int ptta_set_grad_sync_rccl(ptta_handle h, void* comm);

int ptta_set_graph(ptta_handle h, int enable);         /* = ptta_set_option(h, "graph", enable) */

/* Per-handle switches -- the library's ONLY run-time configuration besides the ptta_hparams struct: no environment variable is read after
 * ptta_create (the three create-time validation variables are listed at the end of this comment).  Setting an option waits for the
 * handle's streams and drops its captured graphs; the next step re-captures.  Every non-default value is a correct, slower form of
 * the same step, kept as the check of the default (tests/test_gpu_options.py) or taken by the library itself where the default
 * form does not apply (small maps, N > 16, SyncBatchNorm exchange).  Unknown key / value out of range: -22; a key the handle does
 * not have: -38 (the generic engine has "graph" only; PTTA_DTYPE_MIXED handles keep every key but graph / aux_stream / thru / adam_in_wgrad at 1).
 *   key             default  meaning of 0
 *   "graph"         0        (1:) ptta_step / ptta_step_pipelined / ptta_forward_eval replay captured hipGraphs from their second call on.  Default
 *                            since round 5: direct launches on the caller's stream and the handle's own streams -- measured FASTER than replay
 *                            on ROCm 7.2 for all three engines (MSG_CHN pipelined 1.28 vs 1.31 ms, call by call 1.45 vs 1.52: a replayed
 *                            graph starts its second branch late and runs it slower, profiles/r05_step_stamps.txt; host cost 0.46 ms per call)
 *   "aux_stream"    1        one stream: no second queue for the proxy chain / the heads
 *   "thru"          1        the heads' stream joins the main stream before the loss (1: it runs on into loss + head backward)
 *   "adam_in_wgrad" 1        0: Adam is its own launch behind the weight gradient's reduction (1: that reduction applies it -- MSG_CHN 1layer, no
 *                            gradient exchange between the two; one dependent launch less at the end of the step)
 *   "bwd_w2"        1        (PTTA_DTYPE_MIXED) 0: the narrow data-gradient convolutions take bf16-ROUNDED weights, one MFMA per product (round 5's
 *                            form); 1: weights as bf16 hi + lo, two MFMAs per product, gradient maps still bf16 in HBM.  A rounded weight is a
 *                            systematic 2^-9 error of the gradient's direction, the same at every pixel and step: over 150 steps at 256x320 it
 *                            put the scored depth at 1.1e-3 from the reference (3x the reference's own 1-ulp sensitivity); with hi + lo
 *                            weights the mixed mode stays at that floor (profiles/r06_drift.txt).  Not bit-identical to 0 by construction
 *   "fuse_first"    1        every first-layer convolution + the following 32->32 convolution as two launches (2: the RGB branch fused too)
 *   "fuse_head_bwd" 1        the prediction heads' backward as separate launches
 *   "fuse_heads"    1        proj -> pred as separate Linear launches (1: pred.0 o proj.3 folded into one weight at load time)
 *   "heads_v2"      1        proj's hidden layer materialised (1: recomputed inside the GEMM from analytic BatchNorm statistics)
 *   "cos_in_gemm"   1        d loss_cos / d ref written as a tensor (1: formed in the backward GEMM's operand staging)
 *   "mask_bits"     1        fp32 pre-activation maps as ReLU masks (1: one word of sign bits per pixel)
 *   "stamps"        0        (diagnostic, 1:) fourteen one-thread nodes of the step's and the prefix's graphs write wall_clock64() ticks since
 *                            the step's first node into the debug tensor "stamps" -- where the branches of a replayed step start and end with
 *                            no profiler attached (tools/step_stamps.py, profiles/r05_step_stamps.txt); results unchanged
 * Read-only keys (ptta_get_option; -22 from ptta_set_option):
 *   "pipelined_active"  1 when ptta_step_pipelined runs as itself on this handle as configured now; 0 when it degrades to ptta_step call by call
 *                       (generic-engine backbones, the dual-corner padded path, a statistics exchange or a gradient communicator bound,
 *                       validation arithmetic, the profiling leg) -- same results, the call-by-call price
 *   "thru_active"       1 when the step takes the two-stream `thru` schedule
 * bit-identical to the default: aux_stream, thru, adam_in_wgrad, fuse_first, fuse_head_bwd, mask_bits, graph; within bf16x3's own error (documented
 * in the tests): fuse_heads, heads_v2, cos_in_gemm.
 * Environment, read once per ptta_create (csrc/ptta_kernels.h ptta_create_env) because it decides allocation and arithmetic:
 *   PTTA_CONV_IMPL=naive (direct fp32 kernels, PTTA_DTYPE_F32 only), PTTA_ARITH=exact (fp32 MFMA, MSG_CHN PTTA_DTYPE_F32 only),
 *   PTTA_GRAPH=0|1 (initial value of "graph").  Recommended for the hosting process, not set by the library: HIP_FORCE_DEV_KERNARG=1
 *   before the HIP runtime initialises (kernel arguments in device memory: -7 % on the replayed step). */
int ptta_set_option(ptta_handle h, const char* key, int value);
int ptta_get_option(ptta_handle h, const char* key, int* value_host);

/* Measurement hook for bench.py: while enabled, the step runs kernel by kernel on one stream and EVERY launch is bracketed by hipEvents
 * on that stream, accounted to one of nine classes:  0 / 1 stride-1 3x3 32->32 convolution with ReLU on load, maps above / up to 1/4
 * resolution (together: the dominant kernel class of bench.py's `roofline`);  2 / 3 the same without ReLU (data gradients);  4 / 5 stride-2
 * and transposed convolutions;  6 the MLP heads (GEMMs + BatchNorm finalize);  7 first-layer / prediction convolutions (Cin <= 3 or
 * Cout = 1) and their data gradients;  8 everything else (resampling, loss, weight gradient, Adam, packing).  read returns the summed
 * duration (ms), the algorithmic bytes and MACs (SURVEY.md 8d counting rule: input + output + weight elements per conv / linear layer,
 * 0 for class 8) and the launch count since enable, after synchronising s. */
int ptta_profile(ptta_handle h, int enable);
int ptta_profile_read(ptta_handle h, int klass, double* ms_total_host, double* alg_bytes_host, double* macs_host,
                      int64_t* launches_host, ptta_stream s);

/* Test / debug hooks (not on the hot path). */
int ptta_debug_tensor(ptta_handle h, const char* name, float* dst, int64_t capacity, int64_t* numel_host, ptta_stream s);
int ptta_op_conv32(const float* in_nhwc, const float* weight, const float* bias, float* out_nhwc,
                   int b, int hin, int win, int mode, int relu_in, int in_major, int flip, int dtype, int naive,
                   ptta_stream s);
/* Diagnostic (tools/bench_chain.py): `reps` dependent launches of one stride-1 32->32 convolution (fp32 NHWC, default arithmetic) captured
 * into one hipGraph and replayed `replays` times: microseconds per launch INSIDE a replayed graph.  epi_flags: 2 = ReLU mask from `aux`,
 * 4 = skip addition of `aux`, 8 = the same chain launched directly (no graph), 16 = the layer loop (ONE launch per `reps` layers with a
 * device-wide barrier between layers; plain epilogue only; -62 when a bounded barrier spin ran out).  Synchronises. */
int ptta_op_conv32_chain(const float* in_nhwc, const float* weight, const float* bias, float* buf_a, float* buf_b, const float* aux,
                         int b, int h, int w, int relu_in, int epi_flags, int reps, int replays, float* us_per_launch_host, ptta_stream s);
/* ABI version of this header: ptta_version() of a loaded library must equal the PTTA_ABI_VERSION the binding was written against
 * (proxytta/_lib.py checks it at load).  2: ptta_step_pipelined takes the next frame's pointers; ptta_set_option / ptta_get_option;
 * PTTA_DTYPE_MIXED replaces the bf16 storage mode. */
#define PTTA_ABI_VERSION 2
int ptta_version(void);

#ifdef __cplusplus
}
#endif
#endif
