// Internal interface between the C-ABI dispatch (ptta_api.hip) and the NLSPN engine (nlspn_api.hip).
#pragma once
#include <cstdint>
#include <string>
#include <hip/hip_runtime.h>
#include "../../include/ptta.h"

struct nlspn_engine;
nlspn_engine* nlspn_create(int n, int h, int w, const ptta_hparams* hp, int legacy_offset, int* rc);
void nlspn_destroy(nlspn_engine* e);
const char* nlspn_last_error(nlspn_engine* e);
int nlspn_set_hparams(nlspn_engine* e, const ptta_hparams* hp, hipStream_t s);
int nlspn_set_image_norm(nlspn_engine* e, float div, const float* mean, const float* stdv);
int nlspn_load_weights(nlspn_engine* e, const char* name, const void* tensor, const int64_t* shape, int ndim, hipStream_t s);
int nlspn_bind_adapted(nlspn_engine* e, const char* name, float* p, float* m, float* v);
int nlspn_adapted_count(nlspn_engine* e);
const char* nlspn_adapted_name(nlspn_engine* e, int index, int64_t* numel);
int nlspn_set_adam_step(nlspn_engine* e, int step, hipStream_t s);
int nlspn_get_adam_step(nlspn_engine* e, int* step, hipStream_t s);
int64_t nlspn_embedding_rows(nlspn_engine* e);
int nlspn_forward_train(nlspn_engine* e, const float* image, const float* sparse, float* depth, float* emb, float* ref, hipStream_t s);
int nlspn_forward_eval(nlspn_engine* e, const float* image, const float* sparse, float* depth, hipStream_t s);
int nlspn_step(nlspn_engine* e, const float* image, const float* loss_image, const float* sparse, const float* validity,
               float* depth_out, float* loss_info_out, hipStream_t s);
int nlspn_get_grad(nlspn_engine* e, const char* name, float* dst, int64_t capacity, hipStream_t s);
int nlspn_debug_tensor(nlspn_engine* e, const char* name, float* dst, int64_t capacity, int64_t* numel, hipStream_t s);
int nlspn_loss_forward(nlspn_engine* e, const float* loss_image, const float* depth, const float* sparse, const float* validity,
                       const float* emb, const float* ref, int64_t rows, float w_sd, float w_sm, float w_cos, float* loss_info_out, hipStream_t s);
int nlspn_loss_backward(nlspn_engine* e, const float* loss_image, const float* depth, const float* sparse, const float* validity,
                        const float* emb, const float* ref, int64_t rows, float* grad_depth_out, float* grad_ref_out, hipStream_t s);
int nlspn_backward(nlspn_engine* e, const float* grad_depth, const float* grad_ref, hipStream_t s);
int nlspn_adam_step(nlspn_engine* e, hipStream_t s);
int nlspn_set_grad(nlspn_engine* e, const char* name, const float* src, int64_t numel, hipStream_t s);
