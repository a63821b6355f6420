// Library-internal entry points (not part of the C ABI).
#pragma once
#include "common.h"

int tdx_pixel_sum(const float* g, float* out, int B, int HW, int C, hipStream_t st);
int tdx_reduce_partials(const float* partial, float* out, int nblk, int stride, int count,
                        hipStream_t st);
int tdx_initial_conv_fwd(const float* x, const float* w, const float* bias, float* out, int B, int H,
                         int W, hipStream_t st);
int tdx_initial_conv_wgrad_blocks(int B, int H, int W);
int tdx_initial_conv_wgrad(const float* x, const float* g, float* partial, float* dw, float* db,
                           int B, int H, int W, hipStream_t st);
int tdx_final_conv_fwd(const float* in, const float* w, const float* bias, float* out, int B, int H,
                       int W, hipStream_t st);
int tdx_final_conv_dgrad(const float* g_out, const float* w, float* g_in, int B, int H, int W,
                         hipStream_t st);
int tdx_final_conv_wgrad(const float* in, const float* g_out, float* partial, float* dw, float* db,
                         int B, int H, int W, hipStream_t st);
int tdx_time_embed_fwd(const int64_t* t, const int64_t* y, const float* const* P, float* pre,
                       float* emb, float* t1, float* t2, float* t3, int B, hipStream_t st);
int tdx_time_embed_bwd(const int64_t* t, const int64_t* y, const float* const* P, float* const* G,
                       const float* pre, const float* emb, const float* g_t1, const float* g_t2,
                       const float* g_t3, float* scratch, int B, int ncls, hipStream_t st);
