// Geometry shared by conv.hip (general kernels, C ABI) and conv_mfma.hip (bf16 fast paths).
#pragma once
#include "common.h"

struct ConvGeom {
  int64_t B, H, W, Cin, Cout, Ho, Wo;
  int kh, kw, sh, sw, pt, pl;
};

// bf16 fast paths; each returns false when the shape is not one it handles (the caller then runs the general kernel)
bool neurst_conv_fwd_fast(const bf16_t* x, const float* w, const float* bias, const float* gamma, const float* beta, bf16_t* conv_out,
                          bf16_t* out, float* mean, float* rstd, const ConvGeom& g, hipStream_t s);
bool neurst_conv_bwd_pix_fast(const bf16_t* dout, const bf16_t* conv_out, const bf16_t* out, const float* gamma, const float* mean,
                              const float* rstd, bf16_t* dconv, float* dgamma, float* dbeta, float* dbias, const ConvGeom& g, hipStream_t s);
bool neurst_conv_dx_fast(const bf16_t* dconv, const float* w, bf16_t* dx, const ConvGeom& g, hipStream_t s);
// col_ws: caller workspace of neurst_conv2d_bwd_workspace_bytes() bytes (im2col image for the weight-gradient GEMM), may be null
bool neurst_conv_dw_fast(const bf16_t* x, const bf16_t* dconv, float* dw, void* col_ws, int64_t col_ws_bytes, const ConvGeom& g, hipStream_t s);
int64_t neurst_conv_col_bytes(const ConvGeom& g);
