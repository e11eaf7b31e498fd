// Interface between attention.hip (C ABI, dispatch) and attention2.hip (fused backward / wide forward kernels).
#pragma once
#include "common.h"

struct Mha2Problem {
  const void *q, *k, *v, *o, *d_o;
  const float* lse;
  const uint32_t* keep;
  float* delta;
  void *dq, *dk, *dv;
  int64_t ldq, ldk, ldv, ldo, lddq, lddk, lddv, B, H, Tq, Tk, dh;
  int mask_mode;
  const int32_t* len_k;
  float rate;
  int dtype;
};

int neurst_attention2_set_option(const char* name, int value);   // 1 when the name is one of this file's options
bool neurst_mha2_bwd_eligible(const Mha2Problem& p, int64_t ws_bytes);
int neurst_mha2_bwd_launch(const Mha2Problem& p, float* ws, hipStream_t s);

struct Mha2FwdProblem {
  const void *q, *k, *v;
  void* o;
  float* lse;
  uint32_t* keep;            // bitmap the forward fills (rate > 0; may be null: decisions are drawn and not stored)
  int keep_ready;            // the bitmap already holds the decisions
  int64_t ldq, ldk, ldv, ldo, bsq, bsk, bsv, bso, B, H, Tq, Tk, dh;
  int mask_mode;
  const int32_t* len_k;
  float rate;
  uint32_t key;
  int dtype;
};
bool neurst_mha2_fwd_eligible(const Mha2FwdProblem& p);
int neurst_mha2_fwd_launch(const Mha2FwdProblem& p, hipStream_t s);
