// csm_tokens.h -- launchers of tokens.hip (transformer ops of the layer-program executor), called from run_ops in nets.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace csm {
int launch_layernorm(const float *in, int in_ld, float *out, int out_ld, int64_t rows, int c, const float *gamma, const float *beta,
                     const float *eps /* device pointer */, hipStream_t st);
int launch_attention(const float *qkv, int ld, float *out, int out_ld, int n, int N, int heads, int d, const float *table, int gh, int gw,
                     hipStream_t st);
int launch_tokens(int mode, const float *in, int in_ld, float *out, int out_ld, int n, int np, int c, const float *cls, hipStream_t st);
int launch_depth_to_space(const float *in, int in_ld, float *out, int out_ld, int n, int h, int w, int k, int c, hipStream_t st);
}  // namespace csm
