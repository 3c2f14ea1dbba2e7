// tests/golden/cuda_on_cpu.h -- written for this repo (NOT from the reference).
// Minimal CUDA-execution-model shim so that the reference's *expanded kernel text*
// (captured at fixture-generation time from utils/cupy_utils.py::preprocess_kernel)
// can be executed sequentially on the CPU by tests/golden/make_golden_warp.py.
// One "thread" at a time, raster order => one legal interleaving of the CUDA kernel.
// Only the five float3 ops the kernels use are provided.
#pragma once
#include <cassert>
#include <cmath>
#include <cstring>
#include <algorithm>
using std::min; using std::max; using std::floor; using std::fabs; using std::round; using std::sqrt;
#define __global__
#define __device__
#define __forceinline__ inline
struct csm_dim3 { int x, y, z; };
static csm_dim3 blockIdx, blockDim, gridDim, threadIdx;
struct float3 { float x, y, z; };
static inline float3 make_float3(float x, float y, float z) { float3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline float3 operator+(float3 a, float3 b) { return make_float3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline float3 operator-(float3 a, float3 b) { return make_float3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline float3 operator*(float b, float3 a) { return make_float3(b * a.x, b * a.y, b * a.z); }
static inline float dot(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int atomicCAS(int* a, int cmp, int val) { int old = *a; if (old == cmp) *a = val; return old; }
static inline float atomicAdd(float* a, float v) { float old = *a; *a = old + v; return old; }
