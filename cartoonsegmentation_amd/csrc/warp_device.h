// warp_device.h -- device helpers shared by the point-cloud kernels (warp.hip, warptile.hip, autozoom.hip).
// The arithmetic follows the reference's CUDA text statement by statement (anime_3dkenburns/models/utils.py:76-99, common.py:78-81):
// every file that includes this header is built with -ffp-contract=off, so each fp32/fp64 operation rounds once, as written.
#pragma once
#include "csm_common.h"

namespace csmwarp {

struct ProjConst {
    double focal_baseline;  // focal*baseline, folded in double like the literal in the CUDA text
    double half_w, half_h;  // 0.5*W, 0.5*H
    float focal_f;          // make_float3(0,0,focal).z
    int W, H;
};
struct Shift { float x, y, z; };

template <bool SHIFT>
__device__ __forceinline__ void load_point(const float *__restrict__ P, int64_t N, int64_t p, Shift s,
                                           float &x, float &y, float &z) {
    x = P[p]; y = P[N + p]; z = P[2 * N + p];
    if (SHIFT) {  // common.py:78-81
        float r = z / (z + 0.0000001f);
        x = x * r + s.x; y = y * r + s.y; z = z + s.z;
    }
}

// models/utils.py:76-99  (mixed fp32/fp64 exactly as the untyped CUDA literals evaluate)
__device__ __forceinline__ bool project(float x, float y, float z, const ProjConst &pc, float &fx, float &fy,
                                        float &err) {
    if ((double)z < 0.001) return false;
    float lvx = 0.0f - x, lvy = 0.0f - y, lvz = 0.0f - z;
    float ax = 0.0f - x, ay = 0.0f - y, az = pc.focal_f - z;
    float num = ax * 0.0f + ay * 0.0f + az * 1.0f;
    float den = lvx * 0.0f + lvy * 0.0f + lvz * 1.0f;
    float dist = num / den;
    if ((double)fabsf(den) < 0.001) return false;
    float ix = x + dist * lvx;
    float iy = y + dist * lvy;
    fx = (float)(((double)ix + pc.half_w) - 0.5);
    fy = (float)(((double)iy + pc.half_h) - 0.5);
    err = (float)(1000000.0 - (pc.focal_baseline / ((double)z + 0.0000001)));
    return true;
}

// project() split for callers that evaluate many x shifts of one point (autozoom bands): everything that does not involve x
// (the rejection tests, dist, fy, err) once, then fx per shift.  Statement for statement the expressions of project(); with
// -ffp-contract=off  project(x, y, z) == { project_yz(y, z, ...); fx = project_x(x, dist) }  bit for bit.
__device__ __forceinline__ bool project_yz(float y, float z, const ProjConst &pc, float &dist, float &fy, float &err) {
    if ((double)z < 0.001) return false;
    float lvy = 0.0f - y, lvz = 0.0f - z;
    float az = pc.focal_f - z;
    // num = ax * 0 + ay * 0 + az * 1 and den = lvx * 0 + lvy * 0 + lvz * 1: the zero products of finite operands add nothing
    // (x + 0 == x for every x but -0, and az * 1 + 0 of a signed zero only changes the sign of a zero, which no later statement sees)
    float num = az * 1.0f;
    float den = lvz * 1.0f;
    dist = num / den;
    if ((double)fabsf(den) < 0.001) return false;
    float iy = y + dist * lvy;
    fy = (float)(((double)iy + pc.half_h) - 0.5);
    err = (float)(1000000.0 - (pc.focal_baseline / ((double)z + 0.0000001)));
    return true;
}
__device__ __forceinline__ float project_x(float x, float dist, const ProjConst &pc) {
    float lvx = 0.0f - x;
    float ix = x + dist * lvx;
    return (float)(((double)ix + pc.half_w) - 0.5);
}

__device__ __forceinline__ void corner_weights(float fx, float fy, int &x0, int &y0, float w[4]) {
    x0 = (int)floorf(fx); y0 = (int)floorf(fy);
    float x1 = (float)(x0 + 1), y1 = (float)(y0 + 1), xf = (float)x0, yf = (float)y0;
    w[0] = (x1 - fx) * (y1 - fy);  // NW
    w[1] = (fx - xf) * (y1 - fy);  // NE
    w[2] = (x1 - fx) * (fy - yf);  // SW
    w[3] = (fx - xf) * (fy - yf);  // SE
}

// float min through native integer atomics (replaces the CAS loop of utils/cupy_utils.py:21-29).
// Non-negative floats order like signed ints; negative floats order inversely like unsigned ints.
__device__ __forceinline__ void atomic_min_f32(float *addr, float v) {
    if (v >= 0.0f) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

inline ProjConst make_proj(int H, int W, double focal, double baseline) {
    ProjConst pc;
    pc.focal_baseline = focal * baseline;
    pc.half_w = 0.5 * W; pc.half_h = 0.5 * H;
    pc.focal_f = (float)focal; pc.W = W; pc.H = H;
    return pc;
}

__device__ __forceinline__ uint8_t to_u8(float v) {  // (x*255).clip(0,255).astype(uint8)  kenburns_effect.py:1040
    float u = v * 255.0f;
    u = u < 0.0f ? 0.0f : (u > 255.0f ? 255.0f : u);
    return (uint8_t)u;
}

// models/utils.py:101-135: the corner with the largest bilinear weight (ties resolved in the reference's if/else order);
// returns false when no branch fires (NaN weights)
__device__ __forceinline__ bool argmax_corner(const float w[4], int x0, int y0, int &cx, int &cy) {
    const float nw = w[0], ne = w[1], sw = w[2], se = w[3];
    if (nw >= ne && nw >= sw && nw >= se) { cx = x0; cy = y0; }
    else if (ne >= nw && ne >= sw && ne >= se) { cx = x0 + 1; cy = y0; }
    else if (sw >= nw && sw >= ne && sw >= se) { cx = x0; cy = y0 + 1; }
    else if (se >= nw && se >= ne && se >= sw) { cx = x0 + 1; cy = y0 + 1; }
    else return false;
    return true;
}

}  // namespace csmwarp
