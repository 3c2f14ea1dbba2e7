// Micro-benchmark 2: the k_conv_mfma main loop rebuilt feature by feature (FLAGS) to find which ingredient costs MFMA rate.
//   bit0 barrier per chunk   bit1 double-buffer addressing   bit2 4x ds_write_b128 per chunk   bit3 4x global_load_dwordx4 per
//   chunk, two chunks ahead  bit4 s_setprio around the MFMA phase  bit5 loads from an L2-resident footprint  bit6 all loads
//   from one address
// 64x64 tile, 2x2 waves, chunk = 32 channels = 16 MFMA 32x32x2 per wave; 36.9 KB LDS per block (4 blocks / CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kLd = 36, kStage = 128 * kLd;

template <int FLAGS>
__global__ __launch_bounds__(256, 2) void k_loop(const float *__restrict__ src, float *out, int chunks, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * kStage; i += 256) lds[i] = (float)(i & 7);
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int wm = wave >> 1, wn = wave & 1;
    const float *gp = (FLAGS & 64) ? src : src + ((int64_t)(blockIdx.x % ((FLAGS & 32) ? 2 : 1600)) * 128 + (tid >> 3)) * stride + (tid & 7) * 4;
    const int64_t gstep = (FLAGS & 64) ? 0 : 32, rstep = (FLAGS & 64) ? 0 : 32 * stride;
    float4 r0[4], r1[4];
    auto gload = [&](float4 *r) {
        if (FLAGS & 8) {
#pragma unroll
            for (int it = 0; it < 4; ++it) r[it] = *reinterpret_cast<const float4 *>(gp + (int64_t)it * rstep);
            gp += gstep;
        }
    };
    auto lstore = [&](const float4 *r, int buf) {
        if (FLAGS & 4) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
                *reinterpret_cast<float4 *>(lds + buf * kStage + ((tid >> 3) + 32 * it) * kLd + (tid & 7) * 4) =
                    (FLAGS & 8) ? r[it] : make_float4(1.f, 2.f, 3.f, 4.f);
        }
    };
    auto compute = [&](int buf) {
        const float *A = lds + buf * kStage + (32 * wm + (lane & 31)) * kLd + 4 * (lane >> 5);
        const float *B = lds + buf * kStage + (64 + 32 * wn + (lane & 31)) * kLd + 4 * (lane >> 5);
        if (FLAGS & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4 af = *reinterpret_cast<const float4 *>(A + kb * 8);
            float4 bf = *reinterpret_cast<const float4 *>(B + kb * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
        if (FLAGS & 16) __builtin_amdgcn_s_setprio(0);
    };
    gload(r0); lstore(r0, 0); gload(r1);
    __syncthreads();
    for (int c = 0; c + 1 < chunks; c += 2) {
        gload(r0);
        compute((FLAGS & 2) ? 0 : 0);
        lstore(r1, (FLAGS & 2) ? 1 : 0);
        if (FLAGS & 1) __syncthreads();
        gload(r1);
        compute((FLAGS & 2) ? 1 : 0);
        lstore(r0, 0);
        if (FLAGS & 1) __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[0] = s;
}

template <int FLAGS>
void run(const char *name, const float *src, float *out, int grid, int chunks) {
    size_t ldsb = 2 * kStage * sizeof(float);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_loop<FLAGS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_loop<FLAGS><<<grid, 256, ldsb>>>(src, out, chunks, 2304);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        k_loop<FLAGS><<<grid, 256, ldsb>>>(src, out, chunks, 2304);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flops = (double)grid * 4 * chunks * 16.0 * 4096.0;
    printf("%-52s grid %5d chunks %3d : %8.1f us  %6.1f TF/s (%.0f%%)\n", name, grid, chunks, best * 1e3, flops / best / 1e9,
           flops / best / 1e9 / 157.3 * 100);
}

int main() {
    float *src, *out;
    size_t n = (size_t)1600 * 128 * 2304 + 4096;
    hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4); hipMalloc(&out, 4);
    for (int grid : {1024, 1600, 2048}) {
        const int ch = 72;
        run<0>("mfma+ds_read", src, out, grid, ch);
        run<1>("+barrier", src, out, grid, ch);
        run<3>("+barrier+dbuf", src, out, grid, ch);
        run<7>("+barrier+dbuf+ds_write", src, out, grid, ch);
        run<15>("+barrier+dbuf+ds_write+gload", src, out, grid, ch);
        run<15 + 32>("+barrier+dbuf+ds_write+gload(L2-resident 2.4MB)", src, out, grid, ch);
        run<15 + 64>("+barrier+dbuf+ds_write+gload(same address)", src, out, grid, ch);
        run<8 + 64>("mfma+ds_read+gload(same address), no barrier", src, out, grid, ch);
    }
    return 0;
}
