// Micro-benchmark: attainable rate of v_mfma_f32_32x32x2_f32 on gfx950 under the conditions of k_conv_mfma
// (dependent accumulate chains, 1..4 waves per SIMD, with / without the ds_read_b128 operand fetches).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_peak.hip -o tools/ubench/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float lds[128 * 36];
    for (int i = threadIdx.x; i < 128 * 36; i += 256) {
        // seed < 0: pseudo-random operands in (-1, 1) (realistic bit toggling -> realistic power / clock); else small integers
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = seed < 0.f ? ((float)(h & 0xffffff) / 8388608.0f - 1.0f) : seed * (float)(i & 7);
    }
    __syncthreads();
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *A = lds + ((wave >> 1) * 32 + (lane & 31)) * 36 + 4 * (lane >> 5);
    const float *B = lds + (64 + (wave & 1) * 32 + (lane & 31)) * 36 + 4 * (lane >> 5);
    float a0 = seed, b0 = seed * 2;
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                float4 af = *reinterpret_cast<const float4 *>(A + kb * 8);
                float4 bf = *reinterpret_cast<const float4 *>(B + kb * 8);
#pragma unroll
                for (int j = 0; j < NACC; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[j], 0, 0, 0);
                }
            }
            asm volatile("" ::: "memory");
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC, bool LDS>
void run(const char *name, int blocks_per_cu, int wg, float seed = 1.0f) {
    float *out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 6000, grid = 256 * blocks_per_cu;
    k_mfma<NACC, LDS><<<grid, wg>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k_mfma<NACC, LDS><<<grid, wg>>>(out, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flops = (double)grid * (wg / 64) * iters * 16.0 * NACC * 4096.0;
    printf("%-34s blocks/CU %d wg %3d : %8.3f ms  %7.1f TF/s  (%.1f%% of 157.3)\n", name, blocks_per_cu, wg, best,
           flops / best / 1e9, flops / best / 1e9 / 157.3 * 100);
    hipFree(out);
}

int main() {
    for (int b = 1; b <= 4; ++b) run<1, false>("1 acc chain, regs only", b, 256);
    for (int b = 1; b <= 4; b *= 2) run<2, false>("2 independent acc, regs only", b, 256);
    for (int b = 1; b <= 2; ++b) run<4, false>("4 independent acc, regs only", b, 256);
    for (int b = 1; b <= 4; ++b) run<1, true>("1 acc chain + ds_read_b128", b, 256);
    for (int b = 1; b <= 4; b *= 2) run<2, true>("2 acc (A shared) + ds_read_b128", b, 256);
    for (int b = 1; b <= 2; ++b) run<4, true>("4 acc + ds_read_b128", b, 256);
    for (int b = 1; b <= 4; b *= 2) run<1, true>("1 acc + ds_read_b128, RANDOM operands", b, 256, -1.0f);
    for (int b = 1; b <= 2; ++b) run<4, true>("4 acc + ds_read_b128, RANDOM operands", b, 256, -1.0f);
    return 0;
}
