// Micro-benchmark 4 (round 6): is v_mfma_f32_32x32x16_bf16 reproducible bit for bit by a CPU model?
// A split-bf16 convolution (x = x_hi + x_mid + x_lo, three bf16 terms = all 24 significand bits of an fp32; six of the nine cross products)
// would run the 1x1 / K >= 512 layers at <= 0.375 of today's fp32-MFMA time -- but parity here is BIT-EXACT against an oracle, so the
// hardware's K = 16 accumulation (order, internal width, rounding) must be a known function.  This probe runs the instruction on random
// and adversarial bf16 operands, one 32 x 32 x 16 problem per wave, and dumps operands and results; tools/ubench/mfma_bf16_model.py
// compares candidate models on the CPU (exact rational arithmetic).
// build: hipcc --offload-arch=gfx950 -O2 mfma_bf16_probe.hip -o mfma_bf16_probe ; run: ./mfma_bf16_probe out.bin [ntests]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A [T][32][16], B [T][16][32] as raw bf16 bits; C, D [T][32][32] f32
__global__ void k_probe(const uint16_t *A, const uint16_t *B, const float *C, float *D) {
    const int t = blockIdx.x, lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
    union { bf16x8 v; uint16_t u[8]; } a, b;
    for (int e = 0; e < 8; ++e) {
        a.u[e] = A[(t * 32 + i) * 16 + 8 * kh + e];                 // A[i][k = 8 kh + e]
        b.u[e] = B[(t * 16 + 8 * kh + e) * 32 + i];                 // B[k = 8 kh + e][j = i]
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i];
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * 32 + i] = d[r];
}

static uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }   // (inputs are built exactly representable)

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "mfma_bf16_probe.bin";
    const int T = argc > 2 ? atoi(argv[2]) : 64;
    std::mt19937 rng(12345);
    std::vector<uint16_t> A((size_t)T * 32 * 16), B((size_t)T * 16 * 32);
    std::vector<float> C((size_t)T * 32 * 32), D((size_t)T * 32 * 32);
    auto rnd_bf16 = [&](int emin, int emax) {          // sign, exponent in [emin, emax], 7 random mantissa bits
        const int e = emin + (int)(rng() % (unsigned)(emax - emin + 1));
        const uint32_t u = ((rng() & 1u) << 31) | ((uint32_t)(127 + e) << 23) | ((rng() & 0x7fu) << 16);
        float f; memcpy(&f, &u, 4); return f;
    };
    for (int t = 0; t < T; ++t) {
        // test families: 0 = narrow exponents, 1 = wide exponents (alignment / sticky bits matter), 2 = heavy cancellation (pairs +x, -x'),
        // 3 = one huge term + tiny terms, 4 = C huge against small products, 5 = zero C, small integers (everything exact: layout check)
        const int fam = t % 6;
        for (int i = 0; i < 32; ++i)
            for (int k = 0; k < 16; ++k) {
                float a = fam == 0 ? rnd_bf16(-1, 1) : fam == 1 ? rnd_bf16(-12, 12) : fam == 2 ? rnd_bf16(-2, 2) : fam == 3 ? (k == (i & 15) ? rnd_bf16(10, 12) : rnd_bf16(-12, -8))
                          : fam == 4 ? rnd_bf16(-6, -2) : (float)((int)(rng() % 7) - 3);
                A[((size_t)t * 32 + i) * 16 + k] = bf16_of(a);
            }
        for (int k = 0; k < 16; ++k)
            for (int j = 0; j < 32; ++j) {
                float b = fam == 5 ? (float)((int)(rng() % 7) - 3) : fam == 1 ? rnd_bf16(-12, 12) : rnd_bf16(-2, 2);
                B[((size_t)t * 16 + k) * 32 + j] = bf16_of(b);
            }
        if (fam == 2)           // cancellation: make A's odd-k entries nearly the negatives of the even-k ones, B's equal
            for (int i = 0; i < 32; ++i)
                for (int k = 0; k < 16; k += 2) {
                    A[((size_t)t * 32 + i) * 16 + k + 1] = (uint16_t)(A[((size_t)t * 32 + i) * 16 + k] ^ 0x8000u ^ (rng() & 1u));
                    for (int j = 0; j < 32; ++j) B[((size_t)t * 16 + k + 1) * 32 + j] = B[((size_t)t * 16 + k) * 32 + j];
                }
        for (int q = 0; q < 1024; ++q) {
            uint32_t u = ((rng() & 1u) << 31) | ((uint32_t)(127 + (fam == 4 ? 8 : fam == 3 ? -3 : (int)(rng() % 9) - 4)) << 23) | (rng() & 0x7fffffu);
            float c; memcpy(&c, &u, 4);
            C[(size_t)t * 1024 + q] = fam == 5 ? 0.0f : c;
        }
    }
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    k_probe<<<T, 64>>>(dA, dB, dC, dD);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    FILE *f = fopen(path, "wb");
    int32_t hdr[2] = {T, 0};
    fwrite(hdr, 4, 2, f); fwrite(A.data(), 2, A.size(), f); fwrite(B.data(), 2, B.size(), f); fwrite(C.data(), 4, C.size(), f); fwrite(D.data(), 4, D.size(), f);
    fclose(f);
    printf("wrote %s: %d problems of 32 x 32 x 16\n", path, T);
    return 0;
}
