// Micro-benchmark 3: conv main loop with LDS-DMA staging (global_load_lds_dwordx4), XOR-swizzled 128-B rows, raw s_barrier and
// counted vmcnt; wave tile (32*TM) x (32*TN) of v_mfma_f32_32x32x2_f32.  Loads come from an L2-resident footprint.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WM, int WN, int TM, int TN, int NSTAGE, int PIPE>
__global__ __launch_bounds__(64 * WM * WN + ((PIPE & 8) ? 64 : 0)) void k_glds(const float *__restrict__ src, float *out, int chunks, int64_t stride,
                                                         int footprint_blocks) {
    constexpr int NT = 64 * WM * WN, NW = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, ROWS = BM + BN;
    constexpr int G = ROWS / 8 / NW;                 // glds instructions per wave per chunk (8 rows of 128 B each)
    static_assert(G * 8 * NW == ROWS, "rows must split evenly");
    constexpr int kStage = ROWS * 32;                // floats
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    f32x16 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // loader: wave w, piece g covers rows 8*(w*G+g) .. +7 ; lane -> row +lane/8, physical slot lane%8, logical slot = phys ^ f(row)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;   // LDS byte address of the dynamic segment
    const float *gp[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        int row = 8 * (wave * G + g) + (lane >> 3);
        int slot = (lane & 7) ^ ((row >> 1) & 7);
        gp[g] = src + ((int64_t)(blockIdx.x % footprint_blocks) * ROWS + row) * stride + slot * 4;
    }
    int issue_count = 0;
    auto issue = [&](int stage) {
        const bool a_too = !(PIPE & 16) || (issue_count++ % 5) == 0;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (!a_too && 8 * (wave * G + g) < BM) continue;       // this piece is A rows: only every 5th chunk
            // asm, not the builtin: hipcc would otherwise put s_waitcnt vmcnt(0) in front of every later ds_read (it cannot tell the
            // DMA's LDS target from the stage being read) and serialise the prefetch.  vmcnt is counted by hand below.
            unsigned ldsaddr = __builtin_amdgcn_readfirstlane(lds_base + (stage * kStage + 8 * (wave * G + g) * 32) * 4), keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(gp[g]), "s"(ldsaddr) : "memory");
            gp[g] += 32;
        }
    };
    const int li = lane & 31, lh = lane >> 5;
    auto frags = [&](const float *S, int kb, float4 *af, float4 *bf) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            int row = 32 * (TM * wm + i) + li;
            af[i] = *reinterpret_cast<const float4 *>(S + row * 32 + (((2 * kb + lh) ^ ((row >> 1) & 7)) << 2));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int row = BM + 32 * (TN * wn + j) + li;
            bf[j] = *reinterpret_cast<const float4 *>(S + row * 32 + (((2 * kb + lh) ^ ((row >> 1) & 7)) << 2));
        }
    };
    auto mfmas = [&](const float4 *af, const float4 *bf) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float a = t == 0 ? af[i].x : t == 1 ? af[i].y : t == 2 ? af[i].z : af[i].w;
                    float b = t == 0 ? bf[j].x : t == 1 ? bf[j].y : t == 2 ? bf[j].z : bf[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
                }
    };
    auto compute = [&](int stage) {
        const float *S = lds + stage * kStage;
        if (PIPE & 1) {   // fragments of kb+1 are requested before the MFMAs of kb are issued
            float4 a0[TM], b0[TN], a1[TM], b1[TN];
            frags(S, 0, a0, b0);
            frags(S, 1, a1, b1); __builtin_amdgcn_sched_barrier(0); mfmas(a0, b0); __builtin_amdgcn_sched_barrier(0);
            frags(S, 2, a0, b0); __builtin_amdgcn_sched_barrier(0); mfmas(a1, b1); __builtin_amdgcn_sched_barrier(0);
            frags(S, 3, a1, b1); __builtin_amdgcn_sched_barrier(0); mfmas(a0, b0); __builtin_amdgcn_sched_barrier(0);
            mfmas(a1, b1);
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                float4 af[TM], bf[TN];
                frags(S, kb, af, bf);
                mfmas(af, bf);
            }
        }
    };
    if (PIPE & 8) {
        // wave specialisation: wave NW only moves data (all ROWS/8 pieces of a chunk), waves 0..NW-1 only compute
        constexpr int NP = ROWS / 8;
        if (wave == NW) {
            const float *lp[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                int row = 8 * p + (lane >> 3);
                int slot = (lane & 7) ^ ((row >> 1) & 7);
                lp[p] = src + ((int64_t)(blockIdx.x % footprint_blocks) * ROWS + row) * stride + slot * 4;
            }
            auto issue_all = [&](int stage) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    unsigned ldsaddr = __builtin_amdgcn_readfirstlane(lds_base + (stage * kStage + 8 * p * 32) * 4), keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(lp[p]), "s"(ldsaddr) : "memory");
                    lp[p] += 32;
                }
            };
            issue_all(0);
            for (int c = 0; c < chunks; ++c) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue_all((c + 1) & 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
        for (int c = 0; c < chunks; ++c) {
            __builtin_amdgcn_s_barrier();
            compute(c & 1);
        }
    } else if (NSTAGE == 2) {
        issue(0);
        for (int c = 0; c < chunks; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(PIPE & 4)) __builtin_amdgcn_s_barrier();
            if (!(PIPE & 2)) issue((c + 1) & 1);           // (dead on the last chunk in a real kernel)
            compute(c & 1);
        }
    } else {
        issue(0); issue(1);
        int st = 0;
        for (int c = 0; c < chunks; ++c) {
            if (G == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (G == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (G == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (G == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int nx = st + 2; if (nx >= 3) nx -= 3;
            issue(nx);
            compute(st);
            if (++st == 3) st = 0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 12345.678f) out[0] = s;
}

template <int WM, int WN, int TM, int TN, int NSTAGE, int PIPE>
void run(const char *name, const float *src, float *out, int tiles64, int chunks, int footprint, int stride = 2304) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    size_t ldsb = (size_t)NSTAGE * (BM + BN) * 32 * sizeof(float);
    int grid = (int)((int64_t)tiles64 * 64 * 64 / (BM * BN));
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_glds<WM, WN, TM, TN, NSTAGE, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_glds<WM, WN, TM, TN, NSTAGE, PIPE><<<grid, 64 * WM * WN + ((PIPE & 8) ? 64 : 0), ldsb>>>(src, out, chunks, stride, footprint);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        k_glds<WM, WN, TM, TN, NSTAGE, PIPE><<<grid, 64 * WM * WN + ((PIPE & 8) ? 64 : 0), ldsb>>>(src, out, chunks, stride, footprint);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flops = (double)grid * BM * BN * chunks * 32 * 2.0;
    printf("%-40s %3dx%-3d lds %3zu KB grid %5d : %8.1f us  %6.1f TF/s (%.0f%%)\n", name, BM, BN, ldsb >> 10, grid, best * 1e3,
           flops / best / 1e9, flops / best / 1e9 / 157.3 * 100);
    fflush(stdout);
}

int main() {
    float *src, *out;
    size_t n = (size_t)400 * 256 * 1024 + (size_t)4 * 256 * 2304 + 4096;
    hipMalloc(&src, n * 4); hipMemset(src, 0, n * 4); hipMalloc(&out, 4);
    for (int tiles : {3072}) {
        printf("-- work = %d 64x64 tiles x 32 chunks, rows 4 KB apart, footprint 400 tile-rows (1x1 conv, K = 1024, partly L2-missing)\n", tiles);
        run<2, 2, 1, 1, 2, 0>("64x64 2-stage", src, out, tiles, 32, 400, 1024);
        run<2, 2, 1, 1, 3, 0>("64x64 3-stage", src, out, tiles, 32, 400, 1024);
        run<2, 2, 2, 1, 2, 0>("128x64 2-stage", src, out, tiles, 32, 400, 1024);
        run<2, 2, 2, 1, 3, 0>("128x64 3-stage", src, out, tiles, 32, 400, 1024);
        run<2, 2, 2, 2, 2, 0>("128x128 2-stage", src, out, tiles, 32, 200, 1024);
        run<2, 2, 2, 2, 3, 0>("128x128 3-stage", src, out, tiles, 32, 200, 1024);
        run<2, 2, 2, 2, 2, 0>("128x128 2-stage, L2-resident", src, out, tiles, 32, 2, 1024);
    }
    return 0;
}
