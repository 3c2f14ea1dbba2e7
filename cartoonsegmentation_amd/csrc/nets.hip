// nets.hip -- layer-program executor for the dense nets of the hot path (ISNet refine, LeReS depth,
// RTMDet-Ins) on gfx950.  C ABI: csm_run_program (include/csm355.h).
//
// Design (CDNA4-first, not a cuDNN call sequence):
//   * activations NHWC fp32; a tensor view = (base, channels, channel pitch) so torch.cat is free;
//   * every dense / grouped convolution is ONE implicit-GEMM kernel on the exact-fp32 matrix pipe
//     (v_mfma_f32_32x32x2_f32, 157 TF/s peak): M = output pixels, N = output channels,
//     K = taps x input channels.  256-thread blocks = 4 waves in a WM x WN grid, each wave owns
//     32 x (32*TN) outputs.  K is consumed in chunks of 32 channels of one tap: the A tile
//     (BM pixels x 128 B, fully coalesced because NHWC keeps a pixel's channels contiguous) and the
//     pre-packed weight tile (BN x 128 B) are register-staged into LDS (row pitch 36 floats =>
//     conflict-free ds_read_b128) and double buffered, one barrier per chunk.
//   * global loads run two K-chunks ahead of the MFMAs in two register sets; they are unconditional with a fixed
//     count per step so that hipcc emits counted s_waitcnt vmcnt(N) instead of draining the queue.
//   * folded-BN bias initialises the accumulator; activation / residual are fused in the epilogue.
//   * grouped 3x3 (ResNeXt, 32 groups) runs on the same kernel as block-diagonal 32-channel
//     super-groups (zero-padded weights): fmaf(x, 0, acc) is exact, so numerics are unchanged.
//   * blockIdx -> tile mapping is XCD-aware: the 8 XCDs each get a contiguous range of M tiles so
//     that the 3x3 halo re-reads of neighbouring tiles hit the same 4 MiB L2.
// Numerical contract: see include/csm355.h (one fmaf chain per output, fixed K order).
#include "csm_common.h"
#include "csm_conv.h"
#include "csm_tokens.h"
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>
#include <algorithm>
#include <utility>

#ifndef CSM_ILV
#define CSM_ILV 1        // persistent conv kernels: DMA pieces interleaved with the MFMA groups (0 = burst behind the barrier; A/B builds)
#endif

using namespace csmconv;

namespace {

constexpr int kLdsLd = 36;  // floats per LDS row: 32 + 4 pad (conflict-free b128 reads, see MI355X LDS notes)

typedef float f32x4v __attribute__((ext_vector_type(4)));

// Implicit-GEMM convolution on the exact-fp32 matrix pipe.
//   MT = 32: v_mfma_f32_32x32x2_f32, wave tile 32 x (32*TN); LDS rows hold 32 channels in natural order, a lane
//            (i, h) reads float4 at channel 4h of each 8-block: MFMA t multiplies channels (t, 4+t).
//   MT = 16: v_mfma_f32_16x16x4_f32, wave tile 16 x (16*TN) -- 4x more tiles for small feature maps, so that all
//            1024 SIMDs get work.  LDS 8-blocks are stored permuted [0,2,4,6,1,3,5,7]; lane (i, g) reads float2 at
//            position 2g: MFMA 1 multiplies channels (0,4,1,5), MFMA 2 (2,6,3,7).
// Both give the contract's chain order 0,4,1,5,2,6,3,7 per 8-block, so they are bit-identical to each other.
// FULLK: cin_g % 32 == 0, every chunk is 4 full 8-channel blocks -> the MFMA phase is straight-line code (no branch
// around it: a branch makes hipcc copy the 16 accumulator registers out and back every chunk behind a full MFMA drain).
#ifdef CSM_CONV_ABLATE
#define CSM_DBG(a) ((a).dbg)          // tuning build only (make ABLATE=1): phases can be switched off at run time
#else
#define CSM_DBG(a) 0
#endif
// SER (split-K executed serially): csm_op.ksplit = S cuts K into S runs of chunks, each its own fmaf chain, summed ((p0+p1)+p2)...
// -- that is part of the NUMERICAL contract and follows the per-sample shape only.  How the runs are EXECUTED is a speed decision:
// S blocks along grid z writing raw partials + k_splitk_reduce (small grids: batch 1), or -- SER -- one block that walks all S
// runs and combines them in registers at the run boundaries (`tot = tot + acc; acc = 0`: the same fp32 additions in the same
// order), so a batched program produces the bits of the single-frame program without the partial-sum traffic.
template <int MT, int WM, int WN, int TN, bool FULLK, bool SER = false>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN >= 8 ? 4 : 2)) void k_conv_mfma(ConvArgs a) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = MT * WM, BN = MT * WN * TN;
    constexpr int A_IT = (BM * 8 + NT - 1) / NT, B_IT = (BN * 8 + NT - 1) / NT;
    constexpr bool A_FULL = A_IT * NT == BM * 8, B_FULL = B_IT * NT == BN * 8;
    constexpr int NACC = MT == 32 ? 16 : 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int kStage = (BM + BN) * kLdsLd;   // floats per pipeline stage: A rows then B rows

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = MT == 32 ? (lane & 31) : (lane & 15);
    const int lh = MT == 32 ? (lane >> 5) : (lane >> 4);

    int mt, ntile, zz;
    block_to_tile(mt, ntile, zz);
    const int m0 = mt * BM, n0 = ntile * BN;
    const int g = SER ? zz : zz / a.ksplit, ks = SER ? 0 : zz - g * a.ksplit;
    const int ho = a.out.h, wo = a.out.w;
    const int cin_off = g * a.cin_g, cout_off = g * a.cout_g;

    // Per-thread A rows (output pixels): base pointer of the receptive-field origin and a per-tap validity mask,
    // computed once; the K loop then only adds block-uniform offsets (no integer divisions, ~2 VALU per load).
    const float *rowp[A_IT]; unsigned long long vmask[A_IT];
    const int c4 = (tid & 7) * 4;
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        int row = (tid + NT * it) >> 3;
        int m = m0 + row;
        bool rv = m < a.M && (A_FULL || row < BM);
        int mm = rv ? m : 0;
        int n = mm / (ho * wo), rem = mm - n * ho * wo;
        int oy = rem / wo, ox = rem - oy * wo;
        int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        rowp[it] = a.in.p + ((int64_t)(n * a.in.h + iy0) * a.in.w + ix0) * a.in.ld + cin_off + c4;
        unsigned long long vm = 0ull;
        if (rv)
            for (int kh = 0; kh < a.kh; ++kh)
                for (int kw = 0; kw < a.kw; ++kw) {
                    int iy = iy0 + kh * a.dil, ix = ix0 + kw * a.dil;
                    if (iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w) vm |= 1ull << (kh * a.kw + kw);
                }
        vmask[it] = vm;
    }
    const int Tall = a.kh * a.kw * a.ncb;
    const int c_begin = SER ? 0 : (int)(((int64_t)ks * Tall) / a.ksplit), T = SER ? Tall : (int)(((int64_t)(ks + 1) * Tall) / a.ksplit);
    // loader state = the NEXT chunk to fetch (block-uniform -> SGPRs).  Chunk order = the chain order: 32-channel block outer,
    // taps row-major inner (so that a 3x3 kernel can keep one block's input patch in LDS for all its taps, k_conv_patch).
    const int ntaps = a.kh * a.kw;
    int l_cb = c_begin / ntaps, l_tap = c_begin - l_cb * ntaps;
    int l_kh = l_tap / a.kw, l_kw = l_tap - l_kh * a.kw;
    const float *wp[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it)
        wp[it] = a.w + ((int64_t)g * Tall + c_begin) * a.npad * 32 + (int64_t)(n0 + ((tid + NT * it) >> 3)) * 32 + c4;

    // two register sets: loads run TWO chunks ahead of the MFMAs (set = parity of the chunk), so a chunk's HBM/L2 latency
    // is covered by two full compute phases; hipcc emits the counted vmcnt that leaves the younger set in flight.
    float4 ra[2][A_IT] = {}, rb[2][B_IT] = {};
    unsigned vbits[2] = {0u, 0u};           // validity of each load of a set (A: bit it, B: bit 8+it); zeros are applied at the LDS store
    auto gload = [&](const int set, const bool live) {   // always issues the same number of loads (see below)
        const int64_t toff = ((int64_t)l_kh * a.dil * a.in.w + l_kw * a.dil) * a.in.ld + l_cb * 32;
        const bool cv = live && !(CSM_DBG(a) & 1) && (FULLK || l_cb * 32 + c4 < a.cin_g);
        unsigned vb = 0u;
        // Loads are UNCONDITIONAL and their count per step is fixed (dead lanes / dead steps read a safe address): a branch
        // around a load makes hipcc fall back to vmcnt(0..3) at the next use, which would serialise the two-deep prefetch.
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            bool v = cv && ((vmask[it] >> l_tap) & 1ull);
            const float *p = v ? rowp[it] + toff : a.in.p;
            ra[set][it] = *reinterpret_cast<const float4 *>(p);
            vb |= v ? (1u << it) : 0u;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            int row = (tid + NT * it) >> 3;
            bool v = live && !(CSM_DBG(a) & 1) && n0 + row < a.npad && (B_FULL || row < BN);
            const float *p = v ? wp[it] : a.w;
            rb[set][it] = *reinterpret_cast<const float4 *>(p);
            vb |= v ? (1u << (8 + it)) : 0u;
            wp[it] += (int64_t)a.npad * 32;
        }
        vbits[set] = vb;
        ++l_tap;
        if (++l_kw == a.kw) { l_kw = 0; if (++l_kh == a.kh) { l_kh = 0; l_tap = 0; ++l_cb; } }
    };
    auto put = [&](float *dst, float4 v) {
        if (MT == 32) *reinterpret_cast<float4 *>(dst + c4) = v;
        else {  // permuted 8-block [0,2,4,6,1,3,5,7]
            float *b8 = dst + (c4 & ~7) + 2 * ((c4 >> 2) & 1);
            *reinterpret_cast<float2 *>(b8) = make_float2(v.x, v.z);
            *reinterpret_cast<float2 *>(b8 + 4) = make_float2(v.y, v.w);
        }
    };
    auto lstore = [&](const int set, int buf) {
        if (CSM_DBG(a) & 4) return;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            if (A_FULL || ((tid + NT * it) >> 3) < BM)
                put(lds + buf * kStage + ((tid + NT * it) >> 3) * kLdsLd, (vbits[set] >> it) & 1u ? ra[set][it] : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            if (B_FULL || ((tid + NT * it) >> 3) < BN)
                put(lds + buf * kStage + (BM + ((tid + NT * it) >> 3)) * kLdsLd, (vbits[set] >> (8 + it)) & 1u ? rb[set][it] : make_float4(0.f, 0.f, 0.f, 0.f));
    };

    // accumulators start at the (folded-BN) bias
    float acc[TN][NACC];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        int n = n0 + MT * (TN * wn + tn) + li;
        float b = (a.bias && ks == 0 && n < a.cout_g) ? a.bias[cout_off + n] : 0.0f;
#pragma unroll
        for (int r = 0; r < NACC; ++r) acc[tn][r] = b;
    }

    auto kblock = [&](const float *A, const float *B, int kb) {
        if (MT == 32) {
            float4 af = *reinterpret_cast<const float4 *>(A + kb * 8);
            float4 bf[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[tn] = *reinterpret_cast<const float4 *>(B + tn * 32 * kLdsLd + kb * 8);
            const float av[4] = {af.x, af.y, af.z, af.w};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const float bv = t == 0 ? bf[tn].x : (t == 1 ? bf[tn].y : (t == 2 ? bf[tn].z : bf[tn].w));
                    f32x16 c;
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = acc[tn][r];
                    c = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, c, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tn][r] = c[r];
                }
        } else {
            float2 af = *reinterpret_cast<const float2 *>(A + kb * 8);
            float2 bf[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bf[tn] = *reinterpret_cast<const float2 *>(B + tn * 16 * kLdsLd + kb * 8);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    f32x4v c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) c[r] = acc[tn][r];
                    c = __builtin_amdgcn_mfma_f32_16x16x4f32(t == 0 ? af.x : af.y, t == 0 ? bf[tn].x : bf[tn].y, c, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[tn][r] = c[r];
                }
        }
    };

    int cb = c_begin / ntaps, ctap = c_begin - cb * ntaps;      // position of the chunk being multiplied
    float tot[SER ? TN : 1][SER ? NACC : 1];
    int run = 0, next_b = SER ? (int)((int64_t)Tall / a.ksplit) : 0;          // SER: first chunk of the next run
    auto compute = [&](int chunk) {
        if constexpr (SER) {
            if (chunk == next_b) {                                             // block-uniform: S - 1 times per block
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < NACC; ++r) { tot[tn][r] = run == 0 ? acc[tn][r] : tot[tn][r] + acc[tn][r]; acc[tn][r] = 0.0f; }
                ++run; next_b = (int)(((int64_t)(run + 1) * Tall) / a.ksplit);
            }
        }
        const int buf = chunk & 1;
        const float *A = lds + buf * kStage + (MT * wm + li) * kLdsLd + (MT == 32 ? 4 : 2) * lh;
        const float *B = lds + buf * kStage + (BM + MT * TN * wn + li) * kLdsLd + (MT == 32 ? 4 : 2) * lh;
        if (CSM_DBG(a) & 2) return;
        if (FULLK) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) kblock(A, B, kb);
        } else {
            int rem = a.cin_g - cb * 32;
            if (++ctap == ntaps) { ctap = 0; ++cb; }
            int nkb = rem >= 32 ? 4 : (rem + 7) >> 3;
#pragma unroll 1
            for (int kb = 0; kb < nkb; ++kb) kblock(A, B, kb);
        }
    };
    gload(0, true);                             // chunk c_begin     -> set 0
    lstore(0, c_begin & 1);
    gload(1, c_begin + 1 < T);                  // chunk c_begin + 1 -> set 1, stays in flight
    __syncthreads();
    int chunk = c_begin;
    for (; chunk + 1 < T; chunk += 2) {
        gload(0, chunk + 2 < T);                // two ahead
        compute(chunk);
        lstore(1, (chunk + 1) & 1);             // needs only the older set: counted vmcnt keeps set 0 in flight
        __syncthreads();
        gload(1, chunk + 3 < T);
        compute(chunk + 1);
        lstore(0, (chunk + 2) & 1);             // (a dead step stores zeros into the idle buffer)
        __syncthreads();
    }
    if (chunk < T) compute(chunk);

    // epilogue.  MT=32: lane holds column li, rows (r&3)+8*(r>>2)+4*lh.  MT=16: column li, rows 4*lh + r.
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        int n = n0 + MT * (TN * wn + tn) + li;
        if (n >= a.cout_g) continue;
        float slope = a.slope ? a.slope[cout_off + n] : 0.0f;
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            int row = MT == 32 ? (r & 3) + 8 * (r >> 2) + 4 * lh : 4 * lh + r;
            int m = m0 + MT * wm + row;
            if (m >= a.M) continue;
            float v = acc[tn][r];
            if constexpr (SER) v = tot[tn][r] + v;
            if (CSM_DBG(a) & 8) { if (v == 123.456f) a.out.p[0] = v; continue; }
            if (!SER && a.ksplit > 1) { a.partial[((int64_t)m * a.ksplit + ks) * a.cout_g + n] = v; continue; }
            if (a.res_mode == 1) v += a.res.p[(int64_t)m * a.res.ld + cout_off + n];
            v = apply_act(v, a.act, slope);
            if (a.res_mode == 2) v += a.res.p[(int64_t)m * a.res.ld + cout_off + n];
            a.out.p[(int64_t)m * a.out.ld + cout_off + n] = v;
        }
    }
}


// ---- LDS-DMA implicit-GEMM convolution (the main kernel) ------------------------------------------------------------
// Same arithmetic as k_conv_mfma (one fmaf chain per output, chunk = (32-channel block, tap) block-major, 8-block order
// 0,4,1,5,2,6,3,7) --
// what changes is how operands reach the matrix pipe:
//  * tiles go global -> LDS by `buffer_load_dwordx4 ... lds` (no staging VGPRs, no ds_write, no per-element zero select):
//    one wave-instruction moves 8 rows x 128 B.  Out-of-image taps, M / N tails use the buffer range check: their lanes
//    carry offset 0x80000000, the load is out of range and the DMA writes zeros.
//  * LDS rows are exactly 128 B (the DMA writes lane-linear), 16-B slots XOR-swizzled by (row>>1)&7: applied to the SOURCE
//    address of the DMA and to the ds_read_b128 address, conflict-free for the 4x16 lane groups of ds_read_b128.
//  * each wave owns TM x TN accumulators of 32x32 (independent MFMA chains interleave, A/B fragments reused TN/TM times);
//    block tile (32 TM WM) x (32 TN WN), two LDS stages, ONE raw s_barrier per chunk, vmcnt counted by hand (the loads are
//    asm: with the builtin hipcc puts vmcnt(0) in front of every ds_read and the prefetch serialises).
// Requirements (host-checked, else k_conv_mfma): cin_g % 32 == 0, kh*kw <= 32, views < 2 GiB.

template <int WM, int WN, int TM, int TN, int NS, bool SER = false, bool ILV = (CSM_ILV != 0)>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_dma(ConvArgs a) {
    constexpr int NW = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;          // DMA pieces (8 rows) per wave per chunk
    static_assert(GA * 8 * NW == BM && GB * 8 * NW == BN, "tile rows must split evenly over the waves");
    constexpr int kStageF = (BM + BN) * 32;                     // floats per stage
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    int mt, ntile, zz;
    block_to_tile(mt, ntile, zz, a.ngroup);
    const int m0 = a.m_begin + mt * BM, n0 = ntile * BN;
    const int g = SER ? zz : zz / a.ksplit, ks = SER ? 0 : zz - g * a.ksplit;
    const int ho = a.out.h, wo = a.out.w;
    const int cin_off = g * a.cin_g, cout_off = g * a.cout_g;
    const int Tall = a.kh * a.kw * a.ncb;
    const int c_begin = SER ? 0 : (int)(((int64_t)ks * Tall) / a.ksplit), T = SER ? Tall : (int)(((int64_t)(ks + 1) * Tall) / a.ksplit);

    // buffer descriptors (raw, range-checked): activations view and this op's packed weights
    i32x4 ra, rb;
    {
        uint64_t pa = (uint64_t)a.in.p, pb = (uint64_t)a.w;
        unsigned na = (unsigned)((((int64_t)a.in.n * a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4);
        unsigned nb = (unsigned)((int64_t)a.groups * Tall * a.npad * 128);
        ra = i32x4{(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)na, 0x00020000};
        rb = i32x4{(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)nb, 0x00020000};
    }
    // per-lane loader state.  A piece g: rows 8*(wave*GA+g)+lane/8 of the tile; physical slot lane%8 holds logical slot
    // (lane%8) ^ ((row>>1)&7).  offA = byte offset of (pixel's receptive-field origin, channel) -- may be "negative" (wraps)
    // for border pixels; a VALID tap always brings it back inside the view.
    unsigned offA[GA], vmA[GA], offB[GB];
#pragma unroll
    for (int p = 0; p < GA; ++p) {
        int row = 8 * (wave * GA + p) + (lane >> 3);
        int slot = (lane & 7) ^ ((row >> 1) & 7);
        int m = m0 + row;
        bool rv = m < a.M;
        const RowSetup rs = row_setup(a, rv ? m : 0, rv);
        offA[p] = (unsigned)(((rs.n * a.in.h + rs.iy0) * a.in.w + rs.ix0) * a.in.ld + cin_off + slot * 4) * 4u;
        vmA[p] = rs.vm;
    }
#pragma unroll
    for (int p = 0; p < GB; ++p) {
        int row = 8 * (wave * GB + p) + (lane >> 3);
        int slot = (lane & 7) ^ ((row >> 1) & 7);
        offB[p] = n0 + row < a.npad ? (unsigned)((n0 + row) * 32 + slot * 4) * 4u : kOob;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned ldsA = lds0 + (unsigned)(wave * GA * 8) * 128u, ldsB = lds0 + (unsigned)(BM + wave * GB * 8) * 128u;

    // loader position = the NEXT chunk to fetch (block-uniform); chunk order = 32-channel block outer, taps row-major inner
    const int ntaps = a.kh * a.kw;
    int l_cb = c_begin / ntaps, l_tap = c_begin - l_cb * ntaps;
    int l_kh = l_tap / a.kw, l_kw = l_tap - l_kh * a.kw;
    unsigned l_w = (unsigned)(((int64_t)g * Tall + c_begin) * a.npad * 128);     // byte offset of the chunk's weight tile
    // one DMA piece of the loader's current chunk (pieces 0 .. GA-1: activations, GA .. GA+GB-1: weights); !live: every lane out of range
    auto piece = [&](auto PC, int stage, bool live) {
        constexpr int p = decltype(PC)::value;
        const unsigned sb = (unsigned)stage * (unsigned)(kStageF * 4);
        if constexpr (p < GA) {
            const unsigned coff = (unsigned)(((l_kh * a.dil * a.in.w + l_kw * a.dil) * a.in.ld + l_cb * 32) * 4);
            dma16((live && ((vmA[p] >> l_tap) & 1u)) ? offA[p] + coff : kOob, ra, ldsA + sb + (unsigned)p * 1024u);
        } else
            dma16((!live || offB[p - GA] == kOob) ? kOob : offB[p - GA] + l_w, rb, ldsB + sb + (unsigned)(p - GA) * 1024u);
    };
    auto advance = [&]() {
        l_w += (unsigned)a.npad * 128u;
        ++l_tap;
        if (++l_kw == a.kw) { l_kw = 0; if (++l_kh == a.kh) { l_kh = 0; l_tap = 0; ++l_cb; } }
    };
    auto issue = [&](int stage) {
        [&]<int... P>(std::integer_sequence<int, P...>) { (piece(std::integral_constant<int, P>{}, stage, true), ...); }(std::make_integer_sequence<int, GA + GB>{});
        advance();
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int n = n0 + 32 * (TN * wn + j) + li;
        float b = (a.bias && ks == 0 && n < a.cout_g) ? a.bias[cout_off + n] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
    }
    // MFMA-side fragment addresses: row (32*tile + li), logical slot 2*kb + lh -> physical ^ ((li>>1)&7)
    int sw[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) sw[kb] = ((2 * kb + lh) ^ ((li >> 1) & 7)) * 4;
    const int rowA = (32 * TM * wm + li) * 32, rowB = (BM + 32 * TN * wn + li) * 32;
    f32x16 tot[SER ? TM : 1][SER ? TN : 1];
    int run = 0, next_b = SER ? (int)((int64_t)Tall / a.ksplit) : 0;          // SER: first chunk of the next run
    auto compute = [&](int stage, int chunk) {
        if constexpr (SER) {
            if (chunk == next_b) {                                             // block-uniform: S - 1 times per block
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { tot[i][j][r] = run == 0 ? acc[i][j][r] : tot[i][j][r] + acc[i][j][r]; acc[i][j][r] = 0.0f; }
                ++run; next_b = (int)(((int64_t)(run + 1) * Tall) / a.ksplit);
            }
        }
        const float *S = lds + stage * kStageF;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(S + rowA + i * 1024 + sw[kb]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(S + rowB + j * 1024 + sw[kb]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = t == 0 ? af[i].x : (t == 1 ? af[i].y : (t == 2 ? af[i].z : af[i].w));
                        const float bv = t == 0 ? bf[j].x : (t == 1 ? bf[j].y : (t == 2 ? bf[j].z : bf[j].w));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
    };

    // ILV (two-stage pipeline): the DMA pieces of the next chunk go out BETWEEN the MFMA groups of this one instead of in a burst behind
    // the barrier (see k_conv_dma_p); branch-free -- behind the last chunk the lanes are out of range and the DMA writes zeros into the
    // stage nobody reads any more
    auto compute_ilv = [&](int stage, int fill, int chunk, bool live) {
        if constexpr (SER) {
            if (chunk == next_b) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { tot[i][j][r] = run == 0 ? acc[i][j][r] : tot[i][j][r] + acc[i][j][r]; acc[i][j][r] = 0.0f; }
                ++run; next_b = (int)(((int64_t)(run + 1) * Tall) / a.ksplit);
            }
        }
        const float *S = lds + stage * kStageF;
        float4 af[2][TM], bf[2][TN];
        auto rd = [&](int kb, int buf) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[buf][i] = *reinterpret_cast<const float4 *>(S + rowA + i * 1024 + sw[kb]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[buf][j] = *reinterpret_cast<const float4 *>(S + rowB + j * 1024 + sw[kb]);
        };
        rd(0, 0);
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int kb = G / 4, t = G % 4, buf = kb & 1;
                if constexpr (G < GA + GB) { piece(std::integral_constant<int, G>{}, fill, live); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (t == 1 && kb < 3) { rd(kb + 1, buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = t == 0 ? af[buf][i].x : (t == 1 ? af[buf][i].y : (t == 2 ? af[buf][i].z : af[buf][i].w));
                        const float bv = t == 0 ? bf[buf][j].x : (t == 1 ? bf[buf][j].y : (t == 2 ? bf[buf][j].z : bf[buf][j].w));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 16>{});
        advance();
    };

    if constexpr (ILV && GA + GB <= 16) {
        // NS stages: while chunk c is multiplied, the pieces of chunk c + NS - 1 go out between its MFMA groups (into the stage chunk c - 1
        // was read from), so a piece has NS - 2 further chunks to land in before it is waited for -- HBM / Infinity-Cache misses
        // included.  Every step issues exactly GA + GB pieces (dead ones past the end), so the counted wait "at most (NS - 2)(GA + GB)
        // outstanding" always means "the pieces of this chunk have landed" (loads retire in order).
        for (int s0 = 0; s0 < NS - 1; ++s0) {
            const bool live = c_begin + s0 < T;
            [&]<int... P>(std::integer_sequence<int, P...>) { (piece(std::integral_constant<int, P>{}, s0, live), ...); }(std::make_integer_sequence<int, GA + GB>{});
            advance();
        }
        for (int chunk = c_begin, st = 0; chunk < T; ++chunk, st = (st + 1 == NS ? 0 : st + 1)) {
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"((NS - 2) * (GA + GB)) : "memory");
            __builtin_amdgcn_s_barrier();
            compute_ilv(st, st == 0 ? NS - 1 : st - 1, chunk, chunk + NS - 1 < T);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing (dead) fetches must land before the block's LDS is released
    } else if constexpr (NS == 2) {
        issue(0);
        for (int chunk = c_begin, st = 0; chunk < T; ++chunk, st ^= 1) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces of `chunk` have landed, its fragment reads of stage st^1 have completed ...
            __builtin_amdgcn_s_barrier();                          // ... everybody's have, and everybody is done reading stage st^1
            if (chunk + 1 < T) issue(st ^ 1);
            compute(st, chunk);
        }
    } else {
        // NS stages: the loads of chunk + NS - 1 are issued while chunk is consumed, so a load may take NS - 1 chunk times
        // (L2 misses of the short-K-chunk 1x1 layers) before it stalls the pipe.  vmcnt retires in order: "at most
        // (NS - 2) * (GA + GB) outstanding" == the pieces of `chunk` have landed.
        for (int s0 = 0; s0 < NS - 1; ++s0)
            if (c_begin + s0 < T) issue(s0);
        for (int chunk = c_begin, st = 0; chunk < T; ++chunk, st = (st + 1 == NS ? 0 : st + 1)) {
            if (chunk + NS - 2 < T) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * (GA + GB)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (in the barrier's own block: tools/check_isa_barriers.py)
            __builtin_amdgcn_s_barrier();                          // everybody is done reading the stage refilled next
            if (chunk + NS - 1 < T) issue(st == 0 ? NS - 1 : st - 1);
            compute(st, chunk);
        }
    }

    // epilogue: lane holds column li of each 32x32 tile, rows (r&3) + 8*(r>>2) + 4*lh
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int n = n0 + 32 * (TN * wn + j) + li;
        if (n >= a.cout_g) continue;
        float slope = a.slope ? a.slope[cout_off + n] : 0.0f;
        // (row pointers once per accumulator: the 16 rows of a lane are at compile-time row offsets x the uniform pitch -- no per-element
        // 64-bit multiply; the quarter-rate integer multiplies were ~500 cycles of a tile's epilogue)
        const int64_t ldo = a.out.ld, ldr = a.res.ld;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + 32 * (TM * wm + i) + 4 * lh;
            float *ob = a.out.p + (int64_t)mb * ldo + cout_off + n;
            const float *rb = a.res_mode ? a.res.p + (int64_t)mb * ldr + cout_off + n : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mo = (r & 3) + 8 * (r >> 2), m = mb + mo;
                if (m >= a.M) continue;
                float v = acc[i][j][r];
                if constexpr (SER) v = tot[i][j][r] + v;
                if (!SER && a.ksplit > 1) { a.partial[((int64_t)m * a.ksplit + ks) * a.cout_g + n] = v; continue; }
                if (a.res_mode == 1) v += rb[mo * ldr];
                v = apply_act(v, a.act, slope);
                if (a.res_mode == 2) v += rb[mo * ldr];
                ob[mo * ldo] = v;
            }
        }
    }
}


// ---- persistent form of k_conv_dma: a block walks SEVERAL tiles and the loader runs one chunk ahead ACROSS tile boundaries ----------
// In k_conv_dma every tile pays its prologue (address set-up, the first chunk's DMA round trip: ~3 us) and its epilogue with the matrix
// pipe idle, and because all tiles of a launch take the same time the blocks of a CU stay in lock-step: their prologues never run under
// another block's MFMA phase.  For short-K layers (K = 288: nine chunks, ~15 us of MFMA per tile) that is a fifth of the kernel.  Here
// the grid is one round of resident blocks; a block takes tiles i, i + stride, ... of its XCD's run (the same XCD-aware order), and
// behind the barrier of a tile's LAST chunk it sets the loader up for the NEXT tile and sends that tile's chunk 0 into the free stage:
// the round trip runs under the last chunk's MFMAs and the epilogue's stores.  Same chunks, same chain per output: the bits of every
// other tile configuration.  SER: the serial split-K walk of k_conv_dma (runs combined in registers at the run boundaries).
template <int WM, int WN, int TM, int TN, bool SER = false, bool ILV = (CSM_ILV != 0)>
__global__ __launch_bounds__(64 * WM * WN) void k_conv_dma_p(ConvArgs a, int n_n /* N tiles per group */, int total /* tiles */) {
    constexpr int NW = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int GA = BM / 8 / NW, GB = BN / 8 / NW;
    static_assert(GA * 8 * NW == BM && GB * 8 * NW == BN, "tile rows must split evenly over the waves");
    constexpr int kStageF = (BM + BN) * 32;
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int ho = a.out.h, wo = a.out.w;
    const int Tall = a.kh * a.kw * a.ncb;
    // this block's tiles: XCD x = blockIdx & 7 owns the run [start, start + len) of the (z, m-tile, n-tile) order, n fastest
    const int per = (int)(gridDim.x >> 3), x = (int)(blockIdx.x & 7u), i0 = (int)(blockIdx.x >> 3);
    const int q = total >> 3, r = total & 7;
    const int start = x * q + (x < r ? x : r), len = q + (x < r ? 1 : 0);
    if (i0 >= len) return;
    const int per_z = a.m_tiles * n_n;

    i32x4 ra, rb;
    {
        uint64_t pa = (uint64_t)a.in.p, pb = (uint64_t)a.w;
        unsigned na = (unsigned)((((int64_t)a.in.n * a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4);
        unsigned nb = (unsigned)((int64_t)a.groups * Tall * a.npad * 128);
        ra = i32x4{(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)na, 0x00020000};
        rb = i32x4{(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)nb, 0x00020000};
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned ldsA = lds0 + (unsigned)(wave * GA * 8) * 128u, ldsB = lds0 + (unsigned)(BM + wave * GB * 8) * 128u;

    // loader state of the tile being FETCHED (one chunk ahead of the tile being computed)
    unsigned offA[GA], vmA[GA], offB[GB];
    int l_cb = 0, l_tap = 0, l_kh = 0, l_kw = 0;
    unsigned l_w = 0u;
    auto loader_setup = [&](int k, bool live) {                 // tile k of the run (clamped by the caller); !live: every lane out of range
        const int j = start + k;
        const int g = j / per_z, rem = j - g * per_z;
        int mt, nt;
        rem_to_tile((unsigned)rem, (unsigned)a.m_tiles, (unsigned)n_n, a.ngroup, mt, nt);
        const int m0 = mt * BM, n0 = nt * BN, cin_off = g * a.cin_g;
#pragma unroll
        for (int p = 0; p < GA; ++p) {
            int row = 8 * (wave * GA + p) + (lane >> 3);
            int slot = (lane & 7) ^ ((row >> 1) & 7);
            int m = m0 + row;
            bool rv = live && m < a.M;
            const RowSetup rs = row_setup(a, rv ? m : 0, rv);
            offA[p] = (unsigned)(((rs.n * a.in.h + rs.iy0) * a.in.w + rs.ix0) * a.in.ld + cin_off + slot * 4) * 4u;
            vmA[p] = rs.vm;
        }
#pragma unroll
        for (int p = 0; p < GB; ++p) {
            int row = 8 * (wave * GB + p) + (lane >> 3);
            int slot = (lane & 7) ^ ((row >> 1) & 7);
            offB[p] = (live && n0 + row < a.npad) ? (unsigned)((n0 + row) * 32 + slot * 4) * 4u : kOob;
        }
        l_cb = 0; l_tap = 0; l_kh = 0; l_kw = 0;
        l_w = (unsigned)((int64_t)g * Tall * a.npad * 128);
    };
    // one DMA piece of the loader's current chunk (pieces 0 .. GA-1: activations, GA .. GA+GB-1: weights), then the step to the next chunk
    auto piece = [&](auto PC, int stage) {
        constexpr int p = decltype(PC)::value;
        const unsigned sb = (unsigned)stage * (unsigned)(kStageF * 4);
        if constexpr (p < GA) {
            const unsigned coff = (unsigned)(((l_kh * a.dil * a.in.w + l_kw * a.dil) * a.in.ld + l_cb * 32) * 4);
            dma16(((vmA[p] >> l_tap) & 1u) ? offA[p] + coff : kOob, ra, ldsA + sb + (unsigned)p * 1024u);
        } else
            dma16(offB[p - GA] == kOob ? kOob : offB[p - GA] + l_w, rb, ldsB + sb + (unsigned)(p - GA) * 1024u);
    };
    auto advance = [&]() {
        l_w += (unsigned)a.npad * 128u;
        ++l_tap;
        if (++l_kw == a.kw) { l_kw = 0; if (++l_kh == a.kh) { l_kh = 0; l_tap = 0; ++l_cb; } }
    };
    auto issue = [&](int stage) {
        [&]<int... P>(std::integer_sequence<int, P...>) { (piece(std::integral_constant<int, P>{}, stage), ...); }(std::make_integer_sequence<int, GA + GB>{});
        advance();
    };

    int sw[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) sw[kb] = ((2 * kb + lh) ^ ((li >> 1) & 7)) * 4;
    const int rowA = (32 * TM * wm + li) * 32, rowB = (BM + 32 * TN * wn + li) * 32;
    f32x16 acc[TM][TN];
    f32x16 tot[SER ? TM : 1][SER ? TN : 1];
    int run = 0, next_b = 0;                                    // SER: first chunk of the next run (reset per tile)
    // ILV: the chunk's MFMAs with the DMA pieces of the NEXT chunk spread between them -- piece g goes out behind MFMA group g (a group =
    // one k step of all TM x TN accumulators), so the pieces leave in the first half of the chunk and the matrix pipe never waits for a
    // burst of GA + GB address computations and DMA issues behind the barrier (each costs the wave 60-180 cycles of issue time, which an
    // MFMA in flight covers).  The fragments of k-block kb + 1 are requested behind the second group of kb.  The order is pinned with
    // sched_barrier: hipcc otherwise regroups the asm statements in front of the MFMAs.
    auto compute_ilv = [&](int stage, int fill) {
        const float *S = lds + stage * kStageF;
        float4 af[2][TM], bf[2][TN];
        auto rd = [&](int kb, int buf) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[buf][i] = *reinterpret_cast<const float4 *>(S + rowA + i * 1024 + sw[kb]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[buf][j] = *reinterpret_cast<const float4 *>(S + rowB + j * 1024 + sw[kb]);
        };
        rd(0, 0);
        [&]<int... G>(std::integer_sequence<int, G...>) {
            ([&] {
                constexpr int kb = G / 4, t = G % 4, buf = kb & 1;
                if constexpr (G < GA + GB) { piece(std::integral_constant<int, G>{}, fill); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (t == 1 && kb < 3) { rd(kb + 1, buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = t == 0 ? af[buf][i].x : (t == 1 ? af[buf][i].y : (t == 2 ? af[buf][i].z : af[buf][i].w));
                        const float bv = t == 0 ? bf[buf][j].x : (t == 1 ? bf[buf][j].y : (t == 2 ? bf[buf][j].z : bf[buf][j].w));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }(), ...);
        }(std::make_integer_sequence<int, 16>{});
        static_assert(GA + GB <= 16, "one DMA piece per MFMA group");
        advance();
    };
    auto run_boundary = [&](int chunk) {                        // block-uniform: S - 1 times per tile
        if constexpr (SER) {
            if (chunk == next_b) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int rr = 0; rr < 16; ++rr) { tot[i][j][rr] = run == 0 ? acc[i][j][rr] : tot[i][j][rr] + acc[i][j][rr]; acc[i][j][rr] = 0.0f; }
                ++run; next_b = (int)(((int64_t)(run + 1) * Tall) / a.ksplit);
            }
        }
    };
    auto compute = [&](int stage) {
        const float *S = lds + stage * kStageF;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(S + rowA + i * 1024 + sw[kb]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(S + rowB + j * 1024 + sw[kb]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = t == 0 ? af[i].x : (t == 1 ? af[i].y : (t == 2 ? af[i].z : af[i].w));
                        const float bv = t == 0 ? bf[j].x : (t == 1 ? bf[j].y : (t == 2 ? bf[j].z : bf[j].w));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
    };

    loader_setup(i0, true);
    issue(0);
    int st = 0;
    for (int k = i0; k < len; k += per) {
        const int j = start + k;
        const int g = j / per_z, rem = j - g * per_z;
        int mt, nt;
        rem_to_tile((unsigned)rem, (unsigned)a.m_tiles, (unsigned)n_n, a.ngroup, mt, nt);
        const int m0 = mt * BM, n0 = nt * BN, cout_off = g * a.cout_g;
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            int n = n0 + 32 * (TN * wn + jj) + li;
            float b = (a.bias && n < a.cout_g) ? a.bias[cout_off + n] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) acc[i][jj][rr] = b;
        }
        if constexpr (SER) { run = 0; next_b = (int)((int64_t)Tall / a.ksplit); }
        for (int chunk = 0; chunk + 1 < Tall; ++chunk, st ^= 1) {
            // (lgkmcnt: this wave's fragment reads of the stage refilled next must have COMPLETED before it arrives -- hipcc may sink
            // the last MFMAs of the previous chunk, and with them the wait for their operands, below the barrier)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (ILV) { run_boundary(chunk); compute_ilv(st, st ^ 1); }
            else { issue(st ^ 1); run_boundary(chunk); compute(st); }
        }
        // the tile's last chunk: the NEXT tile's chunk 0 goes out behind the barrier (branch-free: past the end every lane is out
        // of range and the DMA writes zeros into a stage nobody reads)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int kn = k + per;
            const bool more = kn < len;
            loader_setup(more ? kn : k, more);
            if constexpr (!ILV) issue(st ^ 1);
        }
        run_boundary(Tall - 1);
        if constexpr (ILV) compute_ilv(st, st ^ 1); else compute(st);
        st ^= 1;
        // epilogue: lane holds column li of each 32x32 tile, rows (r&3) + 8*(r>>2) + 4*lh
#pragma unroll
        for (int jj = 0; jj < TN; ++jj) {
            int n = n0 + 32 * (TN * wn + jj) + li;
            if (n >= a.cout_g) continue;
            float slope = a.slope ? a.slope[cout_off + n] : 0.0f;
            const int64_t ldo = a.out.ld, ldr = a.res.ld;          // (row pointers once per accumulator, as in k_conv_dma)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mb = m0 + 32 * (TM * wm + i) + 4 * lh;
                float *ob = a.out.p + (int64_t)mb * ldo + cout_off + n;
                const float *rb = a.res_mode ? a.res.p + (int64_t)mb * ldr + cout_off + n : nullptr;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const int mo = (rr & 3) + 8 * (rr >> 2), m = mb + mo;
                    if (m >= a.M) continue;
                    float v = acc[i][jj][rr];
                    if constexpr (SER) v = tot[i][jj][rr] + v;
                    if (a.res_mode == 1) v += rb[mo * ldr];
                    v = apply_act(v, a.act, slope);
                    if (a.res_mode == 2) v += rb[mo * ldr];
                    ob[mo * ldo] = v;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing (dead) fetch must land before the block's LDS is released
}

// Swizzle key of patch pixel pp = py * PW + px (k_conv_patch / k_conv_patch_p): the 16-B slot s of a pixel's 128-B row lives at physical slot
// s ^ key.  A ds_read_b128 is serviced in groups of 16 lanes, conflict-free when they hit 16 distinct 16-B bank slots, i.e. distinct
// (px & 1, key) -- the row pitch of 128 B makes the pixel's parity the upper half of the bank slot, and PW is even.  The 16 lanes of a group
// read 16 tile pixels: with a 16-wide tile they have 16 consecutive px (key = px >> 1 suffices, whatever their rows); with an 8-wide tile 8
// consecutive px on rows of either parity (+ 4 for odd rows).  Round 3 used the key of the LINEAR index ((pp >> 1) & 7), which the 18-pixel
// row pitch of the patch misaligns: two of every sixteen lanes collided and every A-fragment read took 8 LDS cycles instead of 4
// (SQ_LDS_BANK_CONFLICT = 40 % of SQ_LDS_IDX_ACTIVE, profiles/r04_conv_pmc.txt).
template <int PW, int TW> __device__ __forceinline__ int patch_key(int pp) {
    const int py = pp / PW, px = pp - py * PW;
    return ((px >> 1) + (TW == 8 ? 4 * (py & 1) : 0)) & 7;
}

// ---- 3x3 (stride 1, dilation 1) convolution with input-patch re-use -------------------------------------------------------
// k_conv_dma moves the A tile (BM pixels x 32 channels) once per (32-channel block, tap): nine times per block for a 3x3.
// The micro-benchmark (tools/ubench/glds_loop.hip, "A/5") shows that LDS-DMA volume is what costs MFMA rate (64x64 tile:
// 74 % -> 81 %, 128x128: 83 % -> 87 % when the A moves drop five-fold), so here the output tile is a TH x TW pixel rectangle
// and the block keeps the (TH+2) x (TW+2) x 32-channel input PATCH of the current channel block in LDS for all nine taps
// (the chain order is block-major for exactly this reason): the A fragment of output pixel (y, x) under tap (kh, kw) is patch
// pixel (y+kh, x+kw).  Patch: 2 stages (the next block's patch is fetched during the first tap of the current one); weights:
// 2 stages, one tile per tap.  Same 128-B rows / XOR swizzle / buffer-range-check zero fill / raw barrier as k_conv_dma.
template <int WM, int WN, int TM, int TN, int TW, bool SER = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void k_conv_patch(ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int NW = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, TH = BM / TW;
    constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW;
    constexpr int NPP = (NPIX + 7) / 8;                         // patch DMA pieces (8 pixels each)
    constexpr int QP = (NPP + NW - 1) / NW;                     // per wave
    constexpr int GB = BN / 8 / NW;
    static_assert(GB * 8 * NW == BN && TH * TW == BM && (TW == 16 || TW == 8), "tile shape");
    constexpr int kPatchF = QP * NW * 8 * 32, kBF = BN * 32;    // floats per patch stage (every wave's QP pieces have a home) / weight stage
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [patch 0][patch 1][B 0][B 1]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    int mt, ntile, zz;
    block_to_tile(mt, ntile, zz, a.ngroup);
    const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, n = mt / (tiles_x * tiles_y);
    const int n0 = ntile * BN;
    const int g = SER ? zz : zz / a.ksplit, ks = SER ? 0 : zz - g * a.ksplit;
    const int ho = a.out.h, wo = a.out.w;
    const int cin_off = g * a.cin_g, cout_off = g * a.cout_g;
    const int Tall = 9 * a.ncb;
    const int c_begin = SER ? 0 : (int)(((int64_t)ks * Tall) / a.ksplit), T = SER ? Tall : (int)(((int64_t)(ks + 1) * Tall) / a.ksplit);

    i32x4 ra, rb;
    {
        uint64_t pa = (uint64_t)a.in.p, pb = (uint64_t)a.w;
        unsigned na = (unsigned)((((int64_t)a.in.n * a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4);
        unsigned nb = (unsigned)((int64_t)a.groups * Tall * a.npad * 128);
        ra = i32x4{(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)na, 0x00020000};
        rb = i32x4{(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)nb, 0x00020000};
    }
    // patch loader: wave w owns pieces w, w+NW, ...; lane -> patch pixel 8*piece + lane/8, physical slot lane%8
    unsigned offP[QP], offB[GB];
    const int iy0 = ty * TH - a.pad, ix0 = tx * TW - a.pad;
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        int pp = 8 * (wave + q * NW) + (lane >> 3);
        int slot = (lane & 7) ^ patch_key<PW, TW>(pp);
        int py = pp / PW, px = pp - py * PW;
        int iy = iy0 + py, ix = ix0 + px;
        bool v = pp < NPIX && iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w;
        offP[q] = v ? (unsigned)(((n * a.in.h + iy) * a.in.w + ix) * a.in.ld + cin_off + slot * 4) * 4u : kOob;
    }
#pragma unroll
    for (int p = 0; p < GB; ++p) {
        int row = 8 * (wave * GB + p) + (lane >> 3);
        int slot = (lane & 7) ^ ((row >> 1) & 7);
        offB[p] = n0 + row < a.npad ? (unsigned)((n0 + row) * 32 + slot * 4) * 4u : kOob;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned ldsB = lds0 + (unsigned)(2 * kPatchF * 4) + (unsigned)(wave * GB * 8) * 128u;

    // all 32-channel rows of block cb -> patch stage cb & 1.  Branch-free (a branch around the asm makes hipcc shuffle the
    // accumulators): `live` false turns every lane out of range, the DMA then writes zeros into a stage nobody reads any more.
    auto issue_patch = [&](int cb, bool live) {
        const unsigned sb = lds0 + (unsigned)(cb & 1) * (unsigned)(kPatchF * 4);
#pragma unroll
        for (int q = 0; q < QP; ++q)
            dma16((offP[q] == kOob || !live) ? kOob : offP[q] + (unsigned)cb * 128u, ra, sb + (unsigned)(wave + q * NW) * 1024u);
    };
    unsigned l_w = (unsigned)(((int64_t)g * Tall + c_begin) * a.npad * 128);
    auto issue_b = [&](int stage) {
#pragma unroll
        for (int p = 0; p < GB; ++p)
            dma16(offB[p] == kOob ? kOob : offB[p] + l_w, rb, ldsB + (unsigned)stage * (unsigned)(kBF * 4) + (unsigned)p * 1024u);
        l_w += (unsigned)a.npad * 128u;
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int nn = n0 + 32 * (TN * wn + j) + li;
        float b = (a.bias && ks == 0 && nn < a.cout_g) ? a.bias[cout_off + nn] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = b;
    }
    // MFMA row li of sub-tile t = TM*wm + i is tile pixel 32 t + li = (py, px); its patch pixel under tap (kh, kw) is
    // ppb[i] + kh*PW + kw
    int ppb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int rr = 32 * (TM * wm + i) + li;
        ppb[i] = (rr / TW) * PW + (rr % TW);
    }
    int swb[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) swb[kb] = ((2 * kb + lh) ^ ((li >> 1) & 7)) * 4;
    const int rowB = (32 * TN * wn + li) * 32;
    f32x16 tot[SER ? TM : 1][SER ? TN : 1];
    int run = 0, next_b = SER ? (int)((int64_t)Tall / a.ksplit) : 0;          // SER: first chunk of the next run
    auto compute = [&](int cb, int tap, int bstage) {
        if constexpr (SER) {
            if (9 * cb + tap == next_b) {                                      // block-uniform: S - 1 times per block
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) { tot[i][j][r] = run == 0 ? acc[i][j][r] : tot[i][j][r] + acc[i][j][r]; acc[i][j][r] = 0.0f; }
                ++run; next_b = (int)(((int64_t)(run + 1) * Tall) / a.ksplit);
            }
        }
        const float *SP = lds + (cb & 1) * kPatchF;
        const float *SB = lds + 2 * kPatchF + bstage * kBF;
        const int kh = tap / 3, toff = kh * PW + (tap - 3 * kh);
        int arow[TM], asw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { int pp = ppb[i] + toff; arow[i] = pp * 32; asw[i] = patch_key<PW, TW>(pp); }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(SP + arow[i] + (((2 * kb + lh) ^ asw[i]) << 2));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(SB + rowB + j * 1024 + swb[kb]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = t == 0 ? af[i].x : (t == 1 ? af[i].y : (t == 2 ? af[i].z : af[i].w));
                        const float bv = t == 0 ? bf[j].x : (t == 1 ? bf[j].y : (t == 2 ? bf[j].z : bf[j].w));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
    };

    int cb = c_begin / 9, tap = c_begin - 9 * cb;
    issue_patch(cb, true);
    issue_b(0);
    for (int chunk = c_begin, st = 0; chunk < T;) {
        const int tap_end = min(9, tap + (T - chunk));
        // first chunk of this channel block: the weights of the next chunk AND the next block's patch go out behind the barrier
        // (the weight fetch is unconditional: past the end of this K run it lands in a stage nobody reads)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (lgkmcnt: see k_conv_dma_p)
        __builtin_amdgcn_s_barrier();
        issue_b(st ^ 1);
        issue_patch(cb + 1, (cb + 1) * 9 < T);
        compute(cb, tap, st);
        ++chunk; st ^= 1;
        for (++tap; tap < tap_end; ++tap, ++chunk, st ^= 1) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            issue_b(st ^ 1);
            compute(cb, tap, st);
        }
        tap = 0; ++cb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing (dead) fetches must land before the block's LDS is released

    // epilogue: the pixel index is computed once per accumulator row and shared by the TN column tiles
    float slope[TN]; int ncol[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        ncol[j] = n0 + 32 * (TN * wn + j) + li;
        slope[j] = (a.slope && ncol[j] < a.cout_g) ? a.slope[cout_off + ncol[j]] : 0.0f;
    }
    // (one row pointer per accumulator: a lane's 16 pixels sit at compile-time (dy, dx) from its first one -- 4 lh + (r & 3) never carries
    // into the next tile row -- so an element's address is pointer + a UNIFORM offset, no per-element 64-bit multiply)
    const int64_t ldo = a.out.ld, ldr = a.res.ld;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int oyb = ty * TH + (32 / TW) * (TM * wm + i), oxb = tx * TW + 4 * lh;
        const int64_t mb = ((int64_t)n * ho + oyb) * wo + oxb;
        float *ob = a.out.p + mb * ldo + cout_off;
        const float *rb = a.res_mode ? a.res.p + mb * ldr + cout_off : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (r & 3) + 8 * (r >> 2), dy = rl / TW, dx = rl % TW;
            if (oyb + dy >= ho || oxb + dx >= wo) continue;
            const int64_t eo = (int64_t)dy * wo + dx, m = mb + eo;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nn = ncol[j];
                if (nn >= a.cout_g) continue;
                float v = acc[i][j][r];
                if constexpr (SER) v = tot[i][j][r] + v;
                if (!SER && a.ksplit > 1) { a.partial[(m * a.ksplit + ks) * a.cout_g + nn] = v; continue; }
                if (a.res_mode == 1) v += rb[eo * ldr + nn];
                v = apply_act(v, a.act, slope[j]);
                if (a.res_mode == 2) v += rb[eo * ldr + nn];
                ob[eo * ldo + nn] = v;
            }
        }
    }
}

// ---- persistent form of k_conv_patch: the NEXT tile's input patch is fetched during the current tile's taps -------------------------
// With 32 or 64 input channels a tile has one or two channel blocks: k_conv_patch fetches the tile's only (first) patch in its
// prologue, the one DMA round trip of the tile that nothing hides (all blocks of a CU run in lock-step), and the plain DMA kernel has
// one chunk of MFMAs (1.7 us at four 128x32 blocks per CU) to hide every HBM miss behind.  Here a block walks several tiles and the
// sequence of (tile, channel block) patches is double-buffered ACROSS tiles: at the first tap of a tile's last channel block the
// loader switches to the next tile and sends its first patch -- nine taps of MFMAs ahead of its use; the weights of the next tile's
// first tap go out behind the barrier of the last tap.  Same chunks, same chain per output.  SER: the serial split-K walk of k_conv_patch
// (runs combined in registers at the run boundaries); parallel split-K layers keep the one-tile-per-block kernel.
// MINW = waves per SIMD the register allocation must allow (launch bound): 4 caps the 8-wave 128 x 128 tile at 128 VGPRs, so that TWO
// blocks (2 x 80 KB of LDS) share a CU instead of one
template <int WM, int WN, int TM, int TN, int TW, bool SER = false, int MINW = 2, bool ILV = (CSM_ILV != 0)>
__global__ __launch_bounds__(64 * WM * WN, MINW) void k_conv_patch_p(ConvArgs a, int tiles_x, int tiles_y, int n_n, int total) {
    constexpr int NW = WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, TH = BM / TW;
    constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW;
    constexpr int NPP = (NPIX + 7) / 8;
    constexpr int QP = (NPP + NW - 1) / NW;
    constexpr int GB = BN / 8 / NW;
    static_assert(GB * 8 * NW == BN && TH * TW == BM && (TW == 16 || TW == 8), "tile shape");
    constexpr int kPPT = (QP + 7) / 8, kPT = (QP + kPPT - 1) / kPPT;      // patch pieces per tap / taps that carry a slice (<= 8)
    // the counted wait `vmcnt(kPPT)` at the tap after a slice assumes that the slice had exactly kPPT pieces behind the weights: a shorter
    // last slice would let a weight DMA be in flight at the barrier
    static_assert(QP % kPPT == 0, "every patch slice must carry kPPT pieces (counted vmcnt)");
    constexpr int kPatchF = QP * NW * 8 * 32, kBF = BN * 32;
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [patch 0][patch 1][B 0][B 1]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int ho = a.out.h, wo = a.out.w;
    const int ncb = a.ncb, Tall = 9 * ncb;
    const int per = (int)(gridDim.x >> 3), x = (int)(blockIdx.x & 7u), i0 = (int)(blockIdx.x >> 3);
    const int q = total >> 3, r = total & 7;
    const int start = x * q + (x < r ? x : r), len = q + (x < r ? 1 : 0);
    if (i0 >= len) return;
    const int per_z = a.m_tiles * n_n, per_img = tiles_x * tiles_y;

    i32x4 ra, rb;
    {
        uint64_t pa = (uint64_t)a.in.p, pb = (uint64_t)a.w;
        unsigned na = (unsigned)((((int64_t)a.in.n * a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4);
        unsigned nb = (unsigned)((int64_t)a.groups * Tall * a.npad * 128);
        ra = i32x4{(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)na, 0x00020000};
        rb = i32x4{(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)nb, 0x00020000};
    }
    auto tile_of = [&](int k, int &mt, int &nt, int &g) {
        const int j = start + k; g = j / per_z;
        rem_to_tile((unsigned)(j - g * per_z), (unsigned)a.m_tiles, (unsigned)n_n, a.ngroup, mt, nt);
    };
    // patch loader (tile being fetched): wave w owns pieces w, w+NW, ...; lane -> patch pixel 8*piece + lane/8, physical slot lane%8
    unsigned offP[QP], offB[GB];
    auto patch_setup = [&](int k, bool live) {
        int mt, nt, g; tile_of(k, mt, nt, g);
        const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, n = mt / per_img;
        const int iy0 = ty * TH - a.pad, ix0 = tx * TW - a.pad, cin_off = g * a.cin_g;
#pragma unroll
        for (int qq = 0; qq < QP; ++qq) {
            int pp = 8 * (wave + qq * NW) + (lane >> 3);
            int slot = (lane & 7) ^ patch_key<PW, TW>(pp);
            int py = pp / PW, px = pp - py * PW;
            int iy = iy0 + py, ix = ix0 + px;
            bool v = live && pp < NPIX && iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w;
            offP[qq] = v ? (unsigned)(((n * a.in.h + iy) * a.in.w + ix) * a.in.ld + cin_off + slot * 4) * 4u : kOob;
        }
    };
    unsigned l_w = 0u;
    auto b_setup = [&](int k, bool live) {
        int mt, nt, g; tile_of(k, mt, nt, g);
        const int n0 = nt * BN;
#pragma unroll
        for (int p = 0; p < GB; ++p) {
            int row = 8 * (wave * GB + p) + (lane >> 3);
            int slot = (lane & 7) ^ ((row >> 1) & 7);
            offB[p] = (live && n0 + row < a.npad) ? (unsigned)((n0 + row) * 32 + slot * 4) * 4u : kOob;
        }
        l_w = (unsigned)((int64_t)g * Tall * a.npad * 128);
    };
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned ldsB = lds0 + (unsigned)(2 * kPatchF * 4) + (unsigned)(wave * GB * 8) * 128u;
    auto issue_patch = [&](int cb, int pstage) {               // all 32-channel rows of block cb of the tile offP describes
        const unsigned sb = lds0 + (unsigned)pstage * (unsigned)(kPatchF * 4);
#pragma unroll
        for (int qq = 0; qq < QP; ++qq)
            dma16(offP[qq] == kOob ? kOob : offP[qq] + (unsigned)cb * 128u, ra, sb + (unsigned)(wave + qq * NW) * 1024u);
    };
    auto issue_b = [&](int stage) {
#pragma unroll
        for (int p = 0; p < GB; ++p)
            dma16(offB[p] == kOob ? kOob : offB[p] + l_w, rb, ldsB + (unsigned)stage * (unsigned)(kBF * 4) + (unsigned)p * 1024u);
        l_w += (unsigned)a.npad * 128u;
    };

    int ppb[TM];
    int abase[TM][3][4];                                        // (patch pixel of MFMA row li) * 32 + swizzled 16-B slot, per tap column kw and k-block
    static_assert(TW == 16, "abase: the swizzle key of a 16-wide tile depends on the patch column only");
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int rr = 32 * (TM * wm + i) + li;
        ppb[i] = (rr / TW) * PW + (rr % TW);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) abase[i][kw][kb] = ppb[i] * 32 + (((2 * kb + lh) ^ ((((rr % TW) + kw) >> 1) & 7)) << 2);
    }
    int swb[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) swb[kb] = ((2 * kb + lh) ^ ((li >> 1) & 7)) * 4;
    const int rowB = (32 * TN * wn + li) * 32;
    f32x16 acc[TM][TN];
    f32x16 tot[SER ? TM : 1][SER ? TN : 1];
    int run = 0, next_b = 0;                                    // SER: first chunk of the next run (reset per tile)
    auto compute = [&](int pstage, int tap, int bstage) {
        const float *SP = lds + pstage * kPatchF;
        const float *SB = lds + 2 * kPatchF + bstage * kBF;
        const int kh = tap / 3, toff = kh * PW + (tap - 3 * kh);
        int arow[TM], asw[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) { int pp = ppb[i] + toff; arow[i] = pp * 32; asw[i] = patch_key<PW, TW>(pp); }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4 *>(SP + arow[i] + (((2 * kb + lh) ^ asw[i]) << 2));
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4 *>(SB + rowB + j * 1024 + swb[kb]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float av = t == 0 ? af[i].x : (t == 1 ? af[i].y : (t == 2 ? af[i].z : af[i].w));
                        const float bv = t == 0 ? bf[j].x : (t == 1 ? bf[j].y : (t == 2 ? bf[j].z : bf[j].w));
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
        }
    };

    patch_setup(i0, true);
    b_setup(i0, true);
    issue_patch(0, 0);
    issue_b(0);
    int ps = 0, st = 0;                                         // stage of the patch / of the weights about to be consumed
    for (int k = i0; k < len; k += per) {
        int mt, nt, g; tile_of(k, mt, nt, g);
        const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, n = mt / per_img;
        const int n0 = nt * BN, cout_off = g * a.cout_g;
        const int kn = k + per;
        const bool more = kn < len;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int nn = n0 + 32 * (TN * wn + j) + li;
            float b = (a.bias && nn < a.cout_g) ? a.bias[cout_off + nn] : 0.0f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) acc[i][j][rr] = b;
        }
        if constexpr (SER) { run = 0; next_b = (int)((int64_t)Tall / a.ksplit); }
        for (int cb = 0; cb < ncb; ++cb, ps ^= 1) {
            const bool last_cb = cb + 1 == ncb;
            // The next patch of the sequence -- block cb + 1 of this tile or (last block) block 0 of the NEXT tile, for which the loader
            // state is switched at tap 0 -- goes out in SLICES of kPPT pieces behind the weights of taps 0 .. kPT - 1.  vmcnt retires in
            // order, so "at most kPPT outstanding" at the next barrier means the weights have landed and the slice may still fly:
            // every slice has two taps of MFMAs to arrive in (HBM misses included) instead of one.  Branch-free: past the end every
            // lane is out of range and the DMA writes zeros into a stage nobody reads.
            auto chunk = [&](auto TAPC) {
                constexpr int tap = decltype(TAPC)::value;
                // The nine taps are straight-line code: hipcc moves the barrier of tap t + 1 up between the last LDS reads of tap t and
                // the MFMAs that consume them, so a wave could pass the barrier with fragment reads still in flight while the next
                // DMA into that stage is issued behind it (a few wrong values in 10^7, seen once in the 8-wide 64 x 64 tile).  The
                // fragment reads of the previous tap must have COMPLETED before this wave arrives at the barrier:
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (tap >= 1 && tap <= kPT) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kPPT) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if constexpr (tap == 0) { if (last_cb) patch_setup(more ? kn : k, more); }
                if constexpr (tap == 8) { if (last_cb) b_setup(more ? kn : k, more); }      // the weights of the next tile's first tap
                const unsigned psb = lds0 + (unsigned)(ps ^ 1) * (unsigned)(kPatchF * 4);
                const unsigned cbo = (unsigned)(last_cb ? 0 : cb + 1) * 128u;
                if constexpr (!ILV) {
                    issue_b(st ^ 1);
                    if constexpr (tap < kPT) {
#pragma unroll
                        for (int qq = tap * kPPT; qq < (tap + 1) * kPPT && qq < QP; ++qq)
                            dma16(offP[qq] == kOob ? kOob : offP[qq] + cbo, ra, psb + (unsigned)(wave + qq * NW) * 1024u);
                    }
                }
                if constexpr (SER) {
                    if (9 * cb + tap == next_b) {                                  // block-uniform: S - 1 times per tile
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
#pragma unroll
                                for (int rr = 0; rr < 16; ++rr) { tot[i][j][rr] = run == 0 ? acc[i][j][rr] : tot[i][j][rr] + acc[i][j][rr]; acc[i][j][rr] = 0.0f; }
                        ++run; next_b = (int)(((int64_t)(run + 1) * Tall) / a.ksplit);
                    }
                }
                if constexpr (!ILV) compute(ps, tap, st);
                else {
                    // the tap's MFMAs with this tap's DMA pieces between the groups (same order as the burst: the weights of the next tap
                    // first, then the patch slice -- the counted vmcnt at the next barrier relies on it); see k_conv_dma_p
                    constexpr int NSL = tap < kPT ? ((tap + 1) * kPPT <= QP ? kPPT : QP - tap * kPPT) : 0;
                    static_assert(GB + kPPT <= 16, "one DMA piece per MFMA group");
                    // fragment addresses = per-lane bases that do not depend on the tile (abase: 12 per accumulator row) + a tap constant + the
                    // stage: the nine unrolled taps used to keep 72 precomputed addresses per accumulator row alive (187 - 233 VGPRs)
                    constexpr int kh = tap / 3, kw = tap - 3 * kh, toff = kh * PW + kw;
                    const float *SP = lds + ps * kPatchF + toff * 32;
                    const float *SB = lds + 2 * kPatchF + st * kBF;
                    float4 af[2][TM], bf[2][TN];
                    auto rd = [&](int kb, int buf) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) af[buf][i] = *reinterpret_cast<const float4 *>(SP + abase[i][kw][kb]);
#pragma unroll
                        for (int j = 0; j < TN; ++j) bf[buf][j] = *reinterpret_cast<const float4 *>(SB + rowB + j * 1024 + swb[kb]);
                    };
                    rd(0, 0);
                    [&]<int... G>(std::integer_sequence<int, G...>) {
                        ([&] {
                            constexpr int kb = G / 4, t = G % 4, buf = kb & 1;
                            if constexpr (G < GB) {
                                dma16(offB[G] == kOob ? kOob : offB[G] + l_w, rb, ldsB + (unsigned)(st ^ 1) * (unsigned)(kBF * 4) + (unsigned)G * 1024u);
                                __builtin_amdgcn_sched_barrier(0);
                            } else if constexpr (G < GB + NSL) {
                                constexpr int qq = tap * kPPT + (G - GB);
                                dma16(offP[qq] == kOob ? kOob : offP[qq] + cbo, ra, psb + (unsigned)(wave + qq * NW) * 1024u);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                            if constexpr (t == 1 && kb < 3) { rd(kb + 1, buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j) {
                                    const float av = t == 0 ? af[buf][i].x : (t == 1 ? af[buf][i].y : (t == 2 ? af[buf][i].z : af[buf][i].w));
                                    const float bv = t == 0 ? bf[buf][j].x : (t == 1 ? bf[buf][j].y : (t == 2 ? bf[buf][j].z : bf[buf][j].w));
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                                }
                            __builtin_amdgcn_sched_barrier(0);
                        }(), ...);
                    }(std::make_integer_sequence<int, 16>{});
                    l_w += (unsigned)a.npad * 128u;
                }
                st ^= 1;
            };
            chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
            chunk(std::integral_constant<int, 3>{}); chunk(std::integral_constant<int, 4>{}); chunk(std::integral_constant<int, 5>{});
            chunk(std::integral_constant<int, 6>{}); chunk(std::integral_constant<int, 7>{}); chunk(std::integral_constant<int, 8>{});
        }
        // epilogue
        float slope[TN]; int ncol[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            ncol[j] = n0 + 32 * (TN * wn + j) + li;
            slope[j] = (a.slope && ncol[j] < a.cout_g) ? a.slope[cout_off + ncol[j]] : 0.0f;
        }
        const int64_t ldo = a.out.ld, ldr = a.res.ld;              // (row pointers once per accumulator, as in k_conv_patch)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int oyb = ty * TH + (32 / TW) * (TM * wm + i), oxb = tx * TW + 4 * lh;
            const int64_t mb = ((int64_t)n * ho + oyb) * wo + oxb;
            float *ob = a.out.p + mb * ldo + cout_off;
            const float *rb = a.res_mode ? a.res.p + mb * ldr + cout_off : nullptr;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int rl = (rr & 3) + 8 * (rr >> 2), dy = rl / TW, dx = rl % TW;
                if (oyb + dy >= ho || oxb + dx >= wo) continue;
                const int64_t eo = (int64_t)dy * wo + dx;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nn = ncol[j];
                    if (nn >= a.cout_g) continue;
                    float v = acc[i][j][rr];
                    if constexpr (SER) v = tot[i][j][rr] + v;
                    if (a.res_mode == 1) v += rb[eo * ldr + nn];
                    v = apply_act(v, a.act, slope[j]);
                    if (a.res_mode == 2) v += rb[eo * ldr + nn];
                    ob[eo * ldo + nn] = v;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing (dead) fetches must land before the block's LDS is released
}

// ---- weights-stationary 3x3 convolution (stride 1, dilation 1, one K run): the block's WEIGHT PANEL stays in LDS --------------------------
// k_conv_patch(_p) stream a weight tile per tap: one barrier and GB DMA pieces every tap, although a layer with few input channels has
// very little weight data -- all nine taps x all channel blocks of a 32-wide column tile are 36.9 KB at 32 input channels and 73.7 KB at 64
// (ISNet / RTMDet 32- and 64-channel stages, the 32-channel groups of ResNeXt, the Ken Burns GridNets).  Here a persistent block of eight
// waves loads its column tile's panel ONCE, then walks 16 x 16 output tiles: only the (18 x 18 x 32-channel) input patch of the next
// (tile, channel block) streams, double-buffered, its DMA pieces interleaved with the MFMAs of taps 0 .. 2, and the only barrier left is the
// one per (tile, channel block) that publishes a patch -- 144 MFMAs per wave between barriers instead of 16, a quarter of the DMA volume.
// Wave w owns output rows 2w, 2w + 1 of the tile (32 pixels) x all 32 TN columns.  Same chunks, same chain per output as every other
// configuration (block-major: channel block outer, taps row-major inner).  Column tile = (group, N tile): grouped convolutions with
// 32-channel groups are the case "one channel block, one N tile per group".  Blocks of one XCD with consecutive ids work on the same M tiles
// for different column tiles, so the second reader of a patch finds it in that XCD's L2.
template <int TN>
__global__ __launch_bounds__(512, 2) void k_conv_ws(ConvArgs a, int tiles_x, int tiles_y, int n_n, int n_ct) {
    constexpr int NW = 8, TW = 16, TH = 16, BN = 32 * TN;
    constexpr int PH = TH + 2, PW = TW + 2, NPIX = PH * PW, NPP = (NPIX + 7) / 8, QP = (NPP + NW - 1) / NW;
    constexpr int kPatchF = (NPP + 1) * 8 * 32;                 // floats per patch stage: NPP pieces + one dump slot for the surplus pieces of the last round
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [patch 0][patch 1][weight panel: chunk][BN rows][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int ho = a.out.h, wo = a.out.w, ncb = a.ncb, Tall = 9 * ncb;
    // block -> (XCD x, column tile ct, position i0 among the `per` blocks of that column tile on this XCD)
    const int x = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3);
    const int ct = slot % n_ct, i0 = slot / n_ct, per = (int)(gridDim.x >> 3) / n_ct;
    const int total = a.m_tiles, q = total >> 3, r = total & 7;
    const int start = x * q + (x < r ? x : r), len = q + (x < r ? 1 : 0);
    const int g = ct / n_n, n0 = (ct - g * n_n) * BN, cin_off = g * a.cin_g, cout_off = g * a.cout_g, per_img = tiles_x * tiles_y;
    if (i0 >= len) return;

    i32x4 ra, rb;
    {
        uint64_t pa = (uint64_t)a.in.p, pb = (uint64_t)a.w;
        unsigned na = (unsigned)((((int64_t)a.in.n * a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4);
        unsigned nb = (unsigned)((int64_t)a.groups * Tall * a.npad * 128);
        ra = i32x4{(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)na, 0x00020000};
        rb = i32x4{(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)nb, 0x00020000};
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned ldsW = lds0 + (unsigned)(2 * kPatchF * 4);
    // ---- the weight panel, once: piece p = 8 rows of chunk p / (BN / 8); rows beyond the layer's padded width come in as zeros
    {
        const int npieces = Tall * (BN / 8);
        const unsigned wbase = (unsigned)((int64_t)g * Tall * a.npad * 128);
        for (int p = wave; p < npieces; p += NW) {
            const int c = p / (BN / 8), row = 8 * (p - c * (BN / 8)) + (lane >> 3);
            const int sl = (lane & 7) ^ ((row >> 1) & 7);
            const unsigned off = n0 + row < a.npad ? wbase + (unsigned)(((c * a.npad + n0 + row) * 32 + sl * 4) * 4) : kOob;
            dma16(off, rb, ldsW + (unsigned)p * 1024u);
        }
    }
    // ---- patch loader (tile being fetched): wave w owns pieces w, w + NW, ...; lane -> patch pixel 8 * piece + lane / 8, physical slot lane % 8
    unsigned offP[QP];
    auto patch_setup = [&](int k, bool live) {
        const int mt = start + k;
        const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, n = mt / per_img;
        const int iy0 = ty * TH - a.pad, ix0 = tx * TW - a.pad;
#pragma unroll
        for (int qq = 0; qq < QP; ++qq) {
            int pp = 8 * (wave + qq * NW) + (lane >> 3);
            int sl = (lane & 7) ^ patch_key<PW, TW>(pp);
            int py = pp / PW, px = pp - py * PW;
            int iy = iy0 + py, ix = ix0 + px;
            bool v = live && pp < NPIX && iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w;
            offP[qq] = v ? (unsigned)(((n * a.in.h + iy) * a.in.w + ix) * a.in.ld + cin_off + sl * 4) * 4u : kOob;
        }
    };
    auto patch_piece = [&](auto QC, int cb, int pstage) {
        constexpr int qq = decltype(QC)::value;
        const int piece = wave + qq * NW;
        dma16(offP[qq] == kOob ? kOob : offP[qq] + (unsigned)cb * 128u, ra,
              lds0 + (unsigned)pstage * (unsigned)(kPatchF * 4) + (unsigned)(piece < NPP ? piece : NPP) * 1024u);
    };
    int abase[3][4];                                            // (patch pixel of MFMA row li) * 32 + swizzled 16-B slot, per tap column and k-block
    {
        const int rr = 32 * wave + li;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) abase[kw][kb] = ((rr / TW) * PW + (rr % TW)) * 32 + (((2 * kb + lh) ^ ((((rr % TW) + kw) >> 1) & 7)) << 2);
    }
    int swb[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) swb[kb] = li * 32 + ((2 * kb + lh) ^ ((li >> 1) & 7)) * 4;
    f32x16 acc[TN];

    patch_setup(i0, true);
    [&]<int... Q>(std::integer_sequence<int, Q...>) { (patch_piece(std::integral_constant<int, Q>{}, 0, 0), ...); }(std::make_integer_sequence<int, QP>{});
    int ps = 0;
    for (int k = i0; k < len; k += per) {
        const int mt = start + k;
        const int tx = mt % tiles_x, ty = (mt / tiles_x) % tiles_y, n = mt / per_img;
        const int kn = k + per;
        const bool more = kn < len;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nn = n0 + 32 * j + li;
            const float b = (a.bias && nn < a.cout_g) ? a.bias[cout_off + nn] : 0.0f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) acc[j][rr] = b;
        }
        for (int cb = 0; cb < ncb; ++cb, ps ^= 1) {
            const bool last_cb = cb + 1 == ncb;
            // this (tile, channel block)'s patch has landed (every wave waits for its own pieces, then the barrier), and everybody has finished
            // reading the other stage (its fragment reads have completed: lgkmcnt) -- it is refilled below
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (last_cb) patch_setup(more ? kn : k, more);
            const int ncbo = last_cb ? 0 : cb + 1;
            const float *SW = lds + 2 * kPatchF + (cb * 9) * (BN * 32);
            auto tapf = [&](auto TAPC) {
                constexpr int tap = decltype(TAPC)::value, kh = tap / 3, kw = tap - 3 * kh;
                const float *SP = lds + ps * kPatchF + (kh * PW + kw) * 32;
                const float *SB = SW + tap * (BN * 32);
                float4 af[2], bf[2][TN];
                auto rd = [&](int kb, int buf) {
                    af[buf] = *reinterpret_cast<const float4 *>(SP + abase[kw][kb]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[buf][j] = *reinterpret_cast<const float4 *>(SB + j * 1024 + swb[kb]);
                };
                rd(0, 0);
                [&]<int... G>(std::integer_sequence<int, G...>) {
                    ([&] {
                        constexpr int kb = G / 4, t = G % 4, buf = kb & 1;
                        // the next patch goes out four pieces per tap (behind MFMA groups 0, 4, 8, 12 of taps 0, 1, ...)
                        if constexpr (t == 0 && 4 * tap + kb < QP) { patch_piece(std::integral_constant<int, 4 * tap + kb>{}, ncbo, ps ^ 1); __builtin_amdgcn_sched_barrier(0); }
                        if constexpr (t == 1 && kb < 3) { rd(kb + 1, buf ^ 1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            const float av = t == 0 ? af[buf].x : (t == 1 ? af[buf].y : (t == 2 ? af[buf].z : af[buf].w));
                            const float bv = t == 0 ? bf[buf][j].x : (t == 1 ? bf[buf][j].y : (t == 2 ? bf[buf][j].z : bf[buf][j].w));
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }(), ...);
                }(std::make_integer_sequence<int, 16>{});
            };
            static_assert(QP <= 36, "the patch pieces must fit the nine taps");
            tapf(std::integral_constant<int, 0>{}); tapf(std::integral_constant<int, 1>{}); tapf(std::integral_constant<int, 2>{});
            tapf(std::integral_constant<int, 3>{}); tapf(std::integral_constant<int, 4>{}); tapf(std::integral_constant<int, 5>{});
            tapf(std::integral_constant<int, 6>{}); tapf(std::integral_constant<int, 7>{}); tapf(std::integral_constant<int, 8>{});
        }
        // epilogue: lane holds column li of each 32-wide column tile, tile pixels 32 wave + (r & 3) + 8 (r >> 2) + 4 lh
        float slope[TN]; int ncol[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            ncol[j] = n0 + 32 * j + li;
            slope[j] = (a.slope && ncol[j] < a.cout_g) ? a.slope[cout_off + ncol[j]] : 0.0f;
        }
        const int64_t ldo = a.out.ld, ldr = a.res.ld;              // (row pointer once per tile, as in k_conv_patch)
        const int oyb = ty * TH + (32 / TW) * wave, oxb = tx * TW + 4 * lh;
        const int64_t mb = ((int64_t)n * ho + oyb) * wo + oxb;
        float *ob = a.out.p + mb * ldo + cout_off;
        const float *rb = a.res_mode ? a.res.p + mb * ldr + cout_off : nullptr;
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int rl = (rr & 3) + 8 * (rr >> 2), dy = rl / TW, dx = rl % TW;
            if (oyb + dy >= ho || oxb + dx >= wo) continue;
            const int64_t eo = (int64_t)dy * wo + dx;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nn = ncol[j];
                if (nn >= a.cout_g) continue;
                float v = acc[j][rr];
                if (a.res_mode == 1) v += rb[eo * ldr + nn];
                v = apply_act(v, a.act, slope[j]);
                if (a.res_mode == 2) v += rb[eo * ldr + nn];
                ob[eo * ldo + nn] = v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing (dead) fetches must land before the block's LDS is released
}

// ---- stem convolution (cin padded to 4: RTMDet / ISNet / LeReS first layers) -----------------------------------------------
// The generic kernels spend one 32-channel chunk per tap with 4 channels in use.  Here K is packed (tap, channel): a chunk holds
// 8 taps x 4 channels (weights packed to match on the host, program.py::pack_stem_weights), 7 chunks instead of 49 for the
// 7x7.  Each loader thread fetches one pixel's 4 channels for one tap (one float4) and scatters them into the 8-block positions
// that make the MFMA lane order 0,4,1,5,2,6,3,7 walk tap 2j's channels 0..3 and then tap 2j+1's -- the contract's chain.
// 64x64 tile, 2x2 waves, register-staged (the permutation rules out LDS-DMA), one LDS buffer; HBM-bound for the 3x3 stems.
__global__ __launch_bounds__(256) void k_conv_stem(ConvArgs a) {
    constexpr int BM = 64, BN = 64;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * kLdsLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
    int mt, ntile, zz;
    block_to_tile(mt, ntile, zz);
    const int m0 = mt * BM, n0 = ntile * BN;
    const int ho = a.out.h, wo = a.out.w, ntaps = a.kh * a.kw, nck = (ntaps + 7) >> 3;
    // A loader: thread -> rows (tid>>3) and (tid>>3)+32, tap slot j = tid&7 of the chunk
    const int j = tid & 7;
    const float *rowp[2]; int iy0[2], ix0[2]; bool rv[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        int m = m0 + (tid >> 3) + 32 * it;
        rv[it] = m < a.M;
        int mm = rv[it] ? m : 0;
        int n = mm / (ho * wo), rem = mm - n * ho * wo;
        int oy = rem / wo, ox = rem - oy * wo;
        iy0[it] = oy * a.stride - a.pad; ix0[it] = ox * a.stride - a.pad;
        rowp[it] = a.in.p + (int64_t)n * a.in.h * a.in.w * a.in.ld;
    }
    const float *wp[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) wp[it] = a.w + (int64_t)(n0 + (tid >> 3) + 32 * it) * 32 + j * 4;
    float4 ra[2], rb[2];
    auto gload = [&](int c) {
        const int t = 8 * c + j, kh = t / a.kw, kw = t - kh * a.kw;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int iy = iy0[it] + kh * a.dil, ix = ix0[it] + kw * a.dil;
            bool v = rv[it] && t < ntaps && iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w;
            ra[it] = v ? *reinterpret_cast<const float4 *>(rowp[it] + ((int64_t)iy * a.in.w + ix) * a.in.ld) : make_float4(0.f, 0.f, 0.f, 0.f);
            bool vb = n0 + (tid >> 3) + 32 * it < a.npad;
            rb[it] = vb ? *reinterpret_cast<const float4 *>(wp[it] + (int64_t)c * a.npad * 32) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float *row = lds + ((tid >> 3) + 32 * it) * kLdsLd + 8 * (j >> 1) + 2 * (j & 1);
            *reinterpret_cast<float2 *>(row) = make_float2(ra[it].x, ra[it].z);          // (c0, c2) of this tap
            *reinterpret_cast<float2 *>(row + 4) = make_float2(ra[it].y, ra[it].w);      // (c1, c3)
            *reinterpret_cast<float4 *>(lds + (BM + (tid >> 3) + 32 * it) * kLdsLd + j * 4) = rb[it];
        }
    };
    f32x16 acc;
    {
        int n = n0 + 32 * wn + li;
        float b = (a.bias && n < a.cout_g) ? a.bias[n] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = b;
    }
    const float *A = lds + (32 * wm + li) * kLdsLd + 4 * lh;
    const float *B = lds + (BM + 32 * wn + li) * kLdsLd + 4 * lh;
    gload(0);
    for (int c = 0; c < nck; ++c) {
        lstore();
        __syncthreads();
        if (c + 1 < nck) gload(c + 1);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const float4 af = *reinterpret_cast<const float4 *>(A + kb * 8);
            const float4 bf = *reinterpret_cast<const float4 *>(B + kb * 8);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    const int n = n0 + 32 * wn + li;
    if (n >= a.cout_g) return;
    const float slope = a.slope ? a.slope[n] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int m = m0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= a.M) continue;
        float v = acc[r];
        if (a.res_mode == 1) v += a.res.p[(int64_t)m * a.res.ld + n];
        v = apply_act(v, a.act, slope);
        if (a.res_mode == 2) v += a.res.p[(int64_t)m * a.res.ld + n];
        a.out.p[(int64_t)m * a.out.ld + n] = v;
    }
}

// ---- narrow-output convolution (cout <= 4, groups == 1, no split-K): ISNet side outputs / LeReS last conv -------------------
// An N = 1 output wastes 31/32 of an MFMA tile; this is the same fmaf chain (32-channel blocks, taps row-major, 8-channel
// sub-blocks in the order 0,4,1,5,2,6,3,7; out-of-image taps contribute exact zeros) evaluated one output pixel per lane on the VALU.  A lane's chain
// cannot be shared between lanes, so a lane reads whole pixels: straight from global that is 64 scattered 16-B pieces per
// load instruction (TA-bound, measured no faster than the MFMA path); instead the block stages its input region
// (TH x 32 outputs + halo, all channels) into LDS with coalesced loads -- pixel pitch cin+4 floats makes the per-lane
// ds_read_b128 conflict-free -- and the weights too.  HBM-bound: the input is read once.
template <int NOUT, int TH>
__global__ __launch_bounds__(32 * TH) void k_conv_narrow(ConvArgs a, int tiles_x, int tiles_y, int rh, int rw) {
    constexpr int TW = 32, NT = 32 * TH;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int T = a.kh * a.kw * a.ncb, pitch = a.cin_g + 4;
    float *wl = sm;                                   // [cb][tap][NOUT][32] (= chunk order of the packed weights)
    float *xl = sm + ((T * NOUT * 32 + 3) & ~3);      // [rh][rw][pitch]
    const int tid = threadIdx.x;
    for (int i = tid; i < T * NOUT * 32; i += NT) {
        int c = i & 31, n = (i >> 5) % NOUT, ch = i / (32 * NOUT);
        wl[i] = n < a.cout_g ? a.w[((int64_t)ch * a.npad + n) * 32 + c] : 0.0f;
    }
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * a.stride - a.pad, ix0 = ox0 * a.stride - a.pad;
    const int c4n = a.cin_g >> 2, total = rh * rw * c4n;
    for (int i0 = tid; i0 < total; i0 += NT * 8) {               // 8 loads in flight per lane before the first LDS store
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int i = i0 + u * NT;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < total) {
                int c4 = i % c4n, pix = i / c4n;
                int ry = pix / rw, rx = pix - ry * rw;
                int iy = iy0 + ry, ix = ix0 + rx;
                if (iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w)
                    v[u] = *reinterpret_cast<const float4 *>(a.in.p + ((int64_t)(n * a.in.h + iy) * a.in.w + ix) * a.in.ld + c4 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            int i = i0 + u * NT;
            if (i < total) *reinterpret_cast<float4 *>(xl + (i / c4n) * pitch + (i % c4n) * 4) = v[u];
        }
    }
    __syncthreads();
    const int ly = tid >> 5, lx = tid & 31;
    const int oy = oy0 + ly, ox = ox0 + lx;
    if (oy >= a.out.h || ox >= a.out.w) return;
    float acc[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) acc[j] = (a.bias && j < a.cout_g) ? a.bias[j] : 0.0f;
    for (int cb = 0; cb < a.ncb; ++cb)
        for (int kh = 0; kh < a.kh; ++kh)
            for (int kw = 0; kw < a.kw; ++kw) {
                const float *P = xl + ((ly * a.stride + kh * a.dil) * rw + lx * a.stride + kw * a.dil) * pitch;
                const float *W = wl + (cb * a.kh * a.kw + kh * a.kw + kw) * NOUT * 32;
#pragma unroll 4
                for (int c8 = cb * 32; c8 < cb * 32 + 32 && c8 < a.cin_g; c8 += 8) {   // cin_g % 4 == 0; a trailing half block is 4 channels
                    const float4 lo = *reinterpret_cast<const float4 *>(P + c8);
                    const float4 hi = c8 + 4 < a.cin_g ? *reinterpret_cast<const float4 *>(P + c8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float *w8 = W + (c8 & 31);
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) {
                        const float *w = w8 + j * 32;
                        float v = acc[j];
                        v = fmaf(lo.x, w[0], v); v = fmaf(hi.x, w[4], v);
                        v = fmaf(lo.y, w[1], v); v = fmaf(hi.y, w[5], v);
                        v = fmaf(lo.z, w[2], v); v = fmaf(hi.z, w[6], v);
                        v = fmaf(lo.w, w[3], v); v = fmaf(hi.w, w[7], v);
                        acc[j] = v;
                    }
                }
            }
    const int64_t m = ((int64_t)n * a.out.h + oy) * a.out.w + ox;
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        if (j >= a.cout_g) break;
        float v = acc[j];
        float slope = a.slope ? a.slope[j] : 0.0f;
        if (a.res_mode == 1) v += a.res.p[m * a.res.ld + j];
        v = apply_act(v, a.act, slope);
        if (a.res_mode == 2) v += a.res.p[m * a.res.ld + j];
        a.out.p[m * a.out.ld + j] = v;
    }
}

// The same kernel with the input staged ONE 32-channel block at a time (the chain order is block-major anyway): the region of an 8 x 32
// output tile then takes 49 KB instead of 92 KB at 64 channels, three 256-thread blocks share a CU, and each block has 43 KB of loads in
// flight per staging step instead of 16 KB -- the whole-region form ran the 64 -> 1 side output of ISNet at 1.0 TB/s (0.52 ms at batch 16:
// 531 MB of input), bound by bytes in flight, not by arithmetic (576 fmaf per pixel = 15 us of VALU) or LDS.
template <int NOUT>
__global__ __launch_bounds__(256) void k_conv_narrow_cb(ConvArgs a, int tiles_x, int tiles_y, int rh, int rw) {
    constexpr int TH = 8, TW = 32, NT = 256, PITCH = 36;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int taps = a.kh * a.kw, T = taps * a.ncb;
    float *wl = sm;                                   // [cb][tap][NOUT][32] (= chunk order of the packed weights)
    float *xl = sm + ((T * NOUT * 32 + 3) & ~3);      // [rh][rw][PITCH]: the current channel block of the input region
    const int tid = threadIdx.x;
    for (int i = tid; i < T * NOUT * 32; i += NT) {
        int c = i & 31, n = (i >> 5) % NOUT, ch = i / (32 * NOUT);
        wl[i] = n < a.cout_g ? a.w[((int64_t)ch * a.npad + n) * 32 + c] : 0.0f;
    }
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * a.stride - a.pad, ix0 = ox0 * a.stride - a.pad;
    const int ly = tid >> 5, lx = tid & 31;
    const int oy = oy0 + ly, ox = ox0 + lx;
    const bool live = oy < a.out.h && ox < a.out.w;
    float acc[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; ++j) acc[j] = (a.bias && j < a.cout_g) ? a.bias[j] : 0.0f;
    for (int cb = 0; cb < a.ncb; ++cb) {
        const int cw = min(32, a.cin_g - 32 * cb), c4n = cw >> 2, total = rh * rw * c4n;
        __syncthreads();                                         // the previous block's reads are done (first pass: nothing)
        for (int i0 = tid; i0 < total; i0 += NT * 8) {           // 8 loads in flight per lane before the first LDS store
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int i = i0 + u * NT;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < total) {
                    int c4 = i % c4n, pix = i / c4n;
                    int ry = pix / rw, rx = pix - ry * rw;
                    int iy = iy0 + ry, ix = ix0 + rx;
                    if (iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w)
                        v[u] = *reinterpret_cast<const float4 *>(a.in.p + ((int64_t)(n * a.in.h + iy) * a.in.w + ix) * a.in.ld + 32 * cb + c4 * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                int i = i0 + u * NT;
                if (i < total) *reinterpret_cast<float4 *>(xl + (i / c4n) * PITCH + (i % c4n) * 4) = v[u];
            }
        }
        __syncthreads();
        if (!live) continue;
        for (int kh = 0; kh < a.kh; ++kh)
            for (int kw = 0; kw < a.kw; ++kw) {
                const float *P = xl + ((ly * a.stride + kh * a.dil) * rw + lx * a.stride + kw * a.dil) * PITCH;
                const float *W = wl + (cb * taps + kh * a.kw + kw) * NOUT * 32;
#pragma unroll 4
                for (int c8 = 0; c8 < cw; c8 += 8) {             // cin_g % 4 == 0; a trailing half block is 4 channels
                    const float4 lo = *reinterpret_cast<const float4 *>(P + c8);
                    const float4 hi = c8 + 4 < cw ? *reinterpret_cast<const float4 *>(P + c8 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float *w8 = W + c8;
#pragma unroll
                    for (int j = 0; j < NOUT; ++j) {
                        const float *w = w8 + j * 32;
                        float v = acc[j];
                        v = fmaf(lo.x, w[0], v); v = fmaf(hi.x, w[4], v);
                        v = fmaf(lo.y, w[1], v); v = fmaf(hi.y, w[5], v);
                        v = fmaf(lo.z, w[2], v); v = fmaf(hi.z, w[6], v);
                        v = fmaf(lo.w, w[3], v); v = fmaf(hi.w, w[7], v);
                        acc[j] = v;
                    }
                }
            }
    }
    if (!live) return;
    const int64_t m = ((int64_t)n * a.out.h + oy) * a.out.w + ox;
#pragma unroll
    for (int j = 0; j < NOUT; ++j) {
        if (j >= a.cout_g) break;
        float v = acc[j];
        float slope = a.slope ? a.slope[j] : 0.0f;
        if (a.res_mode == 1) v += a.res.p[m * a.res.ld + j];
        v = apply_act(v, a.act, slope);
        if (a.res_mode == 2) v += a.res.p[m * a.res.ld + j];
        a.out.p[m * a.out.ld + j] = v;
    }
}

// LDS bytes of k_conv_narrow for a TH-row tile; 0 = does not fit
static size_t narrow_lds(const ConvArgs &a, int TH, int *rh_out, int *rw_out) {
    int nout = a.cout_g == 1 ? 1 : 4;
    int rh = (TH - 1) * a.stride + (a.kh - 1) * a.dil + 1, rw = 31 * a.stride + (a.kw - 1) * a.dil + 1;
    size_t fl = (((size_t)a.kh * a.kw * a.ncb * nout * 32 + 3) & ~(size_t)3) + (size_t)rh * rw * (a.cin_g + 4);
    if (rh_out) { *rh_out = rh; *rw_out = rw; }
    return fl * 4 <= 150 * 1024 ? fl * 4 : 0;
}

template <int NOUT, int TH>
static int launch_narrow_t(const ConvArgs &a, size_t lds, int rh, int rw, hipStream_t st) {
    static KernelPrep prep;
    (void)prep.ensure([&] { return prepare_kernel(&k_conv_narrow<NOUT, TH>, 32 * TH, (size_t)150 * 1024); });
    int tiles_x = (a.out.w + 31) / 32, tiles_y = (a.out.h + TH - 1) / TH;
    k_conv_narrow<NOUT, TH><<<(unsigned)(tiles_x * tiles_y * a.out.n), 32 * TH, lds, st>>>(a, tiles_x, tiles_y, rh, rw);
    return csm::check_launch("k_conv_narrow");
}

template <int NOUT>
static int launch_narrow_cb_t(const ConvArgs &a, size_t lds, int rh, int rw, hipStream_t st) {
    static KernelPrep prep;
    (void)prep.ensure([&] { return prepare_kernel(&k_conv_narrow_cb<NOUT>, 256, (size_t)64 * 1024); });
    int tiles_x = (a.out.w + 31) / 32, tiles_y = (a.out.h + 7) / 8;
    k_conv_narrow_cb<NOUT><<<(unsigned)(tiles_x * tiles_y * a.out.n), 256, lds, st>>>(a, tiles_x, tiles_y, rh, rw);
    return csm::check_launch("k_conv_narrow_cb");
}

static int launch_narrow(const ConvArgs &a, hipStream_t st) {
    int rh, rw;
    if (a.ncb > 1) {                                  // more than one channel block: stage them one at a time (three blocks per CU)
        const int nout = a.cout_g == 1 ? 1 : 4;
        rh = 7 * a.stride + (a.kh - 1) * a.dil + 1; rw = 31 * a.stride + (a.kw - 1) * a.dil + 1;
        const size_t fl = (((size_t)a.kh * a.kw * a.ncb * nout * 32 + 3) & ~(size_t)3) + (size_t)rh * rw * 36;
        if (fl * 4 <= 54400) return a.cout_g == 1 ? launch_narrow_cb_t<1>(a, fl * 4, rh, rw, st) : launch_narrow_cb_t<4>(a, fl * 4, rh, rw, st);
    }
    size_t lds = narrow_lds(a, 8, &rh, &rw);
    if (lds && lds <= 50 * 1024) return a.cout_g == 1 ? launch_narrow_t<1, 8>(a, lds, rh, rw, st) : launch_narrow_t<4, 8>(a, lds, rh, rw, st);
    lds = narrow_lds(a, 4, &rh, &rw);
    return a.cout_g == 1 ? launch_narrow_t<1, 4>(a, lds, rh, rw, st) : launch_narrow_t<4, 4>(a, lds, rh, rw, st);
}

// split-K tail: v = ((p0 + p1) + p2) + ... in run order, then the usual epilogue
__global__ __launch_bounds__(256) void k_splitk_reduce(ConvArgs a) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.M * a.cout_g) return;
    int n = (int)(idx % a.cout_g); int64_t m = idx / a.cout_g;
    const float *P = a.partial + m * a.ksplit * a.cout_g + n;
    float v = P[0];
    for (int s = 1; s < a.ksplit; ++s) v += P[(int64_t)s * a.cout_g];
    float slope = a.slope ? a.slope[n] : 0.0f;
    if (a.res_mode == 1) v += a.res.p[m * a.res.ld + n];
    v = apply_act(v, a.act, slope);
    if (a.res_mode == 2) v += a.res.p[m * a.res.ld + n];
    a.out.p[m * a.out.ld + n] = v;
}

// depthwise conv (RTMDet CSPNeXt 5x5): lane = (pixel, 4 channels); weights [tap][C]; fmaf chain over taps
struct DwArgs { View in, out; const float *w, *bias, *slope; int kh, kw, stride, pad, dil, act; };
__global__ __launch_bounds__(256) void k_dwconv(DwArgs a) {
    const int c4n = a.out.c >> 2;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)a.out.n * a.out.h * a.out.w * c4n;
    if (idx >= total) return;
    int c = (int)(idx % c4n) * 4; int64_t pix = idx / c4n;
    int ox = (int)(pix % a.out.w); int64_t t = pix / a.out.w; int oy = (int)(t % a.out.h); int n = (int)(t / a.out.h);
    float4 acc = a.bias ? *reinterpret_cast<const float4 *>(a.bias + c) : make_float4(0, 0, 0, 0);
    for (int kh = 0; kh < a.kh; ++kh) {
        int iy = oy * a.stride - a.pad + kh * a.dil;
        if (iy < 0 || iy >= a.in.h) continue;
        for (int kw = 0; kw < a.kw; ++kw) {
            int ix = ox * a.stride - a.pad + kw * a.dil;
            if (ix < 0 || ix >= a.in.w) continue;
            float4 x = *reinterpret_cast<const float4 *>(a.in.p + ((int64_t)(n * a.in.h + iy) * a.in.w + ix) * a.in.ld + c);
            float4 w = *reinterpret_cast<const float4 *>(a.w + (int64_t)(kh * a.kw + kw) * a.out.c + c);
            acc.x = fmaf(x.x, w.x, acc.x); acc.y = fmaf(x.y, w.y, acc.y);
            acc.z = fmaf(x.z, w.z, acc.z); acc.w = fmaf(x.w, w.w, acc.w);
        }
    }
    float4 s = a.slope ? *reinterpret_cast<const float4 *>(a.slope + c) : make_float4(0, 0, 0, 0);
    acc.x = apply_act(acc.x, a.act, s.x); acc.y = apply_act(acc.y, a.act, s.y);
    acc.z = apply_act(acc.z, a.act, s.z); acc.w = apply_act(acc.w, a.act, s.w);
    *reinterpret_cast<float4 *>(a.out.p + pix * a.out.ld + c) = acc;
}


// depthwise conv, stride 1 / dilation 1 (CSPNeXt 5x5): same chain as k_dwconv, but the block first stages its input region
// (8x16 outputs + halo, 32 channels) in LDS -- every input element is used by kh*kw outputs, and from global that re-use
// came out of L2 (measured ~12 TB/s of L2 traffic, L2-bound); from LDS the kernel is HBM-bound.  Out-of-image taps add
// exact zeros.  Thread = (channel quad, 4 output pixels).
__global__ __launch_bounds__(256) void k_dwconv_lds(DwArgs a, int tiles_x, int tiles_y) {
    constexpr int TH = 8, TW = 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int rh = TH + a.kh - 1, rw = TW + a.kw - 1, taps = a.kh * a.kw;
    float *wl = sm;                       // [tap][32]
    float *xl = sm + taps * 32;           // [rh*rw][32]
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int c0 = blockIdx.y * 32;
    const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
    for (int i = tid; i < taps * 32; i += 256) wl[i] = a.w[(int64_t)(i >> 5) * a.out.c + c0 + (i & 31)];
    const int total = rh * rw * 8;
    for (int i0 = tid; i0 < total; i0 += 256 * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int i = i0 + u * 256;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < total) {
                int c4 = i & 7, pix = i >> 3;
                int ry = pix / rw, rx = pix - ry * rw;
                int iy = iy0 + ry, ix = ix0 + rx;
                if (iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w)
                    v[u] = *reinterpret_cast<const float4 *>(a.in.p + ((int64_t)(n * a.in.h + iy) * a.in.w + ix) * a.in.ld + c0 + c4 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int i = i0 + u * 256;
            if (i < total) *reinterpret_cast<float4 *>(xl + (i >> 3) * 32 + (i & 7) * 4) = v[u];
        }
    }
    __syncthreads();
    const int c4 = tid & 7, p0 = tid >> 3;              // pixels p0 + 32*j of the 8x16 tile
    const float4 bias = a.bias ? *reinterpret_cast<const float4 *>(a.bias + c0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 sl = a.slope ? *reinterpret_cast<const float4 *>(a.slope + c0 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = bias;
    for (int kh = 0; kh < a.kh; ++kh)
        for (int kw = 0; kw < a.kw; ++kw) {
            const float4 w = *reinterpret_cast<const float4 *>(wl + (kh * a.kw + kw) * 32 + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pix = p0 + 32 * j, py = pix >> 4, px = pix & 15;
                const float4 x = *reinterpret_cast<const float4 *>(xl + ((py + kh) * rw + px + kw) * 32 + c4 * 4);
                acc[j].x = fmaf(x.x, w.x, acc[j].x); acc[j].y = fmaf(x.y, w.y, acc[j].y);
                acc[j].z = fmaf(x.z, w.z, acc[j].z); acc[j].w = fmaf(x.w, w.w, acc[j].w);
            }
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pix = p0 + 32 * j, oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
        if (oy >= a.out.h || ox >= a.out.w) continue;
        float4 v = acc[j];
        v.x = apply_act(v.x, a.act, sl.x); v.y = apply_act(v.y, a.act, sl.y);
        v.z = apply_act(v.z, a.act, sl.z); v.w = apply_act(v.w, a.act, sl.w);
        *reinterpret_cast<float4 *>(a.out.p + ((int64_t)(n * a.out.h + oy) * a.out.w + ox) * a.out.ld + c0 + c4 * 4) = v;
    }
}

// max pooling (window clipped to the input; ceil_mode handled by the host-computed output size)
__global__ __launch_bounds__(256) void k_maxpool(View in, View out, int k, int stride, int pad) {
    const int c4n = out.c >> 2;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)out.n * out.h * out.w * c4n;
    if (idx >= total) return;
    int c = (int)(idx % c4n) * 4; int64_t pix = idx / c4n;
    int ox = (int)(pix % out.w); int64_t t = pix / out.w; int oy = (int)(t % out.h); int n = (int)(t / out.h);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int ky = 0; ky < k; ++ky) {
        int iy = oy * stride - pad + ky;
        if (iy < 0 || iy >= in.h) continue;
        for (int kx = 0; kx < k; ++kx) {
            int ix = ox * stride - pad + kx;
            if (ix < 0 || ix >= in.w) continue;
            float4 x = *reinterpret_cast<const float4 *>(in.p + ((int64_t)(n * in.h + iy) * in.w + ix) * in.ld + c);
            m.x = fmaxf(m.x, x.x); m.y = fmaxf(m.y, x.y); m.z = fmaxf(m.z, x.z); m.w = fmaxf(m.w, x.w);
        }
    }
    *reinterpret_cast<float4 *>(out.p + pix * out.ld + c) = m;
}

// torch upsample_bilinear2d index/lambda (aten UpSample.h: area_pixel_compute_source_index + guard)
__device__ __forceinline__ void src_index(int dst, int in_size, int out_size, float scale, bool align, int &i0, int &i1,
                                          float &l0, float &l1) {
    if (in_size == out_size) { i0 = i1 = dst; l0 = 1.0f; l1 = 0.0f; return; }
    float real;
    if (align) real = scale * (float)dst;
    else { real = scale * ((float)dst + 0.5f) - 0.5f; if (real < 0.0f) real = 0.0f; }
    i0 = min((int)real, in_size - 1);
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = fminf(fmaxf(real - (float)i0, 0.0f), 1.0f);
    l0 = 1.0f - l1;
}

// bilinear resize: VEC = 4 handles 4 channels per lane with 16-byte accesses (c, pitches and bases 16-byte aligned),
// VEC = 1 is the generic path (single-channel side outputs).  Same expression per element in both.
template <int VEC>
__global__ __launch_bounds__(256) void k_bilinear(View in, View out, int align, float sh, float sw, int act, const float *__restrict__ slope) {
    const int cv = out.c / VEC;
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)out.n * out.h * out.w * cv;
    if (idx >= total) return;
    int c = (int)(idx % cv) * VEC; int64_t pix = idx / cv;
    int ox = (int)(pix % out.w); int64_t t = pix / out.w; int oy = (int)(t % out.h); int n = (int)(t / out.h);
    int y0, y1, x0, x1; float hl0, hl1, wl0, wl1;
    src_index(oy, in.h, out.h, sh, align != 0, y0, y1, hl0, hl1);
    src_index(ox, in.w, out.w, sw, align != 0, x0, x1, wl0, wl1);
    const float *P = in.p + (int64_t)n * in.h * in.w * in.ld + c;
    const float *a00 = P + ((int64_t)y0 * in.w + x0) * in.ld, *a01 = P + ((int64_t)y0 * in.w + x1) * in.ld;
    const float *a10 = P + ((int64_t)y1 * in.w + x0) * in.ld, *a11 = P + ((int64_t)y1 * in.w + x1) * in.ld;
    float *O = out.p + pix * out.ld + c;
    if (VEC == 4) {
        float4 p00 = *reinterpret_cast<const float4 *>(a00), p01 = *reinterpret_cast<const float4 *>(a01);
        float4 p10 = *reinterpret_cast<const float4 *>(a10), p11 = *reinterpret_cast<const float4 *>(a11);
        float4 r;
        r.x = hl0 * (wl0 * p00.x + wl1 * p01.x) + hl1 * (wl0 * p10.x + wl1 * p11.x);
        r.y = hl0 * (wl0 * p00.y + wl1 * p01.y) + hl1 * (wl0 * p10.y + wl1 * p11.y);
        r.z = hl0 * (wl0 * p00.z + wl1 * p01.z) + hl1 * (wl0 * p10.z + wl1 * p11.z);
        r.w = hl0 * (wl0 * p00.w + wl1 * p01.w) + hl1 * (wl0 * p10.w + wl1 * p11.w);
        if (act) {
            r.x = apply_act(r.x, act, slope ? slope[c] : 0.0f); r.y = apply_act(r.y, act, slope ? slope[c + 1] : 0.0f);
            r.z = apply_act(r.z, act, slope ? slope[c + 2] : 0.0f); r.w = apply_act(r.w, act, slope ? slope[c + 3] : 0.0f);
        }
        *reinterpret_cast<float4 *>(O) = r;
    } else {
        const float r = hl0 * (wl0 * a00[0] + wl1 * a01[0]) + hl1 * (wl0 * a10[0] + wl1 * a11[0]);
        O[0] = act ? apply_act(r, act, slope ? slope[c] : 0.0f) : r;
    }
}

// The same resize with the output ROW as the block coordinate (grid = (runs of 256 (pixel, 4-channel) pairs, out.h, out.n)): the sample
// and the row's source rows / weights are wave-uniform, the column index needs one magic-number division -- k_bilinear<4> spends most of
// its instructions in three 64-bit divisions per output word (3.5 TB/s in + out on the 2x decoder upsamplings; this form: see
// profiles/r06_elementwise.txt).  Same expressions per element, same bits.
__global__ __launch_bounds__(256) void k_bilinear_rows(View in, View out, int align, float sh, float sw, int act, const float *__restrict__ slope,
                                                       unsigned cv_mul, unsigned cv_shr) {
    const int cv = out.c >> 2, j = blockIdx.x * 256 + threadIdx.x;
    if (j >= out.w * cv) return;
    const int oy = blockIdx.y, n = blockIdx.z;
    const int ox = (int)fast_div((unsigned)j, cv_mul, cv_shr), c = (j - ox * cv) * 4;
    int y0, y1, x0, x1; float hl0, hl1, wl0, wl1;
    src_index(oy, in.h, out.h, sh, align != 0, y0, y1, hl0, hl1);
    src_index(ox, in.w, out.w, sw, align != 0, x0, x1, wl0, wl1);
    const float *P = in.p + (int64_t)n * in.h * in.w * in.ld + c;
    const float *r0 = P + (int64_t)y0 * in.w * in.ld, *r1 = P + (int64_t)y1 * in.w * in.ld;
    const float4 p00 = *reinterpret_cast<const float4 *>(r0 + x0 * in.ld), p01 = *reinterpret_cast<const float4 *>(r0 + x1 * in.ld);
    const float4 p10 = *reinterpret_cast<const float4 *>(r1 + x0 * in.ld), p11 = *reinterpret_cast<const float4 *>(r1 + x1 * in.ld);
    float4 r;
    r.x = hl0 * (wl0 * p00.x + wl1 * p01.x) + hl1 * (wl0 * p10.x + wl1 * p11.x);
    r.y = hl0 * (wl0 * p00.y + wl1 * p01.y) + hl1 * (wl0 * p10.y + wl1 * p11.y);
    r.z = hl0 * (wl0 * p00.z + wl1 * p01.z) + hl1 * (wl0 * p10.z + wl1 * p11.z);
    r.w = hl0 * (wl0 * p00.w + wl1 * p01.w) + hl1 * (wl0 * p10.w + wl1 * p11.w);
    if (act) {
        r.x = apply_act(r.x, act, slope ? slope[c] : 0.0f); r.y = apply_act(r.y, act, slope ? slope[c + 1] : 0.0f);
        r.z = apply_act(r.z, act, slope ? slope[c + 2] : 0.0f); r.w = apply_act(r.w, act, slope ? slope[c + 3] : 0.0f);
    }
    *reinterpret_cast<float4 *>(out.p + (((int64_t)n * out.h + oy) * out.w + ox) * out.ld + c) = r;
}

__global__ __launch_bounds__(256) void k_nearest(View in, View out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)out.n * out.h * out.w * (out.c >> 2);
    if (idx >= total) return;
    int c4n = out.c >> 2;
    int c = (int)(idx % c4n) * 4; int64_t pix = idx / c4n;
    int ox = (int)(pix % out.w); int64_t t = pix / out.w; int oy = (int)(t % out.h); int n = (int)(t / out.h);
    int fy = out.h / in.h, fx = out.w / in.w;
    int iy = oy / fy, ix = ox / fx;
    *reinterpret_cast<float4 *>(out.p + pix * out.ld + c) =
        *reinterpret_cast<const float4 *>(in.p + ((int64_t)(n * in.h + iy) * in.w + ix) * in.ld + c);
}

// out = act(a + b), or unary act / copy when b.p == nullptr
// float4 form of k_eltwise for modes 0 (act / copy) and 1 (add): channel counts and strides that are multiples of 4, 16-B aligned
// views (everything the layer programs produce).  Same arithmetic per element; 1 thread = 4 channels of one pixel.
__global__ __launch_bounds__(256) void k_eltwise4(View a, View b, View out, int act, int mode, const float *__restrict__ slope) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int c4n = out.c >> 2;
    const int64_t total = (int64_t)out.n * out.h * out.w * c4n;
    if (idx >= total) return;
    const int c = (int)(idx % c4n) * 4; const int64_t pix = idx / c4n;
    float4 v = *reinterpret_cast<const float4 *>(a.p + pix * a.ld + c);
    if (mode == 1) {
        const float4 w = *reinterpret_cast<const float4 *>(b.p + pix * b.ld + c);
        v.x = v.x + w.x; v.y = v.y + w.y; v.z = v.z + w.z; v.w = v.w + w.w;
    }
    float4 sl = float4{0.0f, 0.0f, 0.0f, 0.0f};
    if (slope) sl = *reinterpret_cast<const float4 *>(slope + c);
    v.x = apply_act(v.x, act, sl.x); v.y = apply_act(v.y, act, sl.y); v.z = apply_act(v.z, act, sl.z); v.w = apply_act(v.w, act, sl.w);
    *reinterpret_cast<float4 *>(out.p + pix * out.ld + c) = v;
}
static bool eltwise4_ok(const View &a, const View *b, const View &out, const float *slope) {
    uintptr_t bits = (uintptr_t)a.p | (uintptr_t)out.p | (uintptr_t)slope;
    int lds = a.ld | out.ld | out.c;
    if (b) { bits |= (uintptr_t)b->p; lds |= b->ld; }
    return !(bits & 15) && !(lds & 3);
}

__global__ __launch_bounds__(256) void k_eltwise(View a, View b, View out, int act, int mode, const float *__restrict__ slope) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)out.n * out.h * out.w * out.c;
    if (idx >= total) return;
    int c = (int)(idx % out.c); int64_t pix = idx / out.c;
    float v;
    if (mode == 3) {        // add with a CROPPED first operand: a is up to one row / column larger than out (torch's negative pad)
        const int64_t hw = (int64_t)out.h * out.w; const int64_t n = pix / hw, r = pix - n * hw;
        const int y = (int)(r / out.w), x = (int)(r - (int64_t)y * out.w);
        v = a.p[((n * a.h + y) * a.w + x) * a.ld + c] + b.p[pix * b.ld + c];
    } else {
        v = a.p[pix * a.ld + c];
        if (mode == 1) v = v + b.p[pix * b.ld + c];
        else if (mode == 2) { int64_t n = pix / ((int64_t)out.h * out.w); v = v * b.p[n * b.ld + c]; }
    }
    out.p[pix * out.ld + c] = apply_act(v, act, slope ? slope[c] : 0.0f);
}

// ZoeDepth attractor update (attractor.py:117-208, memory_efficient loop): out_k = b_k + agg_i dist(A_i - b_k)
__global__ __launch_bounds__(256) void k_attractor(View A, View b, View out, const float *__restrict__ par, int flags) {
    const float alpha = par[0];
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)out.n * out.h * out.w * out.c;
    if (idx >= total) return;
    int k = (int)(idx % out.c); int64_t pix = idx / out.c;
    const float c = b.p[pix * b.ld + k];
    const float *a = A.p + pix * A.ld;
    float delta = 0.0f;
    for (int i = 0; i < A.c; ++i) {
        const float dx = a[i] - c;
        float d;
        if (flags & 1) d = csm_expf(-alpha * (fabsf(dx) * fabsf(dx))) * dx;      // exp_attractor, gamma = 2
        else d = dx / (1.0f + alpha * (dx * dx));                                // inv_attractor, gamma = 2
        delta += d;
    }
    if (flags & 2) delta = delta / (float)A.c;
    out.p[pix * out.ld + k] = c + delta;
}

// ConditionalLogBinomial tail + expectation over the bins (dist_layers.py:46-121, zoedepth_v1.py:196-199); NB <= 256
__global__ __launch_bounds__(256) void k_logbinom(View pt, View cen, View out, const float *__restrict__ par) {
    int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)out.n * out.h * out.w;
    if (pix >= total) return;
    const float p_eps = par[0], min_temp = par[1], max_temp = par[2];
    const float *lb = par + 3;
    const float *q = pt.p + pix * pt.ld;
    const float p0 = q[0] + p_eps, p1 = q[1] + p_eps, t0 = q[2] + p_eps, t1 = q[3] + p_eps;
    const float p = p0 / (p0 + p1);
    float t = t0 / (t0 + t1);
    t = (max_temp - min_temp) * t + min_temp;
    const float eps = 1e-4f;                                           // LogBinomial.forward eps
    const float omx = fminf(fmaxf(1.0f - p, eps), 1.0f), x = fminf(fmaxf(p, eps), 1.0f);
    const float lx = csm_logf(x), lo = csm_logf(omx);
    const int K = cen.c;
    const float *c = cen.p + pix * cen.ld;
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const float y = (lb[k] + (float)k * lx + (float)(K - 1 - k) * lo) / t;
        mx = fmaxf(mx, y);
    }
    float den = 0.0f, num = 0.0f;
    for (int k = 0; k < K; ++k) {
        const float y = (lb[k] + (float)k * lx + (float)(K - 1 - k) * lo) / t;
        const float e = csm_expf(y - mx);
        den += e; num += e * c[k];
    }
    out.p[pix * out.ld] = num / den;
}

// global average pool with a fixed, oracle-reproducible reduction tree:
// 256 strided partial sums (sequential), then a binary tree 128,64,...,1, then / (h*w).
__global__ __launch_bounds__(256) void k_gavgpool(View in, View out) {
    __shared__ float part[256];
    int c = blockIdx.x, n = blockIdx.y;
    int hw = in.h * in.w;
    const float *P = in.p + (int64_t)n * hw * in.ld + c;
    float s = 0.0f;
    for (int i = threadIdx.x; i < hw; i += 256) s += P[(int64_t)i * in.ld];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out.p[(int64_t)n * out.ld + c] = part[0] / (float)hw;
}

// same reduction order as k_gavgpool (256 strided sequential partials per channel, then the tree 128,...,1), but a block owns
// 32 channels and lane t reads the 32 consecutive channels of pixels t, t+256, ...: 128-B pieces instead of one float per
// 1-KB stride (the per-channel kernel fetched 195 MB for a 26 MB tensor).
__global__ __launch_bounds__(256) void k_gavgpool32(View in, View out) {
    __shared__ float part[256][33];
    const int c0 = blockIdx.x * 32, n = blockIdx.y, t = threadIdx.x;
    const int hw = in.h * in.w;
    const float *P = in.p + (int64_t)n * hw * in.ld + c0;
    float s[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) s[c] = 0.0f;
    for (int i = t; i < hw; i += 256) {
        const float4 *q = reinterpret_cast<const float4 *>(P + (int64_t)i * in.ld);
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            float4 v = q[c4];
            s[4 * c4] += v.x; s[4 * c4 + 1] += v.y; s[4 * c4 + 2] += v.z; s[4 * c4 + 3] += v.w;
        }
    }
#pragma unroll
    for (int c = 0; c < 32; ++c) part[t][c] = s[c];
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        for (int i = t; i < st * 32; i += 256) {
            int r = i >> 5, c = i & 31;
            part[r][c] += part[r + st][c];
        }
        __syncthreads();
    }
    if (t < 32) out.p[(int64_t)n * out.ld + c0 + t] = part[0][t] / (float)hw;
}

__global__ __launch_bounds__(256) void k_nchw_to_nhwc(const float *__restrict__ src, int csrc, View out) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t total = (int64_t)out.n * out.h * out.w * out.c;
    if (idx >= total) return;
    int c = (int)(idx % out.c); int64_t pix = idx / out.c;
    int64_t hw = (int64_t)out.h * out.w; int64_t n = pix / hw, p = pix - n * hw;
    out.p[pix * out.ld + c] = c < csrc ? src[(n * csrc + c) * hw + p] : 0.0f;
}

__global__ __launch_bounds__(256) void k_nhwc_to_nchw(View in, float *__restrict__ dst) {
    int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t hw = (int64_t)in.h * in.w;
    int64_t total = (int64_t)in.n * in.c * hw;
    if (idx >= total) return;
    int64_t p = idx % hw; int64_t t = idx / hw; int c = (int)(t % in.c); int64_t n = t / in.c;
    dst[idx] = in.p[(n * hw + p) * in.ld + c];
}

// LDS-tiled forms for wide tensors (the 64 / 69-channel planes around the inpainting splat: 270-290 MB each).  The one-element-
// per-lane kernels above read (resp. write) 64 different cache lines per wave and ran at 1.25 TB/s (read + write); here a block
// moves 64 pixels x C channels through LDS, global accesses on both sides are contiguous runs (pixels of one channel plane /
// channels of consecutive pixels), the [c][65] pitch keeps both LDS phases conflict-free.
constexpr int kTrPix = 64;
__global__ __launch_bounds__(256) void k_nchw_to_nhwc_tile(const float *__restrict__ src, int csrc, View out) {
    extern __shared__ float tr[];                      // [out.c][kTrPix + 1]
    const int64_t hw = (int64_t)out.h * out.w;
    const int64_t tiles = (hw + kTrPix - 1) / kTrPix;
    const int64_t n = blockIdx.x / tiles, p0 = (blockIdx.x - n * tiles) * kTrPix;
    const int C = out.c;
    for (int i = threadIdx.x; i < C * kTrPix; i += 256) {
        const int c = i >> 6, p = i & 63;
        tr[c * (kTrPix + 1) + p] = (c < csrc && p0 + p < hw) ? src[(n * csrc + c) * hw + p0 + p] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * kTrPix; i += 256) {
        const int p = i / C, c = i - p * C;
        if (p0 + p < hw) out.p[(n * hw + p0 + p) * out.ld + c] = tr[c * (kTrPix + 1) + p];
    }
}
__global__ __launch_bounds__(256) void k_nhwc_to_nchw_tile(View in, float *__restrict__ dst) {
    extern __shared__ float tr[];                      // [in.c][kTrPix + 1]
    const int64_t hw = (int64_t)in.h * in.w;
    const int64_t tiles = (hw + kTrPix - 1) / kTrPix;
    const int64_t n = blockIdx.x / tiles, p0 = (blockIdx.x - n * tiles) * kTrPix;
    const int C = in.c;
    for (int i = threadIdx.x; i < C * kTrPix; i += 256) {
        const int p = i / C, c = i - p * C;
        tr[c * (kTrPix + 1) + p] = p0 + p < hw ? in.p[(n * hw + p0 + p) * in.ld + c] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * kTrPix; i += 256) {
        const int c = i >> 6, p = i & 63;
        if (p0 + p < hw) dst[(n * C + c) * hw + p0 + p] = tr[c * (kTrPix + 1) + p];
    }
}

// N tiles per group of the tile order (rem_to_tile): grouping pays when the layer's weights do not fit an XCD's 4 MB L2 next to the
// activation tiles in flight; the group's weight slices should take about half of it.  Speed only.
static int g_ngroup_enable = 1;     // csm_debug_conv_tuner_options bit 1 clears it (A/B measurements)
static int choose_ngroup(const ConvArgs &a, int BN) {
    if (!g_ngroup_enable || a.groups != 1) return 0;
    const int nn = (a.cout_g + BN - 1) / BN;
    int64_t kbytes = (int64_t)a.kh * a.kw * a.ncb * 128;                 // packed weight bytes of one output channel
    if (a.ksplit > 1 && !a.serial) kbytes /= a.ksplit;                   // (parallel split-K: a z slice reads its K run only)
    if (nn < 2 || kbytes * a.npad <= (3ll << 20)) return 0;
    int best = 0;
    for (int d = 1; d < nn; ++d)
        if (nn % d == 0 && (int64_t)d * BN * kbytes <= (2ll << 20)) best = d;
    if (!best && (int64_t)BN * kbytes <= (7ll << 19)) best = 1;
    return best;
}

static int launch_reduce(const ConvArgs &a, hipStream_t st) {
    k_splitk_reduce<<<(unsigned)(((int64_t)a.M * a.cout_g + 255) / 256), 256, 0, st>>>(a);
    return csm::check_launch("k_splitk_reduce");
}

template <int MT, int WM, int WN, int TN, bool FULLK, bool SER>
int launch_conv_k(const ConvArgs &a0, hipStream_t st) {
    constexpr int BM = MT * WM, BN = MT * WN * TN;
    ConvArgs a = a0;
    a.m_tiles = (a.M + BM - 1) / BM;
    size_t lds = (size_t)2 * (BM + BN) * kLdsLd * sizeof(float);
    static KernelPrep prep;
    (void)prep.ensure([&] { return prepare_kernel(&k_conv_mfma<MT, WM, WN, TN, FULLK, SER>, 64 * WM * WN, lds); });
    dim3 grid(a.m_tiles, (a.cout_g + BN - 1) / BN, a.groups * (SER ? 1 : a.ksplit));
    k_conv_mfma<MT, WM, WN, TN, FULLK, SER><<<grid, 64 * WM * WN, lds, st>>>(a);
    int rc = csm::check_launch("k_conv_mfma");
    if (rc || SER || a.ksplit <= 1) return rc;
    return launch_reduce(a, st);
}

template <int MT, int WM, int WN, int TN>
int launch_conv(const ConvArgs &a, hipStream_t st) {
    const bool full = (a.cin_g & 31) == 0;
    if (a.ksplit > 1 && a.serial) return full ? launch_conv_k<MT, WM, WN, TN, true, true>(a, st) : launch_conv_k<MT, WM, WN, TN, false, true>(a, st);
    return full ? launch_conv_k<MT, WM, WN, TN, true, false>(a, st) : launch_conv_k<MT, WM, WN, TN, false, false>(a, st);
}

// Grid quantisation: a launch of `total` equal tiles on S = 256 x (blocks per CU) slots takes ceil(total / S) rounds; with 3.1 rounds
// (the 40 x 40 x 1024 layers of ResNeXt at batch 8: 800 tiles of 128 x 128) a quarter of the machine time is an almost empty fourth
// round.  Every tile configuration produces the same bits, so a launch may MIX them: when `split` is set the big tiles cover whole
// rounds only and the remaining rows (less than ~0.6 of a round) are covered by a second launch of 64 x 64 tiles, which spreads
// them over all CUs.  Speed only; chosen per layer by the autotuner (csm_op.tile bit 7).
static int conv_split_rows(const ConvArgs &a, int BM, int BN, int blocks_per_cu) {
    const int64_t n_n = (int64_t)((a.cout_g + BN - 1) / BN) * a.groups * ((a.ksplit > 1 && !a.serial) ? a.ksplit : 1);
    const int64_t m_tiles = (a.M + BM - 1) / BM, slots = 256ll * (blocks_per_cu > 0 ? blocks_per_cu : 1);
    const double rounds = (double)(m_tiles * n_n) / (double)slots;
    const int64_t full = (int64_t)rounds;
    const double frac = rounds - (double)full;
    if (full < 1 || frac < 0.02 || frac > 0.6) return 0;
    const int64_t mt_main = full * slots / n_n;
    if (mt_main <= 0 || mt_main >= m_tiles) return 0;
    return (int)(mt_main * BM);
}

template <int WM, int WN, int TM, int TN, int NS, bool SER>
int launch_conv_dma_t(const ConvArgs &a0, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    ConvArgs a = a0;
    size_t lds = (size_t)NS * (BM + BN) * 128;
    static KernelPrep prep;
    const int blocks_per_cu = prep.ensure([&] { return prepare_kernel(&k_conv_dma<WM, WN, TM, TN, NS, SER>, 64 * WM * WN, lds); });
    if (a.split && (BM > 64 || BN > 64) && (a.ksplit <= 1 || SER)) {
        const int rows = conv_split_rows(a, BM, BN, blocks_per_cu);
        if (rows > 0) {
            ConvArgs tail = a;
            tail.m_begin = a.m_begin + rows; tail.split = 0;
            a.M = a.m_begin + rows; a.split = 0;
            int rc = launch_conv_dma_t<WM, WN, TM, TN, NS, SER>(a, st);
            if (rc) return rc;
            return launch_conv_dma_t<2, 2, 1, 1, 2, SER>(tail, st);
        }
    }
    a.m_tiles = (a.M - a.m_begin + BM - 1) / BM;
    a.ngroup = choose_ngroup(a, BN);
    dim3 grid(a.m_tiles, (a.cout_g + BN - 1) / BN, a.groups * (SER ? 1 : a.ksplit));
    k_conv_dma<WM, WN, TM, TN, NS, SER><<<grid, 64 * WM * WN, lds, st>>>(a);
    int rc = csm::check_launch("k_conv_dma");
    if (rc || SER || a.ksplit <= 1) return rc;
    return launch_reduce(a, st);
}

template <int WM, int WN, int TM, int TN, int NS = 2>
int launch_conv_dma(const ConvArgs &a, hipStream_t st) {
    if constexpr (NS == 2)        // (three- / four-stage tiles measured slower than two stages on every layer, also with interleaved issue: r04g)
        if (a.ksplit > 1 && a.serial) return launch_conv_dma_t<WM, WN, TM, TN, 2, true>(a, st);
    return launch_conv_dma_t<WM, WN, TM, TN, NS, false>(a, st);
}

// persistent launch: one round of resident blocks (a multiple of 8, at most one block per tile); layers that split K take the
// one-tile-per-block kernel of the same shape
template <int WM, int WN, int TM, int TN, bool SER>
int launch_conv_dma_p_t(const ConvArgs &a0, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    ConvArgs a = a0;
    const size_t lds = (size_t)2 * (BM + BN) * 128;
    static KernelPrep prep;
    const int blocks_per_cu = prep.ensure([&] { return prepare_kernel(&k_conv_dma_p<WM, WN, TM, TN, SER>, 64 * WM * WN, lds); });
    a.m_tiles = (a.M + BM - 1) / BM;
    a.ngroup = choose_ngroup(a, BN);
    const int n_n = (a.cout_g + BN - 1) / BN;
    const int64_t total = (int64_t)a.m_tiles * n_n * a.groups;
    if (total >= (1ll << 30)) return launch_conv_dma<WM, WN, TM, TN>(a0, st);
    int64_t grid = 256ll * blocks_per_cu;
    if (grid > ((total + 7) & ~7ll)) grid = (total + 7) & ~7ll;
    k_conv_dma_p<WM, WN, TM, TN, SER><<<(unsigned)grid, 64 * WM * WN, lds, st>>>(a, n_n, (int)total);
    return csm::check_launch("k_conv_dma_p");
}
template <int WM, int WN, int TM, int TN>
int launch_conv_dma_p(const ConvArgs &a, hipStream_t st) {
    if (a.m_begin != 0) return launch_conv_dma<WM, WN, TM, TN>(a, st);
    if (a.ksplit > 1) {
        if (a.serial && a.groups == 1) return launch_conv_dma_p_t<WM, WN, TM, TN, true>(a, st);
        return launch_conv_dma<WM, WN, TM, TN>(a, st);                          // parallel split-K: one tile per block + reduce
    }
    return launch_conv_dma_p_t<WM, WN, TM, TN, false>(a, st);
}

template <int WM, int WN, int TM, int TN, int TW, bool SER>
int launch_conv_patch_t(const ConvArgs &a0, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, TH = BM / TW;
    constexpr int NW = WM * WN, NPP = ((((TH + 2) * (TW + 2) + 7) / 8 + NW - 1) / NW) * NW;   // pieces, padded to the wave count
    ConvArgs a = a0;
    const int tiles_x = (a.out.w + TW - 1) / TW, tiles_y = (a.out.h + TH - 1) / TH;
    a.m_tiles = tiles_x * tiles_y * a.out.n;
    a.ngroup = choose_ngroup(a, BN);
    size_t lds = ((size_t)2 * NPP * 8 * 32 + (size_t)2 * BN * 32) * 4;
    static KernelPrep prep;
    (void)prep.ensure([&] { return prepare_kernel(&k_conv_patch<WM, WN, TM, TN, TW, SER>, 64 * WM * WN, lds); });
    dim3 grid(a.m_tiles, (a.cout_g + BN - 1) / BN, a.groups * (SER ? 1 : a.ksplit));
    k_conv_patch<WM, WN, TM, TN, TW, SER><<<grid, 64 * WM * WN, lds, st>>>(a, tiles_x, tiles_y);
    int rc = csm::check_launch("k_conv_patch");
    if (rc || SER || a.ksplit <= 1) return rc;
    return launch_reduce(a, st);
}

template <int WM, int WN, int TM, int TN, int TW>
int launch_conv_patch(const ConvArgs &a, hipStream_t st) {
    if (a.ksplit > 1 && a.serial) return launch_conv_patch_t<WM, WN, TM, TN, TW, true>(a, st);
    return launch_conv_patch_t<WM, WN, TM, TN, TW, false>(a, st);
}

template <int WM, int WN, int TM, int TN, int TW, bool SER, int MINW = 2>
int launch_conv_patch_p_t(const ConvArgs &a0, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, TH = BM / TW;
    constexpr int NW = WM * WN, NPP = ((((TH + 2) * (TW + 2) + 7) / 8 + NW - 1) / NW) * NW;
    ConvArgs a = a0;
    const int tiles_x = (a.out.w + TW - 1) / TW, tiles_y = (a.out.h + TH - 1) / TH;
    a.m_tiles = tiles_x * tiles_y * a.out.n;
    a.ngroup = choose_ngroup(a, BN);
    const size_t lds = ((size_t)2 * NPP * 8 * 32 + (size_t)2 * BN * 32) * 4;
    static KernelPrep prep;
    const int blocks_per_cu = prep.ensure([&] { return prepare_kernel(&k_conv_patch_p<WM, WN, TM, TN, TW, SER, MINW>, 64 * WM * WN, lds); });
    const int n_n = (a.cout_g + BN - 1) / BN;
    const int64_t total = (int64_t)a.m_tiles * n_n * a.groups;
    if (total >= (1ll << 30)) return launch_conv_patch<WM, WN, TM, TN, TW>(a0, st);
    int64_t grid = 256ll * blocks_per_cu;
    if (grid > ((total + 7) & ~7ll)) grid = (total + 7) & ~7ll;
    k_conv_patch_p<WM, WN, TM, TN, TW, SER, MINW><<<(unsigned)grid, 64 * WM * WN, lds, st>>>(a, tiles_x, tiles_y, n_n, (int)total);
    return csm::check_launch("k_conv_patch_p");
}
template <int WM, int WN, int TM, int TN, int TW, int MINW = 2>
int launch_conv_patch_p(const ConvArgs &a, hipStream_t st) {
    if (a.ksplit > 1) {
        if (a.serial && a.groups == 1) return launch_conv_patch_p_t<WM, WN, TM, TN, TW, true, MINW>(a, st);
        return launch_conv_patch<WM, WN, TM, TN, TW>(a, st);                   // parallel split-K: one tile per block + reduce
    }
    return launch_conv_patch_p_t<WM, WN, TM, TN, TW, false, MINW>(a, st);
}

// weights-stationary launch: one round of blocks, 8 XCDs x (column tiles x `per` blocks), every block keeps ITS column tile's weight panel
constexpr int kWsPatchBytes = 2 * (((16 + 2) * (16 + 2) + 7) / 8 + 1) * 1024;      // two patch stages of k_conv_ws
static bool ws_fits(const ConvArgs &a, int BN) {
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dil == 1 && a.ksplit <= 1 && a.m_begin == 0 && (a.cin_g & 31) == 0 &&
           (size_t)kWsPatchBytes + (size_t)9 * a.ncb * BN * 128 <= (size_t)160 * 1024;
}
template <int TN>
int launch_conv_ws(const ConvArgs &a0, hipStream_t st) {
    constexpr int BN = 32 * TN;
    ConvArgs a = a0;
    const int tiles_x = (a.out.w + 15) / 16, tiles_y = (a.out.h + 15) / 16;
    a.m_tiles = tiles_x * tiles_y * a.out.n;
    const int n_n = (a.cout_g + BN - 1) / BN, n_ct = a.groups * n_n;
    const size_t lds = (size_t)kWsPatchBytes + (size_t)9 * a.ncb * BN * 128;
    static KernelPrep prep;
    (void)prep.ensure([&] { return prepare_kernel(&k_conv_ws<TN>, 512, (size_t)160 * 1024); });
    // 32 CUs per XCD, one block per CU: `per` blocks share a column tile's M range on an XCD (at least one; never more than it has tiles)
    int per = 32 / n_ct;
    if (per < 1) per = 1;
    const int len_max = (a.m_tiles + 7) / 8;
    if (per > len_max) per = len_max;
    k_conv_ws<TN><<<(unsigned)(8 * n_ct * per), 512, lds, st>>>(a, tiles_x, tiles_y, n_n, n_ct);
    return csm::check_launch("k_conv_ws");
}

static bool narrow_eligible(const ConvArgs &a) {
    return a.cout_g <= 4 && a.groups == 1 && a.ksplit == 1 && !(a.cin_g & 3) && !(a.in.ld & 3) && narrow_lds(a, 4, nullptr, nullptr) != 0;
}

static bool dma_eligible(const ConvArgs &a);
static bool patch_eligible(const ConvArgs &a) {
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dil == 1 && dma_eligible(a);
}

static bool dma_eligible(const ConvArgs &a) {
    int64_t bytes_in = (((int64_t)a.in.n * a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4;
    int64_t bytes_w = (int64_t)a.groups * a.kh * a.kw * a.ncb * a.npad * 128;
    return (a.cin_g & 31) == 0 && a.kh * a.kw <= 32 && bytes_in < (1ll << 31) && bytes_w < (1ll << 31) && !(a.in.ld & 3) &&
           !(((uintptr_t)a.in.p | (uintptr_t)a.w) & 15);
}

// ---- tile selection ---------------------------------------------------------------------------------------
// Measured on the three nets (tools/conv_bench.py --sweep, profiles/): occupancy beats register-tile reuse in this
// two-stage pipeline, so the default is the 64x64 tile (4 blocks = 16 waves per CU); narrow outputs get narrow tiles.
enum { CFG_128x128_4w = 0, CFG_128x64 = 1, CFG_64x64 = 2, CFG_128x128_8w = 3, CFG_128x32 = 4, CFG_64x16 = 5,
       // LDS-DMA kernel (k_conv_dma)
       CFG_D64x64 = 6, CFG_D128x64 = 7, CFG_D128x128 = 8, CFG_D128x128_8w = 9, CFG_D256x128_8w = 10, CFG_D64x128 = 11, CFG_D128x32 = 12,
       CFG_NARROW = 13,   // k_conv_narrow (cout <= 4)
       // odd tile heights (1x4 waves, wave tile 32*TM x 32): more block counts for the tuner to dodge grid quantisation with
       CFG_D96x128 = 14, CFG_D160x128 = 15, CFG_D224x128 = 16, CFG_D192x128 = 17,
       // 3x3 patch re-use kernel (k_conv_patch); _w8 = 8-pixel-wide output tiles for small maps
       CFG_P64x64 = 18, CFG_P128x64 = 19, CFG_P64x128 = 20, CFG_P128x128 = 21, CFG_P256x128 = 22, CFG_P128x32 = 23,
       CFG_P64x64_w8 = 24, CFG_P128x128_w8 = 25, CFG_P128x32_w8 = 26, CFG_P128x128_8w = 27,
       // three LDS stages (loads two chunks ahead) and 256 x 64 tiles (N = 64 layers: the B tile is shared by four 64 x 64 wave tiles)
       CFG_D64x64_s3 = 28, CFG_D128x64_s3 = 29, CFG_D64x128_s3 = 30, CFG_D128x128_s3 = 31, CFG_D128x128_8w_s3 = 32,
       CFG_D256x128_8w_s3 = 33, CFG_D256x64 = 34, CFG_D256x64_s3 = 35, CFG_P256x64 = 36, CFG_D64x64_s4 = 37,
       // persistent blocks, loader one chunk ahead across tile boundaries (k_conv_dma_p); ksplit == 1 layers
       CFG_Q64x64 = 38, CFG_Q128x64 = 39, CFG_Q64x128 = 40, CFG_Q128x128_8w = 41, CFG_Q128x32 = 42,
       // persistent patch kernel (k_conv_patch_p): the next tile's patch is fetched during the current tile's taps
       CFG_R128x32 = 43, CFG_R64x64 = 44, CFG_R128x64 = 45, CFG_R128x32_w8 = 46, CFG_R128x128_8w = 47, CFG_R64x128 = 48, CFG_R64x64_w8 = 49,
       // the 8-wave persistent patch tile capped at 128 VGPRs: two blocks per CU
       CFG_R128x128_8w_o4 = 50,
       // weights-stationary 3x3 (k_conv_ws): 16 x 16 pixel tiles x 32 / 64 output channels, the column tile's whole weight panel in LDS
       CFG_W256x32 = 51, CFG_W256x64 = 52,
       CFG_COUNT = 53 };
static int g_force_cfg = -1;
static int g_force_serial = -1;    // tests: -1 = rule / tuned, 0 = parallel split-K, 1 = serial split-K
static int g_tune_split = 1;       // tuner: consider mixed-tile launches (csm_debug_conv_tuner_options)
static int g_dbg = 0;
constexpr int kTileSerial = 64;
constexpr int kTileSplit = 128;    // csm_op.tile bit 7: mixed-tile launch (big tiles for whole rounds + 64 x 64 tiles for the rest)

static void read_force_env() {
    static bool env_read = false;
    if (env_read) return;      // tuning aid: CSM_FORCE_CONV_CFG=<n> forces one tile configuration for every eligible conv
    env_read = true;
    const char *e = getenv("CSM_FORCE_CONV_CFG");
    if (e && *e) g_force_cfg = atoi(e);
}

// Default rule when an op carries no tuned tile (csm_op.tile == 0): the LDS-DMA kernel with the 64x64 tile wins or ties on
// every layer of the three nets at batch 1 (profiles/r01_conv_sweep.txt); narrow outputs get narrow tiles.
static int choose_cfg(const ConvArgs &a, int N) {
    read_force_env();
    if (g_force_cfg >= 0 && g_force_cfg < CFG_COUNT) return g_force_cfg;
    if (N <= 16) return CFG_64x16;      // (k_conv_narrow is an autotune candidate only: it wins on some cout = 1 layers, loses on others)
    if (N <= 32) return dma_eligible(a) ? CFG_D128x32 : CFG_128x32;
    return CFG_D64x64;
}

static int launch_conv_cfg(int cfg, const ConvArgs &a, hipStream_t st) {
    switch (cfg) {
#ifndef CSM_PROBE          // (development: -DCSM_PROBE compiles the persistent kernels only, for a quick look at their ISA)
        case CFG_128x128_4w: return launch_conv<32, 4, 1, 4>(a, st);
        case CFG_128x64: return launch_conv<32, 4, 1, 2>(a, st);
        case CFG_128x128_8w: return launch_conv<32, 4, 2, 2>(a, st);
        case CFG_128x32: return launch_conv<32, 4, 1, 1>(a, st);
        case CFG_64x16: return launch_conv<16, 4, 1, 1>(a, st);
        case CFG_D64x64: return launch_conv_dma<2, 2, 1, 1>(a, st);
        case CFG_D128x64: return launch_conv_dma<2, 2, 2, 1>(a, st);
        case CFG_D64x128: return launch_conv_dma<2, 2, 1, 2>(a, st);
        case CFG_D128x128: return launch_conv_dma<2, 2, 2, 2>(a, st);
        case CFG_D128x128_8w: return launch_conv_dma<2, 4, 2, 1>(a, st);
        case CFG_D256x128_8w: return launch_conv_dma<4, 2, 2, 2>(a, st);
        case CFG_D128x32: return launch_conv_dma<4, 1, 1, 1>(a, st);
        case CFG_NARROW: return launch_narrow(a, st);
        case CFG_D96x128: return launch_conv_dma<1, 4, 3, 1>(a, st);
        case CFG_D160x128: return launch_conv_dma<1, 4, 5, 1>(a, st);
        case CFG_D224x128: return launch_conv_dma<1, 4, 7, 1>(a, st);
        case CFG_D192x128: return launch_conv_dma<1, 4, 6, 1>(a, st);
        case CFG_P64x64: return launch_conv_patch<2, 2, 1, 1, 16>(a, st);
        case CFG_P128x64: return launch_conv_patch<2, 2, 2, 1, 16>(a, st);
        case CFG_P64x128: return launch_conv_patch<2, 2, 1, 2, 16>(a, st);
        case CFG_P128x128: return launch_conv_patch<2, 2, 2, 2, 16>(a, st);
        case CFG_P256x128: return launch_conv_patch<4, 2, 2, 2, 16>(a, st);
        case CFG_P128x32: return launch_conv_patch<4, 1, 1, 1, 16>(a, st);
        case CFG_P64x64_w8: return launch_conv_patch<2, 2, 1, 1, 8>(a, st);
        case CFG_P128x128_w8: return launch_conv_patch<2, 2, 2, 2, 8>(a, st);
        case CFG_P128x32_w8: return launch_conv_patch<4, 1, 1, 1, 8>(a, st);
        case CFG_P128x128_8w: return launch_conv_patch<2, 4, 2, 1, 16>(a, st);
        case CFG_D64x64_s3: return launch_conv_dma<2, 2, 1, 1, 3>(a, st);
        case CFG_D128x64_s3: return launch_conv_dma<2, 2, 2, 1, 3>(a, st);
        case CFG_D64x128_s3: return launch_conv_dma<2, 2, 1, 2, 3>(a, st);
        case CFG_D128x128_s3: return launch_conv_dma<2, 2, 2, 2, 3>(a, st);
        case CFG_D128x128_8w_s3: return launch_conv_dma<2, 4, 2, 1, 3>(a, st);
        case CFG_D256x128_8w_s3: return launch_conv_dma<4, 2, 2, 2, 3>(a, st);
        case CFG_D256x64: return launch_conv_dma<4, 1, 2, 2>(a, st);
        case CFG_D256x64_s3: return launch_conv_dma<4, 1, 2, 2, 3>(a, st);
        case CFG_P256x64: return launch_conv_patch<4, 1, 2, 2, 16>(a, st);
        case CFG_D64x64_s4: return launch_conv_dma<2, 2, 1, 1, 4>(a, st);
#endif
        case CFG_Q64x64: return launch_conv_dma_p<2, 2, 1, 1>(a, st);
        case CFG_Q128x64: return launch_conv_dma_p<2, 2, 2, 1>(a, st);
        case CFG_Q64x128: return launch_conv_dma_p<2, 2, 1, 2>(a, st);
        case CFG_Q128x128_8w: return launch_conv_dma_p<2, 4, 2, 1>(a, st);
        case CFG_Q128x32: return launch_conv_dma_p<4, 1, 1, 1>(a, st);
        case CFG_R128x32: return launch_conv_patch_p<4, 1, 1, 1, 16>(a, st);
        case CFG_R64x64: return launch_conv_patch_p<2, 2, 1, 1, 16>(a, st);
        case CFG_R128x64: return launch_conv_patch_p<2, 2, 2, 1, 16>(a, st);
        case CFG_R128x32_w8: return launch_conv_patch_p<4, 1, 1, 1, 16>(a, st);     // (the 8-wide persistent tiles were dropped, see CFG_R64x64_w8)
        case CFG_R128x128_8w: return launch_conv_patch_p<2, 4, 2, 1, 16>(a, st);
        case CFG_R64x128: return launch_conv_patch_p<2, 2, 1, 2, 16>(a, st);
        case CFG_R128x128_8w_o4: return launch_conv_patch_p<2, 4, 2, 1, 16, 4>(a, st);
        case CFG_W256x32: return launch_conv_ws<1>(a, st);
        case CFG_W256x64: return launch_conv_ws<2>(a, st);
        // (the 8-wide persistent tiles: where the LDS-read / barrier hazard of the unrolled taps showed; found by tools/check_persistent.py,
        // fixed in k_conv_patch_p, the variants themselves stay out)
        case CFG_R64x64_w8: return launch_conv_patch_p<2, 2, 1, 1, 16>(a, st);
        default: return launch_conv<32, 2, 2, 1>(a, st);
    }
}

inline unsigned blocks_for(int64_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace

static int make_view(const csm_tensor_desc *tensors, int n_tensors, int id, float *workspace, void *const *ext,
                     int n_ext, View &v) {
    if (id < 0 || id >= n_tensors) { csm::set_error("tensor id %d out of range", id); return CSM_ERR_ARG; }
    const csm_tensor_desc &t = tensors[id];
    float *base;
    if (t.ext >= 0) {
        if (t.ext >= n_ext || !ext[t.ext]) { csm::set_error("ext slot %d missing", t.ext); return CSM_ERR_ARG; }
        base = (float *)ext[t.ext];
    } else base = workspace;
    v.p = base + t.offset; v.n = t.n; v.h = t.h; v.w = t.w; v.c = t.c; v.ld = t.ld;
    return CSM_OK;
}

static int run_ops(const csm_op *ops, int n_ops, const csm_tensor_desc *tensors, int n_tensors, const float *weights,
                   float *workspace, void *const *ext, int n_ext, hipStream_t st, hipEvent_t *ev) {
    if (ev) CSM_HIP(hipEventRecord(ev[0], st));
    for (int i = 0; i < n_ops; ++i) {
        const csm_op &op = ops[i];
        View in{}, in1{}, out{};
        int rc = make_view(tensors, n_tensors, op.in0, workspace, ext, n_ext, in); if (rc) return rc;
        rc = make_view(tensors, n_tensors, op.out, workspace, ext, n_ext, out); if (rc) return rc;
        if (op.in1 >= 0) { rc = make_view(tensors, n_tensors, op.in1, workspace, ext, n_ext, in1); if (rc) return rc; }
        switch (op.kind) {
            case CSM_OP_CONV: {
                ConvArgs a{};
                a.in = in; a.out = out; a.res = in1;
                a.w = weights + op.w_off; a.bias = op.b_off >= 0 ? weights + op.b_off : nullptr;
                a.slope = op.aux_off >= 0 ? weights + op.aux_off : nullptr;
                a.kh = op.kh; a.kw = op.kw; a.stride = op.stride; a.pad = op.pad; a.dil = op.dil;
                a.groups = op.groups; a.cin_g = op.cin_g; a.cout_g = op.cout_g; a.npad = (op.cout_g + 31) / 32 * 32;
                a.act = op.act; a.res_mode = op.in1 >= 0 ? op.res_mode : 0;
                a.M = out.n * out.h * out.w; a.ncb = (op.cin_g + 31) / 32;
                set_fast_div((unsigned)(out.h * out.w), a.dv_hw_mul, a.dv_hw_shr); set_fast_div((unsigned)out.w, a.dv_w_mul, a.dv_w_shr);
                a.ksplit = op.ksplit > 1 ? op.ksplit : 1; a.partial = nullptr;
                if (a.ksplit > 1) {
                    View sc{};
                    if (op.groups != 1) { csm::set_error("op %d: ksplit needs groups == 1", i); return CSM_ERR_ARG; }
                    rc = make_view(tensors, n_tensors, op.scratch, workspace, ext, n_ext, sc); if (rc) return rc;
                    a.partial = sc.p;
                }
                if ((in.ld & 3) || (op.cin_g & 3) || (((uintptr_t)in.p) & 15)) {
                    csm::set_error("op %d: conv input must be 16-byte aligned with channels %% 4 == 0", i); return CSM_ERR_ARG;
                }
                a.dbg = g_dbg;
                if (op.flags & CSM_CONV_FLAG_WINOGRAD) {      // Winograd F(2x2, 3x3): its own arithmetic (part of the lowering's contract), its own kernel
                    if (!wino_eligible(a)) { csm::set_error("op %d: Winograd flag on an ineligible convolution (3x3 / stride 1 / pad 1 / dense / cin %% 32 / cout %% 64 / ksplit 1)", i); return CSM_ERR_ARG; }
                    rc = launch_conv_wino(a, st);
                    if (rc) return rc;
                    break;
                }
                if (op.flags & CSM_CONV_FLAG_WINOGRAD4) {     // Winograd F(4x4, 3x3): the same layer class, 36 instead of 64 products per 4x4 outputs (wino4.hip)
                    if (op.scratch >= 0) {                    // optional scratch of the row-split execution forms (small launches): speed only
                        View sc{};
                        rc = make_view(tensors, n_tensors, op.scratch, workspace, ext, n_ext, sc); if (rc) return rc;
                        if ((int64_t)sc.n * sc.h * sc.w * sc.c < wino4_scratch_floats(out.n, out.h, out.w, op.cout_g) || sc.ld != sc.c || (((uintptr_t)sc.p) & 15)) {
                            csm::set_error("op %d: Winograd F(4x4) scratch tensor too small or not contiguous", i); return CSM_ERR_ARG;
                        }
                        a.partial = sc.p;
                    }
                    if (!wino4_eligible(a)) { csm::set_error("op %d: Winograd F(4x4) flag on an ineligible convolution (3x3 / stride 1 / pad 1 / dense / cin %% 32 / cout %% 64 / ksplit 1)", i); return CSM_ERR_ARG; }
                    rc = launch_conv_wino4(a, st);
                    if (rc) return rc;
                    break;
                }
                if (op.flags & CSM_CONV_FLAG_GROUPED) {       // narrow groups on the vector pipe: the direct chain, its own weight image (grouped.hip)
                    if (!grouped_eligible(a)) { csm::set_error("op %d: grouped-conv flag on an ineligible convolution (3x3 / stride 1 / pad 1 / cin_g == cout_g in {8, 16, 32} / channels %% 32 / ksplit 1)", i); return CSM_ERR_ARG; }
                    rc = launch_conv_grouped(a, st);
                    if (rc) return rc;
                    break;
                }
                if (op.flags & 2) {      // stem: (tap, channel)-packed K (weights packed by the host for exactly this kernel)
                    if (op.groups != 1 || op.cin_g != 4 || a.ksplit != 1) { csm::set_error("op %d: stem flag needs groups 1, cin 4, ksplit 1", i); return CSM_ERR_ARG; }
                    dim3 grid((a.M + 63) / 64, (op.cout_g + 63) / 64, 1);
                    k_conv_stem<<<grid, 256, 0, st>>>(a);
                    rc = csm::check_launch("k_conv_stem");
                    if (rc) return rc;
                    break;
                }
                // csm_op.tile = 1 + configuration (+ kTileSerial: split-K runs walked by one block).  Untuned ops: serial once the
                // batch supplies enough output tiles by itself (speed only: both executions give the same bits)
                const int tcfg = op.tile & (kTileSerial - 1);
                a.split = (op.tile & kTileSplit) != 0 && g_force_cfg < 0;
                const bool tuned = tcfg > 0 && tcfg <= CFG_COUNT && g_force_cfg < 0;
                a.serial = a.ksplit > 1 && (g_force_serial >= 0 ? g_force_serial != 0 : tuned ? (op.tile & kTileSerial) != 0
                                            : (int64_t)((a.M + 63) / 64) * ((op.cout_g + 63) / 64) >= 512);
                int cfg = tuned ? tcfg - 1 : choose_cfg(a, op.cout_g);
                if ((cfg == CFG_W256x32 || cfg == CFG_W256x64) && !(patch_eligible(a) && ws_fits(a, cfg == CFG_W256x32 ? 32 : 64))) cfg = CFG_R64x64;
                if (((cfg >= CFG_P64x64 && cfg <= CFG_P128x128_8w) || cfg == CFG_P256x64 || cfg >= CFG_R128x32) && !patch_eligible(a)) cfg = CFG_D64x64;
                if (cfg == CFG_NARROW && !narrow_eligible(a)) cfg = CFG_64x16;
                if (cfg >= CFG_D64x64 && cfg != CFG_NARROW && !dma_eligible(a)) cfg = op.cout_g <= 16 ? CFG_64x16 : (op.cout_g <= 32 ? CFG_128x32 : CFG_64x64);
                rc = launch_conv_cfg(cfg, a, st);
                if (rc) return rc;
                break;
            }
            case CSM_OP_DWCONV: {
                if ((in.ld & 3) || (out.ld & 3) || (out.c & 3)) { csm::set_error("op %d: dwconv needs c%%4==0", i); return CSM_ERR_ARG; }
                DwArgs a{in, out, weights + op.w_off, op.b_off >= 0 ? weights + op.b_off : nullptr,
                         op.aux_off >= 0 ? weights + op.aux_off : nullptr, op.kh, op.kw, op.stride, op.pad, op.dil, op.act};
                size_t lds = ((size_t)op.kh * op.kw * 32 + (size_t)(8 + op.kh - 1) * (16 + op.kw - 1) * 32) * 4;
                if (op.stride == 1 && op.dil == 1 && !(out.c & 31) && lds <= 64 * 1024 && out.h == in.h + 2 * op.pad - op.kh + 1) {
                    int tiles_x = (out.w + 15) / 16, tiles_y = (out.h + 7) / 8;
                    k_dwconv_lds<<<dim3((unsigned)(tiles_x * tiles_y * out.n), (unsigned)(out.c / 32)), 256, lds, st>>>(a, tiles_x, tiles_y);
                } else
                    k_dwconv<<<blocks_for((int64_t)out.n * out.h * out.w * (out.c >> 2)), 256, 0, st>>>(a);
                break;
            }
            case CSM_OP_MAXPOOL:
                if ((in.ld & 3) || (out.ld & 3) || (out.c & 3)) { csm::set_error("op %d: maxpool needs c%%4==0", i); return CSM_ERR_ARG; }
                k_maxpool<<<blocks_for((int64_t)out.n * out.h * out.w * (out.c >> 2)), 256, 0, st>>>(in, out, op.kh, op.stride, op.pad);
                break;
            case CSM_OP_BILINEAR: {
                bool align = op.flags & 1;
                float sh, sw;
                if (align) { sh = out.h > 1 ? (float)(in.h - 1) / (float)(out.h - 1) : 0.0f; sw = out.w > 1 ? (float)(in.w - 1) / (float)(out.w - 1) : 0.0f; }
                else { sh = (float)in.h / (float)out.h; sw = (float)in.w / (float)out.w; }
                bool vec = !(out.c & 3) && !(in.ld & 3) && !(out.ld & 3) && !(((uintptr_t)in.p | (uintptr_t)out.p) & 15);
                const float *bsl = op.aux_off >= 0 ? weights + op.aux_off : nullptr;
                if (vec && out.h <= 65535 && out.n <= 65535 && (int64_t)out.w * (out.c >> 2) < (1ll << 30)) {
                    unsigned mul, shr;
                    set_fast_div((unsigned)(out.c >> 2), mul, shr);
                    k_bilinear_rows<<<dim3(blocks_for((int64_t)out.w * (out.c >> 2)), (unsigned)out.h, (unsigned)out.n), 256, 0, st>>>(in, out, align ? 1 : 0, sh, sw, op.act, bsl, mul, shr);
                } else if (vec) k_bilinear<4><<<blocks_for((int64_t)out.n * out.h * out.w * (out.c >> 2)), 256, 0, st>>>(in, out, align ? 1 : 0, sh, sw, op.act, bsl);
                else k_bilinear<1><<<blocks_for((int64_t)out.n * out.h * out.w * out.c), 256, 0, st>>>(in, out, align ? 1 : 0, sh, sw, op.act, bsl);
                break;
            }
            case CSM_OP_NEAREST:
                if ((in.ld & 3) || (out.ld & 3) || (out.c & 3)) { csm::set_error("op %d: nearest needs c%%4==0", i); return CSM_ERR_ARG; }
                k_nearest<<<blocks_for((int64_t)out.n * out.h * out.w * (out.c >> 2)), 256, 0, st>>>(in, out);
                break;
            case CSM_OP_ADD:
                if (in.h < out.h || in.w < out.w || in.h > out.h + 1 || in.w > out.w + 1 || in1.h != out.h || in1.w != out.w) {
                    csm::set_error("op %d: add: the first operand may exceed the output by at most one row / column", i); return CSM_ERR_ARG;
                }
                if (in.h == out.h && in.w == out.w && eltwise4_ok(in, &in1, out, nullptr))
                    k_eltwise4<<<blocks_for((int64_t)out.n * out.h * out.w * (out.c >> 2)), 256, 0, st>>>(in, in1, out, op.act, 1, nullptr);
                else
                    k_eltwise<<<blocks_for((int64_t)out.n * out.h * out.w * out.c), 256, 0, st>>>(in, in1, out, op.act,
                                                                                                (in.h != out.h || in.w != out.w) ? 3 : 1, nullptr);
                break;
            case CSM_OP_SCALE:
                k_eltwise<<<blocks_for((int64_t)out.n * out.h * out.w * out.c), 256, 0, st>>>(in, in1, out, op.act, 2, nullptr);
                break;
            case CSM_OP_ACT:
            case CSM_OP_COPY:
                if (eltwise4_ok(in, nullptr, out, op.aux_off >= 0 ? weights + op.aux_off : nullptr))
                    k_eltwise4<<<blocks_for((int64_t)out.n * out.h * out.w * (out.c >> 2)), 256, 0, st>>>(in, in1, out, op.kind == CSM_OP_ACT ? op.act : 0, 0,
                                                                                                        op.aux_off >= 0 ? weights + op.aux_off : nullptr);
                else
                    k_eltwise<<<blocks_for((int64_t)out.n * out.h * out.w * out.c), 256, 0, st>>>(in, in1, out, op.kind == CSM_OP_ACT ? op.act : 0, 0,
                                                                                                  op.aux_off >= 0 ? weights + op.aux_off : nullptr);
                break;
            case CSM_OP_GAVGPOOL:
                if (!(in.c & 31) && !(in.ld & 3) && !(((uintptr_t)in.p) & 15)) k_gavgpool32<<<dim3(in.c / 32, in.n), 256, 0, st>>>(in, out);
                else k_gavgpool<<<dim3(in.c, in.n), 256, 0, st>>>(in, out);
                break;
            case CSM_OP_ATTRACTOR:
                if (op.aux_off < 0 || in.n != in1.n || in.h != in1.h || in.w != in1.w) { csm::set_error("op %d: attractor operands", i); return CSM_ERR_ARG; }
                k_attractor<<<blocks_for((int64_t)out.n * out.h * out.w * out.c), 256, 0, st>>>(in, in1, out, weights + op.aux_off, op.flags);
                break;
            case CSM_OP_LOGBINOM:
                if (op.aux_off < 0 || in.c < 4 || in1.c > 256) { csm::set_error("op %d: logbinom operands", i); return CSM_ERR_ARG; }
                k_logbinom<<<blocks_for((int64_t)out.n * out.h * out.w), 256, 0, st>>>(in, in1, out, weights + op.aux_off);
                break;
            case CSM_OP_NCHW_TO_NHWC:
                if (out.c >= 8 && out.c <= 240)
                    k_nchw_to_nhwc_tile<<<(unsigned)(out.n * (((int64_t)out.h * out.w + kTrPix - 1) / kTrPix)), 256,
                                          sizeof(float) * (size_t)out.c * (kTrPix + 1), st>>>(in.p, in.c, out);
                else
                    k_nchw_to_nhwc<<<blocks_for((int64_t)out.n * out.h * out.w * out.c), 256, 0, st>>>(in.p, in.c, out);
                break;
            case CSM_OP_NHWC_TO_NCHW:
                if (in.c >= 8 && in.c <= 240)
                    k_nhwc_to_nchw_tile<<<(unsigned)(in.n * (((int64_t)in.h * in.w + kTrPix - 1) / kTrPix)), 256,
                                          sizeof(float) * (size_t)in.c * (kTrPix + 1), st>>>(in, out.p);
                else
                    k_nhwc_to_nchw<<<blocks_for((int64_t)in.n * in.h * in.w * in.c), 256, 0, st>>>(in, out.p);
                break;
            case CSM_OP_LAYERNORM: {
                if (op.w_off < 0 || op.b_off < 0 || op.aux_off < 0 || in.c != out.c) { csm::set_error("op %d: layernorm operands", i); return CSM_ERR_ARG; }
                rc = csm::launch_layernorm(in.p, in.ld, out.p, out.ld, (int64_t)in.n * in.h * in.w, in.c, weights + op.w_off, weights + op.b_off,
                                           weights + op.aux_off /* {eps}, read on the device */, st);
                if (rc) return rc;
                break;
            }
            case CSM_OP_ATTENTION: {
                const int heads = op.groups, d = op.cin_g;
                if (heads < 1 || in.c != 3 * heads * d || out.c != heads * d || in.w != 1 || out.h != in.h) { csm::set_error("op %d: attention operands", i); return CSM_ERR_ARG; }
                rc = csm::launch_attention(in.p, in.ld, out.p, out.ld, in.n, in.h, heads, d, op.aux_off >= 0 ? weights + op.aux_off : nullptr, op.kh, op.kw, st);
                if (rc) return rc;
                break;
            }
            case CSM_OP_TOKENS: {
                const int mode = op.flags;
                const int np = mode == 0 ? in.h * in.w : out.h * out.w;
                const bool ok = mode == 0 ? (out.h == np + 1 && out.w == 1 && out.c == in.c && op.aux_off >= 0)
                                          : (in.h == np + 1 && in.w == 1 && out.c == (mode == 1 ? 2 : 1) * in.c);
                if (!ok || in.n != out.n) { csm::set_error("op %d: tokens operands (mode %d)", i, mode); return CSM_ERR_ARG; }
                rc = csm::launch_tokens(mode, in.p, in.ld, out.p, out.ld, in.n, np, in.c, op.aux_off >= 0 ? weights + op.aux_off : nullptr, st);
                if (rc) return rc;
                break;
            }
            case CSM_OP_DEPTH_TO_SPACE: {
                const int k = op.stride;
                if (k < 1 || out.h != in.h * k || out.w != in.w * k || in.c != k * k * out.c) { csm::set_error("op %d: depth_to_space operands", i); return CSM_ERR_ARG; }
                rc = csm::launch_depth_to_space(in.p, in.ld, out.p, out.ld, in.n, in.h, in.w, k, out.c, st);
                if (rc) return rc;
                break;
            }
            default:
                csm::set_error("op %d: unknown kind %d", i, op.kind);
                return CSM_ERR_ARG;
        }
        rc = csm::check_launch("program op");
        if (rc) { csm::set_error("op %d (kind %d) launch failed", i, op.kind); return rc; }
        if (ev) CSM_HIP(hipEventRecord(ev[i + 1], st));
    }
    return CSM_OK;
}

extern "C" int csm_run_program(const csm_op *ops, int n_ops, const csm_tensor_desc *tensors, int n_tensors,
                               const float *weights, float *workspace, void *const *ext, int n_ext, void *stream) {
    CSM_REQUIRE(ops && tensors && n_ops >= 0 && n_tensors > 0);
    return run_ops(ops, n_ops, tensors, n_tensors, weights, workspace, ext, n_ext, (hipStream_t)stream, nullptr);
}

// HIP events released on every exit path (the CSM_HIP early returns inside the timing loops used to leak them)
struct EventSet {
    std::vector<hipEvent_t> ev;
    bool ok = true;
    explicit EventSet(size_t n) {
        ev.reserve(n);
        for (size_t i = 0; i < n; ++i) {
            hipEvent_t e;
            hipError_t r = hipEventCreate(&e);
            if (r != hipSuccess) { csm::set_error("hipEventCreate: %s", hipGetErrorString(r)); ok = false; return; }
            ev.push_back(e);
        }
    }
    ~EventSet() { for (auto e : ev) (void)hipEventDestroy(e); }
    EventSet(const EventSet &) = delete;
    EventSet &operator=(const EventSet &) = delete;
};

// Same as csm_run_program, but brackets every op with HIP events on `stream`, synchronises the stream and returns
// the per-op durations (ms) in op_ms[n_ops].  Measurement aid for bench.py's roofline (not graph-capturable).
extern "C" int csm_run_program_profile(const csm_op *ops, int n_ops, const csm_tensor_desc *tensors, int n_tensors,
                                       const float *weights, float *workspace, void *const *ext, int n_ext, void *stream,
                                       float *op_ms) {
    CSM_REQUIRE(ops && tensors && op_ms && n_ops >= 0 && n_tensors > 0);
    hipStream_t st = (hipStream_t)stream;
    EventSet evs(n_ops + 1);
    if (!evs.ok) return CSM_ERR_HIP;
    int rc = run_ops(ops, n_ops, tensors, n_tensors, weights, workspace, ext, n_ext, st, evs.ev.data());
    if (rc == CSM_OK) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) { csm::set_error("profile sync: %s", hipGetErrorString(e)); rc = CSM_ERR_HIP; }
        else for (int i = 0; i < n_ops; ++i) (void)hipEventElapsedTime(&op_ms[i], evs.ev[i], evs.ev[i + 1]);
    }
    return rc;
}

// layer signature -> tuned tile; process-global, shared by every program and (through the mutex) by host threads that drive
// different GPUs / streams
static std::map<std::array<int, 16>, int> g_tile_cache;
static std::mutex g_tile_mutex;

extern "C" int csm_conv_autotune(csm_op *ops, int n_ops, const csm_tensor_desc *tensors, int n_tensors, const float *weights,
                                 float *workspace, void *const *ext, int n_ext, void *stream, int reps) {
    CSM_REQUIRE(ops && tensors && n_ops >= 0 && n_tensors > 0);
    read_force_env();
    if (g_force_cfg >= 0) return 0;
    if (reps < 1) reps = 3;
    hipStream_t st = (hipStream_t)stream;
    EventSet evs(2);
    if (!evs.ok) return -CSM_ERR_HIP;
    hipEvent_t *ev = evs.ev.data();
    int tuned = 0, rc = CSM_OK;
    for (int i = 0; i < n_ops && rc == CSM_OK; ++i) {
        csm_op &op = ops[i];
        if (op.kind != CSM_OP_CONV || (op.flags & (CSM_CONV_FLAG_STEM | CSM_CONV_FLAG_WINOGRAD | CSM_CONV_FLAG_WINOGRAD4 | CSM_CONV_FLAG_GROUPED))) continue;          // stems, Winograd and vector-pipe grouped layers have one dedicated kernel
        const int npad = (op.cout_g + 31) / 32 * 32;
        static const int cand_all[] = {CFG_64x64, CFG_128x32, CFG_64x16, CFG_D64x64, CFG_D128x64, CFG_D64x128, CFG_D128x128,
                                       CFG_D128x128_8w, CFG_D256x128_8w, CFG_D128x32, CFG_NARROW, CFG_D96x128, CFG_D160x128,
                                       CFG_D224x128, CFG_D192x128, CFG_P64x64, CFG_P128x64, CFG_P64x128, CFG_P128x128, CFG_P256x128,
                                       CFG_P128x32, CFG_P64x64_w8, CFG_P128x128_w8, CFG_P128x32_w8, CFG_P128x128_8w,
                                       CFG_Q64x64, CFG_Q128x64, CFG_Q64x128, CFG_Q128x128_8w, CFG_Q128x32,
                                       CFG_R128x32, CFG_R64x64, CFG_R128x64, CFG_R128x128_8w, CFG_R64x128, CFG_R128x128_8w_o4,
                                       CFG_W256x32, CFG_W256x64};
        static const int cand_bn[] = {64, 32, 16, 64, 64, 128, 128, 128, 128, 32, 4, 128, 128, 128, 128,
                                      64, 64, 128, 128, 128, 32, 64, 128, 32, 128,
                                      64, 64, 128, 128, 32,
                                      32, 64, 64, 128, 128, 128, 32, 64};
        // identical layers (same shapes / strides / split) share one measurement, also across programs
        View vin{}, vout{};
        rc = make_view(tensors, n_tensors, op.in0, workspace, ext, n_ext, vin); if (rc) break;
        rc = make_view(tensors, n_tensors, op.out, workspace, ext, n_ext, vout); if (rc) break;
        const std::array<int, 16> key = {vin.n, vin.h, vin.w, vin.ld, vout.h, vout.w, vout.ld, op.kh, op.kw, op.stride, op.dil,
                                         op.groups, op.cin_g, op.cout_g, op.ksplit, op.pad};
        {
            std::lock_guard<std::mutex> lk(g_tile_mutex);
            auto hit = g_tile_cache.find(key);
            if (hit != g_tile_cache.end()) { op.tile = hit->second; ++tuned; continue; }
        }
        // one timed run of `op` with the tile in op.tile: min over `n` repetitions after one warm-up (which also sets the LDS attribute)
        auto time_tile = [&](int n, float &tmin) -> int {
            tmin = 1e30f;
            for (int r = 0; r <= n; ++r) {
                hipError_t he = hipEventRecord(ev[0], st);
                int rc2 = CSM_OK;
                if (he == hipSuccess) rc2 = run_ops(&op, 1, tensors, n_tensors, weights, workspace, ext, n_ext, st, nullptr);
                if (rc2) return rc2;
                if (he == hipSuccess) he = hipEventRecord(ev[1], st);
                if (he == hipSuccess) he = hipEventSynchronize(ev[1]);
                if (he != hipSuccess) { csm::set_error("conv_autotune timing: %s", hipGetErrorString(he)); return CSM_ERR_HIP; }
                float ms = 0.f; (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
                if (r > 0 && ms < tmin) tmin = ms;
            }
            return CSM_OK;
        };
        float best = 1e30f; int best_cfg = -1;
        std::vector<std::pair<float, int>> timed;
        // warm-up: the first candidates of a layer used to be timed on a GPU that had just idled (clock ramp, cold L2 / Infinity Cache) and
        // measured 3-5 % slower than the same kernel a few milliseconds later -- enough to lose against a slower tile timed afterwards
        {
            op.tile = 0;
            float tw = 0.f;
            for (int w = 0; w < 4 && tw < 4.0f && rc == CSM_OK; ++w) { float t1; rc = time_tile(4, t1); tw += 5.0f * t1; }
            if (rc) break;
        }
        for (size_t c = 0; c < sizeof(cand_all) / sizeof(int); ++c) {
            if (cand_bn[c] >= 2 * npad && cand_bn[c] > 32) continue;      // tile much wider than the output: never wins
            if (cand_all[c] >= CFG_Q64x64) {                                     // persistent blocks: layers that do not split K
                if (cand_all[c] >= CFG_R128x32 && !(op.kh == 3 && op.kw == 3 && op.stride == 1 && op.dil == 1)) continue;
            }
            else if (((cand_all[c] >= CFG_P64x64 && cand_all[c] <= CFG_P128x128_8w) || cand_all[c] == CFG_P256x64) &&
                     !(op.kh == 3 && op.kw == 3 && op.stride == 1 && op.dil == 1)) continue;
            if ((cand_all[c] == CFG_W256x32 || cand_all[c] == CFG_W256x64) &&
                (op.ksplit > 1 || (op.cin_g & 31) || (size_t)kWsPatchBytes + (size_t)9 * ((op.cin_g + 31) / 32) * cand_bn[c] * 128 > (size_t)160 * 1024 ||
                 op.groups * ((op.cout_g + cand_bn[c] - 1) / cand_bn[c]) > 64)) continue;
            if (cand_bn[c] == 4 && (op.cout_g > 4 || op.groups != 1 || op.ksplit > 1)) continue;
            if (cand_bn[c] == 16 && op.cout_g > 16) continue;
            if (cand_bn[c] == 32 && op.cout_g > 64 && (op.cout_g % 64) != 32) continue;   // (96, 160 ... outputs: 32-wide tiles waste no MFMA columns)
            for (int ser = 0; ser <= (op.ksplit > 1 ? 1 : 0) && rc == CSM_OK; ++ser) {     // split-K layers: both executions
                if (cand_all[c] >= CFG_Q64x64 && op.ksplit > 1 && !ser) continue;              // (the persistent kernels walk split K serially only)
                const bool dfam = cand_all[c] >= CFG_D64x64 && cand_all[c] <= CFG_D192x128 && cand_all[c] != CFG_NARROW && cand_all[c] != CFG_D64x64;
                for (int sp = 0; sp <= ((dfam && g_tune_split && (op.ksplit <= 1 || ser)) ? 1 : 0) && rc == CSM_OK; ++sp) {   // mixed-tile launch
                    op.tile = cand_all[c] + 1 + (ser ? kTileSerial : 0) + (sp ? kTileSplit : 0);
                    float tmin;
                    rc = time_tile(reps, tmin);
                    if (rc) break;
                    timed.emplace_back(tmin, op.tile - 1);
                }
            }
            if (rc) break;
        }
        if (rc) break;
        // the three fastest are within a few percent of each other on many layers and one sample of three is noisy: time them again
        // (twice the repetitions, minimum of both rounds) before choosing
        std::sort(timed.begin(), timed.end());
        static const int retime = getenv("CSM_TUNE_RETIME") ? atoi(getenv("CSM_TUNE_RETIME")) : 1;
        if (!retime && !timed.empty()) { best = timed[0].first; best_cfg = timed[0].second; }
        // the finalists (five fastest) are timed again in three INTERLEAVED rounds (a b c d e a b c d e ...), so that a drift of the clock
        // or of the cache state hits all of them alike; the minimum over all of a candidate's samples decides
        const size_t nfin = std::min<size_t>(timed.size(), 5);
        for (int round = 0; retime && round < 3 && rc == CSM_OK; ++round)
            for (size_t k = 0; k < nfin && rc == CSM_OK; ++k) {
                op.tile = timed[k].second + 1;
                float tmin;
                rc = time_tile(reps, tmin);
                if (rc) break;
                if (tmin < timed[k].first) timed[k].first = tmin;
            }
        for (size_t k = 0; retime && k < nfin; ++k)
            if (timed[k].first < best) { best = timed[k].first; best_cfg = timed[k].second; }
        if (rc) break;
        op.tile = best_cfg >= 0 ? best_cfg + 1 : 0;
        { std::lock_guard<std::mutex> lk(g_tile_mutex); g_tile_cache[key] = op.tile; }
        ++tuned;
    }
    return rc == CSM_OK ? tuned : -rc;
}

// Tuned tiles as a text file (one line per layer signature: 16 integers + tile) so that a later process can skip the
// measurement: load merges into the in-memory table that csm_conv_autotune consults first.
extern "C" int csm_conv_tile_cache_save(const char *path) {
    CSM_REQUIRE(path);
    FILE *f = fopen(path, "w");
    if (!f) { csm::set_error("cannot write %s", path); return CSM_ERR_ARG; }
    std::lock_guard<std::mutex> lk(g_tile_mutex);
    for (const auto &kv : g_tile_cache) {
        for (int v : kv.first) fprintf(f, "%d ", v);
        fprintf(f, "%d\n", kv.second);
    }
    fclose(f);
    return CSM_OK;
}

extern "C" int csm_conv_tile_cache_load(const char *path) {
    CSM_REQUIRE(path);
    FILE *f = fopen(path, "r");
    if (!f) return 0;                       // no file yet: nothing cached
    int n = 0;
    for (;;) {
        std::array<int, 16> key; int tile = 0; bool ok = true;
        for (int &v : key) ok = ok && fscanf(f, "%d", &v) == 1;
        if (!ok || fscanf(f, "%d", &tile) != 1) break;
        if (tile >= 0 && (tile & (kTileSerial - 1)) <= CFG_COUNT && tile < 2 * kTileSplit) { std::lock_guard<std::mutex> lk(g_tile_mutex); g_tile_cache[key] = tile; ++n; }
    }
    fclose(f);
    return n;
}

// debug / tuning knob: low byte = forced conv tile configuration (-1 = built-in rule), bits 8.. = phase-ablation flags
// (ConvArgs::dbg).  Not part of the stable ABI.
extern "C" int csm_debug_force_conv_cfg(int cfg) {
    if (cfg >= 0) { g_force_cfg = cfg & 0xff; g_dbg = cfg >> 8; } else { g_force_cfg = -1; g_dbg = 0; }
    return CSM_OK;
}

// measurement aid: which launch forms the autotuner may choose from (bit 0: mixed-tile launches); default all
extern "C" int csm_debug_conv_tuner_options(int options) {
    g_tune_split = options & 1;
    g_ngroup_enable = (options & 2) ? 0 : 1;          // bit 1: N-grouped tile order off
    return CSM_OK;
}

// tests: force how split-K layers are executed (-1 = tuned / rule, 0 = S blocks + reduce kernel, 1 = one block walks the runs)
extern "C" int csm_debug_force_splitk_serial(int mode) {
    g_force_serial = mode < 0 ? -1 : (mode ? 1 : 0);
    return CSM_OK;
}
