// wino.hip -- exact-fp32 Winograd F(2x2, 3x3) convolution on the matrix pipe (gfx950), fused: input transform in registers,
// 16 frequency GEMMs on v_mfma_f32_32x32x2_f32, output transform + bias + residual + activation in the epilogue.
//
// Why: 3x3 / stride-1 dense convolutions carry two thirds of the dense nets' time (ISNet almost entirely; LeReS' decoder; RTMDet's
// CSP blocks) and the direct implicit-GEMM kernels of nets.hip run them at 0.87-0.91 of the fp32 MFMA peak -- there is nothing left in
// the inner loop, so the lever is the FLOP count: F(2x2, 3x3) needs 16 multiplications per 2x2 output tile and channel pair where the
// direct form needs 36 (2.25x), with transform matrices made of 0, +-1, +-1/2 (error: a few fp32 ulps).
//
// Arithmetic = the "Winograd contract" of include/csm355.h, restated independently in oracle/nets_oracle.c::orc_conv_wino; this
// kernel reproduces it bit for bit (every transform value is ONE fp32 operation, every product sum ONE fmaf chain in the direct
// contract's channel order -- fp32 MFMA is bitwise an fmaf chain over k).
//
// Mapping (CDNA4-first):
//   * block = 4 waves (WM x WN = 2 x 2), ONE block per CU (512 registers per wave): 16 x 4 Winograd tiles (32 x 8 output pixels) x 64
//     output channels.  A wave owns 32 tiles x 32 channels x ALL 16 frequencies = sixteen 32x32 accumulators (256 AGPRs), so the output
//     transform is lane-local: no exchange, no extra pass over HBM.
//   * the MFMA A operand never exists in memory: a lane (tile, k-half) reads its tile's 4 x 4 raw input window from the LDS patch
//     (16 ds_read_b128 = 4 channels each), does the 32 adds of B^T d B per channel in registers (VALU issues in the shadow of the
//     64-cycle fp32 MFMAs) and feeds 64 MFMAs with the result.  LDS holds only the raw (OH + 2) x (OW + 2) x 32-channel patch, moved by
//     LDS-DMA, stored de-interleaved by column parity so that the stride-2 tile windows of a lane group are consecutive 128-B rows
//     (XOR-swizzled 16-B slots: conflict-free ds_read_b128).
//   * transformed weights U are packed on the host in exactly the LDS image of one pipeline step (8 input channels: [f][k-half][64 co][4]
//     = 32 KB), so the B side is a linear LDS-DMA copy and a B fragment is ONE conflict-free ds_read_b128 per frequency.
//   * pipeline: step = 8 input channels = 64 MFMAs per wave (4096 cycles); ONE barrier per step, placed in front of the step's last
//     MFMA group, after which the other U stage is known to have landed and the stage just read is refilled -- every DMA piece has a
//     full step to land; fragments, raw windows and transform values are all produced one MFMA group ahead of their use.
#include "csm_conv.h"
#include <utility>
#include <cstdlib>

using namespace csmconv;

namespace {

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const f32x4n *lds_f4_ptr;

__device__ __forceinline__ f32x4n lds_read4(unsigned byte_addr) { return *(lds_f4_ptr)(size_t)byte_addr; }

// LDS-DMA piece with a scalar byte offset on the global side (no address VALU): 64 lanes x 16 B -> LDS [lds_byte_addr, +1 KB)
__device__ __forceinline__ void dma16s(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte_addr) : "memory");
}

template <int N> using ic = std::integral_constant<int, N>;
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// raw barrier with its waits in ONE asm block (the wait can never be separated from the barrier by a basic-block boundary or by a sunk
// LDS read: tools/check_isa_barriers.py): everybody's DMA pieces counted by vmcnt have landed, everybody's LDS reads have completed
template <int VM> __device__ __forceinline__ void wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(VM) : "memory");
}

// B^T rows: one fp32 operation per value (the contract's order)
__device__ __forceinline__ float bt_row(int i, float d0, float d1, float d2, float d3) {
    return i == 0 ? d0 - d2 : (i == 1 ? d1 + d2 : (i == 2 ? d2 - d1 : d1 - d3));
}

// ABL: development ablations of the main loop (timing only, results invalid): 1 = no DMA, 2 = no barrier / waits, 4 = no LDS reads,
// 8 = no transform VALU; OPT bits: 1 = counted vmcnt at the barriers (the patch pieces of this step may stay in flight) and the whole
// U stage issued behind the barrier
template <int WM, int WN, int ABL = 0, int OPT = 0>
__global__ __launch_bounds__(64 * WM * WN, 1) void k_conv_wino(ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int NW = WM * WN;
    constexpr int OH = 4 * WM, OW = 32;                       // output pixels of a block
    constexpr int PH = OH + 2, PWH = 18;                       // patch rows; entries per (parity, row): px = 2 pxh + parity < OW + 2, padded to even
    constexpr int NENT = 2 * PH * PWH, NPP = (NENT + 7) / 8;   // patch entries (128 B each) / DMA pieces
    constexpr int QP = (NPP + NW - 1) / NW;                    // patch pieces per wave
    constexpr int QG = (QP + 2) / 3;                           // ... issued in three steps
    constexpr unsigned kPatchB = NPP * 1024u, kUB = 32768u, kU0 = 2u * kPatchB;
    constexpr int UPW = 32 / NW;                               // U pieces per wave per stage
    static_assert(WN == 2 && UPW * NW == 32 && (NENT % 8) == 0 && (PWH & 1) == 0, "tile shape");
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [patch 0][patch 1][U 0][U 1]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    int mt, ntile, zz;
    block_to_tile(mt, ntile, zz, 0);
    const int btx = mt % tiles_x, bty = (mt / tiles_x) % tiles_y, n = mt / (tiles_x * tiles_y);
    const int ho = a.out.h, wo = a.out.w;
    const int ncb = a.ncb, nsteps = 4 * ncb;
    const int oy0 = bty * OH, ox0 = btx * OW;

    i32x4 ra, rb;
    {
        uint64_t pa = (uint64_t)a.in.p, pb = (uint64_t)a.w;
        unsigned na = (unsigned)((((int64_t)a.in.n * a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4);
        unsigned nb = (unsigned)((int64_t)(a.cout_g / 64) * nsteps * kUB);
        ra = i32x4{(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)na, 0x00020000};
        rb = i32x4{(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)nb, 0x00020000};
    }
    // ---- patch loader: wave w owns pieces w, w + NW, ... (a piece past the end repeats the last one: same bytes, same place) ----
    // entry e = (parity * PH + py) * PWH + pxh holds patch pixel (py, 2 pxh + parity); its 16-B slot s sits at physical slot s ^ key,
    // key = (pxh >> 1) & 7: the 16 lanes of a ds_read_b128 group read 16 distinct pxh (mod 16) of rows whose first entry is even.
    unsigned offP[QP], ldsP[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        int p = wave + q * NW;
        if (p > NPP - 1) p = NPP - 1;
        const int e = 8 * p + (lane >> 3);
        const int par = e / (PH * PWH), rem = e - par * (PH * PWH), py = rem / PWH, pxh = rem - py * PWH;
        const int px = 2 * pxh + par, slot = (lane & 7) ^ ((pxh >> 1) & 7);
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool v = px < OW + 2 && iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w;
        offP[q] = v ? (unsigned)(((n * a.in.h + iy) * a.in.w + ix) * a.in.ld + slot * 4) * 4u : kOob;
        ldsP[q] = (unsigned)p * 1024u;
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned voffU = (unsigned)lane * 16u;
    unsigned u_src = (unsigned)(ntile * nsteps) * kUB + (unsigned)(wave * UPW) * 1024u;   // this wave's first piece of the NEXT U stage to fetch
    const unsigned ldsU = lds0 + kU0 + (unsigned)(wave * UPW) * 1024u;

    // pieces [k0, k1) of U stage `u_src` -> LDS stage par
    auto issue_u = [&](int par, int k0, int k1, bool live) {
        const unsigned vo = live ? voffU : kOob;
#pragma unroll
        for (int k = k0; k < k1; ++k) dma16s(vo, rb, u_src + (unsigned)k * 1024u, ldsU + (unsigned)par * kUB + (unsigned)k * 1024u);
    };
    // pieces [q0, q1) of the patch of channel block cbn -> patch stage cbn & 1
    auto issue_patch = [&](int cbn, int q0, int q1, bool live) {
        const unsigned sb = lds0 + (unsigned)(cbn & 1) * kPatchB;
#pragma unroll
        for (int q = q0; q < q1; ++q)
            if (q < QP) dma16s(live ? offP[q] : kOob, ra, (unsigned)cbn * 128u, sb + ldsP[q]);
    };

    // ---- fragment addresses ----
    // raw window of the lane's tile (tx = li & 15, ty = 2 wm + (li >> 4)): position (i, j) is entry e0 + ((j & 1) PH + i) PWH + (j >> 1)
    const int tx = li & 15, ty = 2 * wm + (li >> 4);
    unsigned rbase[2];
#pragma unroll
    for (int jh = 0; jh < 2; ++jh)
        rbase[jh] = lds0 + (unsigned)((2 * ty) * PWH + tx) * 128u + (unsigned)((lh ^ (((tx + jh) >> 1) & 7)) << 4);
    const unsigned ub = lds0 + kU0 + (unsigned)(lh * 64 + wn * 32 + li) * 16u;

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;

    float R[16][4];           // raw window of the NEXT step (4 channels per position)
    float T[4][4][4];         // row-transformed window of the current step: T[i][j][c]
    float V[2][4][4];         // A operands of an MFMA group: V[buf][j][c]
    f32x4n Bq[2][4];          // B fragments of an MFMA group

    auto read_raw = [&](int cbn, int sub, int p0, int p1) {            // positions [p0, p1) of step (cbn, sub) -> R
        unsigned b[2];
#pragma unroll
        for (int jh = 0; jh < 2; ++jh) b[jh] = (rbase[jh] + (unsigned)(cbn & 1) * kPatchB) ^ (unsigned)(sub << 5);
#pragma unroll
        for (int p = p0; p < p1; ++p) {
            const int i = p >> 2, j = p & 3;
            const f32x4n v = lds_read4(b[j >> 1] + (unsigned)((((j & 1) * PH + i) * PWH + (j >> 1)) * 128));
            R[p][0] = v.x; R[p][1] = v.y; R[p][2] = v.z; R[p][3] = v.w;
        }
    };
    auto read_b = [&](int par, int g, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) Bq[buf][j] = lds_read4(ub + (unsigned)par * kUB + (unsigned)(4 * g + j) * 2048u);
    };
    // op k of the row pass (64 ops): T[i][j][c] from R, k = 16 i + 4 j + c
    auto row_op = [&](auto K) {
        constexpr int k = decltype(K)::value, i = k >> 4, j = (k >> 2) & 3, c = k & 3;
        T[i][j][c] = bt_row(i, R[0 + j][c], R[4 + j][c], R[8 + j][c], R[12 + j][c]);
    };
    // op k of the column pass of group g (16 ops): V[buf][j][c] from T[g], k = 4 j + c
    auto col_op = [&](auto G, auto BUF, auto K) {
        constexpr int g = decltype(G)::value, buf = decltype(BUF)::value, k = decltype(K)::value, j = k >> 2, c = k & 3;
        V[buf][j][c] = bt_row(j, T[g][0][c], T[g][1][c], T[g][2][c], T[g][3][c]);
    };

    // ---- DMA schedule (every piece gets at least three MFMA groups = 3 000 cycles to land before the barrier that publishes it) ----
    //   U stage of step s + 2 -> the stage step s read: first half behind the barrier of step s (group 3), second half in group 0 of step
    //   s + 1.  Patch of channel block cb + 2 -> the stage block cb read (released by the barrier of step (cb, 2)): a third each in group
    //   1 of steps (cb, 3), (cb + 1, 0), (cb + 1, 1); it is first read in step (cb + 1, 3), behind the barrier of (cb + 1, 2).
    // ---- prologue ----
    issue_patch(0, 0, QP, true);
    issue_u(0, 0, UPW, true); u_src += kUB;
    if constexpr (OPT & 1) { issue_u(1, 0, UPW, nsteps > 1); u_src += kUB; }
    else issue_u(1, 0, UPW / 2, nsteps > 1);
    issue_patch(1, 0, QG, ncb > 1);
    wait_barrier<0>();
    read_raw(0, 0, 0, 16);
    read_b(0, 0, 0);
    static_for<64>([&](auto K) { row_op(K); });
    static_for<16>([&](auto K) { col_op(ic<0>{}, ic<0>{}, K); });

    // ---- main loop: one iteration = one 32-channel block = 4 steps x 4 MFMA groups of 16 ----
    for (int cb = 0; cb < ncb; ++cb) {
        static_for<4>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            const int cbn = q == 3 ? cb + 1 : cb;                  // channel block / sub-step of the NEXT step
            constexpr int subn = (q + 1) & 3;
            const int s = 4 * cb + q;
            static_for<4>([&](auto G) {
                constexpr int g = decltype(G)::value, buf = g & 1;
                if constexpr (g == 3) {
                    // everybody's pieces of the next U stage (and the older patch pieces) have landed; everybody has finished reading this
                    // step's U stage and -- at q == 2 -- this channel block's patch.  OPT 1: the patch pieces sent in group 1 of THIS step
                    // (q != 2) are the youngest loads in flight and are not needed before the barrier of step (cb + 1, 2): they may stay
                    // outstanding (loads retire in order)
                    if constexpr (!(ABL & 2)) {
                        if constexpr ((OPT & 1) && q != 2) wait_barrier<(QG + 3) / 4 * 4>();
                        else wait_barrier<0>();
                    }
                    if constexpr (!(ABL & 4)) read_b((q + 1) & 1, 0, buf ^ 1);
                } else if constexpr (!(ABL & 4)) {
                    read_b(q & 1, g + 1, buf ^ 1);
                    read_raw(cbn, subn, g == 0 ? 0 : (g == 1 ? 6 : 11), g == 0 ? 6 : (g == 1 ? 11 : 16));
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<16>([&](auto M) {
                    constexpr int m = decltype(M)::value, tt = m >> 2, j = m & 3;
                    const f32x4n bf = Bq[buf][j];
                    const float bv = tt == 0 ? bf.x : (tt == 1 ? bf.y : (tt == 2 ? bf.z : bf.w));
                    acc[4 * g + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[buf][j][tt], bv, acc[4 * g + j], 0, 0, 0);
                    // this slot's share of the transform work for the NEXT group
                    if constexpr (ABL & 8) {
                    } else if constexpr (g < 3) {
                        col_op(ic<g + 1>{}, ic<(buf ^ 1)>{}, M);
                    } else {                                        // next step: row pass (64 ops), then group 0's column pass (16)
                        static_for<5>([&](auto E) {
                            constexpr int k = 5 * m + decltype(E)::value;
                            if constexpr (k < 64) row_op(ic<k>{});
                            else if constexpr (k < 80) col_op(ic<0>{}, ic<0>{}, ic<k - 64>{});
                        });
                    }
                    // DMA pieces in MFMA slots 1, 5, 9, 13 (OPT 1: the whole U stage behind the barrier, slots 1, 3, ..., 15)
                    if constexpr (ABL & 1) {
                    } else if constexpr ((OPT & 1) && g == 3 && (m & 1) == 1) {
                        issue_u(q & 1, (m >> 1) * (UPW / 8), ((m >> 1) + 1) * (UPW / 8), s + 2 < nsteps);
                        if constexpr (m == 15) u_src += kUB;
                    } else if constexpr ((m & 3) == 1) {
                        constexpr int k = m >> 2;
                        if constexpr ((OPT & 1) && (g == 3 || g == 0)) {
                        } else if constexpr (g == 3) {
                            issue_u(q & 1, k * (UPW / 8), (k + 1) * (UPW / 8), s + 2 < nsteps);
                        } else if constexpr (g == 0) {
                            issue_u((q + 1) & 1, UPW / 2 + k * (UPW / 8), UPW / 2 + (k + 1) * (UPW / 8), s + 1 < nsteps);
                            if constexpr (k == 3) u_src += kUB;
                        } else if constexpr (g == 1 && q != 2) {
                            constexpr int grp = q == 3 ? 0 : q;                       // q = 3, 0, 1 -> thirds 0, 1, 2 ... (q == 0 -> 1, q == 1 -> 2)
                            constexpr int third = q == 3 ? 0 : (q == 0 ? 1 : 2);
                            (void)grp;
                            const int cbp = q == 3 ? cb + 2 : cb + 1;
                            constexpr int per = (QG + 3) / 4;
                            issue_patch(cbp, third * QG + k * per, third * QG + (k + 1) * per < (third + 1) * QG ? third * QG + (k + 1) * per : (third + 1) * QG, cbp < ncb);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing (dead) fetches must land before the block's LDS is released

    // ---- epilogue: output transform A^T M A (columns j first), bias, residual, activation.  Lane (li, lh) holds output channel li of the
    // wave's 32 and, in accumulator element r, the tile (tx, ty) = ((r & 3) + 8 ((r >> 2) & 1) + 4 lh, r >> 3) of the wave's 16 x 2.
    // Everything is indexed at compile time (a run-time r would turn the 256 accumulator registers into a waterfall) and the activation
    // switch sits OUTSIDE the element loops: the 64 outputs of a lane are formed in registers first ----
    const int co = ntile * 64 + 32 * wn + li;
    const float bias = a.bias ? a.bias[co] : 0.0f;
    const float slope = a.slope ? a.slope[co] : 0.0f;
    const int64_t ldo = a.out.ld, ldr = a.res.ld;
    const int oyb = oy0 + 4 * wm, oxb = ox0 + 8 * lh;
    const int64_t mb = ((int64_t)n * ho + oyb) * wo + oxb;
    float *ob = a.out.p + mb * ldo + co;
    const float *rp = a.res_mode ? a.res.p + mb * ldr + co : nullptr;
    const bool full = oy0 + OH <= ho && ox0 + OW <= wo;        // block-uniform: interior block tiles take the unguarded path
    // ACT: compile-time activation (-1 = the run-time switch of apply_act); FULL: no bounds checks
    auto epilogue = [&](auto ACT, auto FULL, auto RES) {
        constexpr int act_c = decltype(ACT)::value;
        constexpr bool has_res = decltype(RES)::value;
        // residual: all 64 loads of a lane go out back to back BEFORE the first use (one wait instead of 64 load round trips in series)
        float resv[has_res ? 64 : 1];
        if constexpr (has_res)
            static_for<64>([&](auto E) {
                constexpr int e = decltype(E)::value, r = e >> 2, aa = (e >> 1) & 1, b = e & 1;
                constexpr int dy = 2 * (r >> 3) + aa, dx = 2 * ((r & 3) + 8 * ((r >> 2) & 1)) + b;
                const int64_t eo = (int64_t)dy * wo + dx;
                if constexpr (decltype(FULL)::value) resv[e] = rp[eo * ldr];
                else resv[e] = (oyb + dy < ho && oxb + dx < wo) ? rp[eo * ldr] : 0.0f;
            });
        static_for<16>([&](auto RR) {
            constexpr int r = decltype(RR)::value;
            float sj[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float m0 = acc[4 * i][r], m1 = acc[4 * i + 1][r], m2 = acc[4 * i + 2][r], m3 = acc[4 * i + 3][r];
                sj[i][0] = (m0 + m1) + m2;
                sj[i][1] = (m1 - m2) - m3;
            }
            static_for<4>([&](auto P) {
                constexpr int aa = decltype(P)::value >> 1, b = decltype(P)::value & 1;
                constexpr int dy = 2 * (r >> 3) + aa, dx = 2 * ((r & 3) + 8 * ((r >> 2) & 1)) + b;
                float v = (aa == 0 ? (sj[0][b] + sj[1][b]) + sj[2][b] : (sj[1][b] - sj[2][b]) - sj[3][b]) + bias;
                auto finish = [&]() {
                    const int64_t eo = (int64_t)dy * wo + dx;
                    if constexpr (has_res) { if (a.res_mode == 1) v += resv[4 * r + 2 * aa + b]; }
                    v = apply_act(v, act_c >= 0 ? act_c : a.act, slope);
                    if constexpr (has_res) { if (a.res_mode == 2) v += resv[4 * r + 2 * aa + b]; }
                    ob[eo * ldo] = v;
                };
                if constexpr (decltype(FULL)::value) finish();
                else if (oyb + dy < ho && oxb + dx < wo) finish();
            });
        });
    };
    auto epilogue_act = [&](auto FULL, auto RES) {
        switch (a.act) {                                       // uniform: the common activations get a body without the per-element switch
            case CSM_ACT_NONE: epilogue(ic<CSM_ACT_NONE>{}, FULL, RES); break;
            case CSM_ACT_RELU: epilogue(ic<CSM_ACT_RELU>{}, FULL, RES); break;
            case CSM_ACT_SILU: epilogue(ic<CSM_ACT_SILU>{}, FULL, RES); break;
            case CSM_ACT_PRELU: epilogue(ic<CSM_ACT_PRELU>{}, FULL, RES); break;
            default: epilogue(ic<-1>{}, FULL, RES); break;
        }
    };
    if (a.res_mode) { if (full) epilogue_act(std::true_type{}, std::true_type{}); else epilogue_act(std::false_type{}, std::true_type{}); }
    else if (full) epilogue_act(std::true_type{}, std::false_type{});
    else epilogue_act(std::false_type{}, std::false_type{});
}


// ---- eight-wave form: two waves per SIMD ----------------------------------------------------------------------------------------------
// k_conv_wino runs ONE wave per SIMD (its 16 accumulators + operands take the whole 512-register file of a lane), and the counters say
// what that costs (profiles/r05_wino_ablation.txt): nothing overlaps the matrix pipe but the wave's own next instructions, and the issue
// time of the transform VALU, the LDS reads and above all the LDS-DMA pieces (~85 cycles each, eleven per step) shows up one for one --
// the loop runs at 0.63 of the MFMA rate although no unit is busy.  Here the SAME block tile (16 x 4 Winograd tiles x 64 channels, same
// LDS image, same packed weights, same arithmetic, bit for bit) is worked by EIGHT waves: wave (fh, wm, wn) owns the frequency rows
// i = 2 fh, 2 fh + 1 of its 32 tiles x 32 channels -- eight accumulators, 128 registers -- so two waves share a SIMD and one's loads,
// transforms and DMA issue run under the other's MFMAs.  A frequency row needs only two of the four raw window rows (B^T has two
// non-zeros per row), so the transform work per wave halves as well.  The price is the output transform's row pass, which now spans two
// waves: after the main loop each wave does the column pass on its two rows in registers, the partner waves swap half of the results
// through LDS (32 floats per lane, the memory of the patch stages) and each finishes the outputs of half the tiles.
// Pipeline of a step (8 channels = 2 MFMA groups of 16 per wave), every register buffer single:
//   phase A: MFMA group 0 | B fragments of group 1 | transform of group 1's window rows (R -> V[1]) | R <- group-0 rows of step s + 1
//   barrier  (the only one: the next U stage has landed, this step's U stage and -- at q == 3 -- this channel block's patch are released)
//   phase B: MFMA group 1 | B fragments of group 0 of step s + 1 | DMA: U stage s + 2, a third of a patch | R -> V[0] of s + 1 | R <- group-1 rows
// GEO: block-tile geometry (speed only: an output's arithmetic does not depend on the tile it falls in).  0 = 16 x 4 Winograd tiles (32 x 8
// output pixels), 1 = 8 x 8 tiles (16 x 16 pixels): the square tile wastes nothing on 80 x 80 maps (a 32-wide tile pads them to 96) and
// less on 40 / 45-pixel maps; the launcher takes the geometry that covers the map with fewer block tiles.  With 8-wide tiles a
// ds_read_b128 lane group spans four tile rows, so the slot swizzle is keyed on the patch row as well: (px/2 >> 1) ^ 4 ((py >> 1) & 1).
template <int ABL = 0, int GEO = 0>
__global__ __launch_bounds__(512, 2) void k_conv_wino8(ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int NW = 8;
    constexpr int OH = GEO ? 16 : 8, OW = GEO ? 16 : 32, PH = OH + 2, PWH = GEO ? 10 : 18;
    constexpr int NENT = 2 * PH * PWH, NPP = (NENT + 7) / 8;   // 360 entries, 45 pieces
    constexpr int QP = (NPP + NW - 1) / NW, QG = (QP + 2) / 3;  // 6 patch pieces per wave, 2 per third
    constexpr unsigned kPatchB = NPP * 1024u, kUB = 32768u, kU0 = 2u * kPatchB;
    constexpr int UPW = 32 / NW;                               // 4 U pieces per wave per stage
    constexpr unsigned kOob = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [patch 0][patch 1][U 0][U 1]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    int mt, ntile, zz;
    block_to_tile(mt, ntile, zz, 0);
    const int btx = mt % tiles_x, bty = (mt / tiles_x) % tiles_y, n = mt / (tiles_x * tiles_y);
    const int ho = a.out.h, wo = a.out.w;
    const int ncb = a.ncb, nsteps = 4 * ncb;
    const int oy0 = bty * OH, ox0 = btx * OW;

    i32x4 ra, rb;
    {
        uint64_t pa = (uint64_t)a.in.p, pb = (uint64_t)a.w;
        unsigned na = (unsigned)((((int64_t)a.in.n * a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4);
        unsigned nb = (unsigned)((int64_t)(a.cout_g / 64) * nsteps * kUB);
        ra = i32x4{(int)(unsigned)pa, (int)(unsigned)(pa >> 32), (int)na, 0x00020000};
        rb = i32x4{(int)(unsigned)pb, (int)(unsigned)(pb >> 32), (int)nb, 0x00020000};
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const unsigned voffU = (unsigned)lane * 16u;
    unsigned u_src = (unsigned)(ntile * nsteps) * kUB + (unsigned)(wave * UPW) * 1024u;
    const unsigned ldsU = lds0 + kU0 + (unsigned)(wave * UPW) * 1024u;
    auto issue_u1 = [&](int par, int k, bool live) {
        dma16s(live ? voffU : kOob, rb, u_src + (unsigned)k * 1024u, ldsU + (unsigned)par * kUB + (unsigned)k * 1024u);
    };
    // the first two U stages go out before anything else is computed: their addresses are scalar, and the round trip runs under the
    // set-up of the patch offsets and the zeroing of the accumulators (the prologue of a 64-channel tile is a tenth of its life)
#pragma unroll
    for (int k = 0; k < UPW; ++k) issue_u1(0, k, true);
    u_src += kUB;
#pragma unroll
    for (int k = 0; k < UPW; ++k) issue_u1(1, k, nsteps > 1);
    u_src += kUB;
    unsigned offP[QP], ldsP[QP];
#pragma unroll
    for (int q = 0; q < QP; ++q) {
        int p = wave + q * NW;
        if (p > NPP - 1) p = NPP - 1;
        const int e = 8 * p + (lane >> 3);
        const int par = e / (PH * PWH), rem = e - par * (PH * PWH), py = rem / PWH, pxh = rem - py * PWH;
        const int px = 2 * pxh + par, slot = (lane & 7) ^ (((pxh >> 1) ^ (GEO ? 4 * ((py >> 1) & 1) : 0)) & 7);
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        const bool v = px < OW + 2 && iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w;
        offP[q] = v ? (unsigned)(((n * a.in.h + iy) * a.in.w + ix) * a.in.ld + slot * 4) * 4u : kOob;
        ldsP[q] = (unsigned)p * 1024u;
    }
    auto issue_patch1 = [&](int cbn, int q, bool live) {
        dma16s(live ? offP[q] : kOob, ra, (unsigned)cbn * 128u, lds0 + (unsigned)(cbn & 1) * kPatchB + ldsP[q]);
    };
#pragma unroll
    for (int q = 0; q < QP; ++q) issue_patch1(0, q, true);
#pragma unroll
    for (int q = 0; q < QG; ++q) issue_patch1(1, q, ncb > 1);

    const int tx = GEO ? (li & 7) : (li & 15), ty = GEO ? 4 * wm + (li >> 3) : 2 * wm + (li >> 4);
    // window position (i, j) of the lane's tile: entry e0 + ((j & 1) PH + i) PWH + (j >> 1), slot (2 sub + lh) ^ key; the key of GEO 1
    // carries the row bit ((2 ty + i) >> 1) & 1 = (ty + (i >> 1)) & 1: folded in here for i < 2, flipped (address bit 6) for i >= 2
    unsigned rbase[2];
#pragma unroll
    for (int jh = 0; jh < 2; ++jh)
        rbase[jh] = lds0 + (unsigned)((2 * ty) * PWH + tx) * 128u + (unsigned)((lh ^ ((((tx + jh) >> 1) ^ (GEO ? 4 * (ty & 1) : 0)) & 7)) << 4);
    const unsigned ub = lds0 + kU0 + (unsigned)(lh * 64 + wn * 32 + li) * 16u;

    f32x16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.0f;

    auto body = [&](auto FH) {
        constexpr int kfh = decltype(FH)::value;
        // frequency row i = 2 kfh + gg is  d[rowA] (+/-) d[rowB]  over the window rows (B^T): i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
        auto rowA = [](int i) constexpr { return i == 0 ? 0 : (i == 2 ? 2 : 1); };
        auto rowB = [](int i) constexpr { return i == 0 ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3)); };
        float R[8][4];            // window rows of ONE group: [0..3] = rowA columns 0..3 (overwritten by the row pass), [4..7] = rowB
        float V[2][4][4];         // A operands of the two groups
        f32x4n Bq[2][4];          // B fragments of the two groups

        auto read_raw = [&](auto GG, int cbn, int sub) {       // the two window rows of group gg of step (cbn, sub) -> R
            constexpr int i = 2 * kfh + decltype(GG)::value;
            unsigned b[2];
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) b[jh] = (rbase[jh] + (unsigned)(cbn & 1) * kPatchB) ^ (unsigned)(sub << 5);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int row = p < 4 ? rowA(i) : rowB(i), j = p & 3;
                const f32x4n v = lds_read4((b[j >> 1] ^ (unsigned)((GEO && row >= 2) ? 64 : 0)) + (unsigned)((((j & 1) * PH + row) * PWH + (j >> 1)) * 128));
                R[p][0] = v.x; R[p][1] = v.y; R[p][2] = v.z; R[p][3] = v.w;
            }
        };
        auto read_b = [&](auto GG, int par) {
            constexpr int gg = decltype(GG)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) Bq[gg][j] = lds_read4(ub + (unsigned)par * kUB + (unsigned)(4 * (2 * kfh + gg) + j) * 2048u);
        };
        // transform op k of group gg (32 ops): k < 16: row pass, R[j][c] = R[j][c] (+/-) R[4 + j][c]; k >= 16: column pass into V[gg]
        auto xf_op = [&](auto GG, auto K) {
            constexpr int gg = decltype(GG)::value, i = 2 * kfh + gg, k = decltype(K)::value;
            if constexpr (k < 16) {
                constexpr int j = k >> 2, c = k & 3;
                R[j][c] = (i == 1) ? R[j][c] + R[4 + j][c] : R[j][c] - R[4 + j][c];
            } else {
                constexpr int j = (k - 16) >> 2, c = k & 3;
                V[gg][j][c] = bt_row(j, R[0][c], R[1][c], R[2][c], R[3][c]);
            }
        };
        // one MFMA group with its prefetch / transform / DMA work spread over the 16 slots.
        //   NG: the group whose operands are prepared during this phase (the other one); its window rows are in R on entry
        auto phase = [&](auto GG, auto Q, int cb) {
            constexpr int gg = decltype(GG)::value, q = decltype(Q)::value;
            const int s = 4 * cb + q;
            // operands being prepared: phase A (gg = 0) prepares group 1 of THIS step, phase B prepares group 0 of the NEXT step
            const int cbn = q == 3 ? cb + 1 : cb;
            constexpr int subn = (q + 1) & 3;
            if constexpr (gg == 1) {
                if constexpr (!(ABL & 2)) wait_barrier<0>();
                if constexpr (!(ABL & 4)) read_b(ic<0>{}, (q + 1) & 1);
            } else if constexpr (!(ABL & 4)) read_b(ic<1>{}, q & 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<16>([&](auto M) {
                constexpr int m = decltype(M)::value, tt = m >> 2, j = m & 3;
                const f32x4n bf = Bq[gg][j];
                const float bv = tt == 0 ? bf.x : (tt == 1 ? bf.y : (tt == 2 ? bf.z : bf.w));
                acc[4 * gg + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[gg][j][tt], bv, acc[4 * gg + j], 0, 0, 0);
                if constexpr (m < 8 && !(ABL & 8)) {
                    static_for<4>([&](auto E) { xf_op(ic<(gg ^ 1)>{}, ic<4 * m + decltype(E)::value>{}); });
                }
                if constexpr (m == 8 && !(ABL & 4)) {           // R is free: the window rows of the group prepared in the NEXT phase
                    if constexpr (gg == 0) read_raw(ic<0>{}, cbn, subn);          // (phase B prepares group 0 of step s + 1)
                    else read_raw(ic<1>{}, cbn, subn);                             // (phase A of step s + 1 prepares its group 1)
                }
                if constexpr (gg == 1 && !(ABL & 1)) {
                    if constexpr (m >= 9 && m < 9 + UPW) {
                        issue_u1(q & 1, m - 9, s + 2 < nsteps);
                        if constexpr (m == 9 + UPW - 1) u_src += kUB;
                    }
                    if constexpr (q != 2 && m >= 13 && m < 13 + QG) {
                        constexpr int third = q == 3 ? 0 : (q == 0 ? 1 : 2);
                        const int cbp = q == 3 ? cb + 2 : cb + 1;
                        if constexpr (third * QG + (m - 13) < QP) issue_patch1(cbp, third * QG + (m - 13), cbp < ncb);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        // prologue (U 0, U 1, patch 0 and the first third of patch 1 are in flight): the operands of group 0 of step 0, the window rows of its group 1
        wait_barrier<0>();
        read_b(ic<0>{}, 0);
        read_raw(ic<0>{}, 0, 0);
        static_for<32>([&](auto K) { xf_op(ic<0>{}, K); });
        read_raw(ic<1>{}, 0, 0);

        for (int cb = 0; cb < ncb; ++cb) {
            static_for<4>([&](auto Q) {
                phase(ic<0>{}, Q, cb);
                phase(ic<1>{}, Q, cb);
            });
        }
        wait_barrier<0>();        // trailing (dead) fetches landed, trailing prefetch reads done, for every wave: the patch stages become the exchange area

        // ---- output transform.  Column pass (j) on this wave's two frequency rows, in registers: s[ii][b] per accumulator element r ----
        // rows i = 2 kfh + ii.  The row pass needs all four rows: wave fh = 0 finishes tiles r < 8, wave fh = 1 tiles r >= 8; each sends
        // the partner its s values of the OTHER eight elements (32 floats per lane) through LDS.
        constexpr int keep0 = 8 * kfh, send0 = 8 * (kfh ^ 1);
        const unsigned xw = lds0 + (unsigned)wave * 8192u + (unsigned)lane * 16u, xr = lds0 + (unsigned)(wave ^ 4) * 8192u + (unsigned)lane * 16u;
        static_for<8>([&](auto RR) {
            constexpr int r = send0 + decltype(RR)::value;
            f32x4n v;
            v.x = (acc[0][r] + acc[1][r]) + acc[2][r]; v.y = (acc[1][r] - acc[2][r]) - acc[3][r];
            v.z = (acc[4][r] + acc[5][r]) + acc[6][r]; v.w = (acc[5][r] - acc[6][r]) - acc[7][r];
            *(__attribute__((address_space(3))) f32x4n *)(size_t)(xw + (unsigned)decltype(RR)::value * 1024u) = v;
        });
        wait_barrier<0>();
        const int co = ntile * 64 + 32 * wn + li;
        const float bias = a.bias ? a.bias[co] : 0.0f;
        const float slope = a.slope ? a.slope[co] : 0.0f;
        const int64_t ldo = a.out.ld, ldr = a.res.ld;
        // the tiles this wave finishes (accumulator elements 8 kfh + rr): GEO 0: tile row kfh of the wave's two, columns (rr & 3) + 8 ((rr >> 2) & 1)
        // + 4 lh; GEO 1: tile rows 2 kfh + (rr >> 2) of the wave's four, columns (rr & 3) + 4 lh
        const int oyb = oy0 + (GEO ? 8 * wm + 4 * kfh : 4 * wm + 2 * kfh), oxb = ox0 + 8 * lh;
        const int64_t mb = ((int64_t)n * ho + oyb) * wo + oxb;
        float *ob = a.out.p + mb * ldo + co;
        const float *rp = a.res_mode ? a.res.p + mb * ldr + co : nullptr;
        const bool full = oy0 + OH <= ho && ox0 + OW <= wo;
        auto epilogue = [&](auto ACT, auto FULL, auto RES) {
            constexpr int act_c = decltype(ACT)::value;
            constexpr bool has_res = decltype(RES)::value;
            float resv[has_res ? 32 : 1];
            if constexpr (has_res)
                static_for<32>([&](auto E) {
                    constexpr int e = decltype(E)::value, rr = e >> 2, aa = (e >> 1) & 1, b = e & 1;
                    constexpr int dy = GEO ? 2 * (rr >> 2) + aa : aa, dx = GEO ? 2 * (rr & 3) + b : 2 * ((rr & 3) + 8 * ((rr >> 2) & 1)) + b;
                    const int64_t eo = (int64_t)dy * wo + dx;
                    if constexpr (decltype(FULL)::value) resv[e] = rp[eo * ldr];
                    else resv[e] = (oyb + dy < ho && oxb + dx < wo) ? rp[eo * ldr] : 0.0f;
                });
            static_for<8>([&](auto RR) {
                constexpr int rr = decltype(RR)::value, r = keep0 + rr;
                const f32x4n got = lds_read4(xr + (unsigned)rr * 1024u);     // the partner's rows of this tile: (row0 b0, row0 b1, row1 b0, row1 b1)
                float sj[4][2];
                const float o00 = (acc[0][r] + acc[1][r]) + acc[2][r], o01 = (acc[1][r] - acc[2][r]) - acc[3][r];
                const float o10 = (acc[4][r] + acc[5][r]) + acc[6][r], o11 = (acc[5][r] - acc[6][r]) - acc[7][r];
                if constexpr (kfh == 0) {
                    sj[0][0] = o00; sj[0][1] = o01; sj[1][0] = o10; sj[1][1] = o11;
                    sj[2][0] = got.x; sj[2][1] = got.y; sj[3][0] = got.z; sj[3][1] = got.w;
                } else {
                    sj[0][0] = got.x; sj[0][1] = got.y; sj[1][0] = got.z; sj[1][1] = got.w;
                    sj[2][0] = o00; sj[2][1] = o01; sj[3][0] = o10; sj[3][1] = o11;
                }
                static_for<4>([&](auto P) {
                    constexpr int aa = decltype(P)::value >> 1, b = decltype(P)::value & 1;
                    constexpr int dy = GEO ? 2 * (rr >> 2) + aa : aa, dx = GEO ? 2 * (rr & 3) + b : 2 * ((rr & 3) + 8 * ((rr >> 2) & 1)) + b;
                    float v = (aa == 0 ? (sj[0][b] + sj[1][b]) + sj[2][b] : (sj[1][b] - sj[2][b]) - sj[3][b]) + bias;
                    auto finish = [&]() {
                        const int64_t eo = (int64_t)dy * wo + dx;
                        if constexpr (has_res) { if (a.res_mode == 1) v += resv[4 * rr + 2 * aa + b]; }
                        v = apply_act(v, act_c >= 0 ? act_c : a.act, slope);
                        if constexpr (has_res) { if (a.res_mode == 2) v += resv[4 * rr + 2 * aa + b]; }
                        ob[eo * ldo] = v;
                    };
                    if constexpr (decltype(FULL)::value) finish();
                    else if (oyb + dy < ho && oxb + dx < wo) finish();
                });
            });
        };
        auto epilogue_act = [&](auto FULL, auto RES) {
            switch (a.act) {
                case CSM_ACT_NONE: epilogue(ic<CSM_ACT_NONE>{}, FULL, RES); break;
                case CSM_ACT_RELU: epilogue(ic<CSM_ACT_RELU>{}, FULL, RES); break;
                case CSM_ACT_SILU: epilogue(ic<CSM_ACT_SILU>{}, FULL, RES); break;
                case CSM_ACT_PRELU: epilogue(ic<CSM_ACT_PRELU>{}, FULL, RES); break;
                default: epilogue(ic<-1>{}, FULL, RES); break;
            }
        };
        if (a.res_mode) { if (full) epilogue_act(std::true_type{}, std::true_type{}); else epilogue_act(std::false_type{}, std::true_type{}); }
        else if (full) epilogue_act(std::true_type{}, std::false_type{});
        else epilogue_act(std::false_type{}, std::false_type{});
    };
    if (fh == 0) body(ic<0>{}); else body(ic<1>{});
}

}  // namespace

namespace csmconv {

bool wino_eligible(const ConvArgs &a) {
    const int64_t bytes_in = (((int64_t)a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4;           // ONE sample (a launch takes as many samples as fit 2 GiB)
    const int64_t bytes_w = (int64_t)(a.cout_g / 64) * a.ncb * 4 * 32768;
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dil == 1 && a.pad == 1 && a.groups == 1 && a.ksplit <= 1 && (a.cin_g & 31) == 0 &&
           (a.cout_g & 63) == 0 && bytes_in < (1ll << 31) && bytes_w < (1ll << 31) && !(a.in.ld & 3) && !(((uintptr_t)a.in.p | (uintptr_t)a.w) & 15) &&
           a.out.h == a.in.h && a.out.w == a.in.w;
}

static int launch_conv_wino_chunk(const ConvArgs &a0, hipStream_t st);

// The kernels address the activations through a 32-bit buffer descriptor (range-checked LDS-DMA): a launch covers as many SAMPLES as fit
// 2 GiB of input view; larger batches are split by sample (independent work: the same bits whatever the split).  CSM_WINO_MAX_BYTES lowers
// the limit (tests).
int launch_conv_wino(const ConvArgs &a0, hipStream_t st) {
    const char *le = getenv("CSM_WINO_MAX_BYTES");
    const long long lv = le ? atoll(le) : 0;
    const int64_t limit = lv > 0 ? (int64_t)lv : (int64_t)((1ll << 31) - 1);
    const int64_t per_sample = (int64_t)a0.in.h * a0.in.w * a0.in.ld * 4;
    int chunk = (int)(limit / (per_sample > 0 ? per_sample : 1));
    if (chunk < 1) chunk = 1;
    if (chunk >= a0.in.n) return launch_conv_wino_chunk(a0, st);
    for (int n0 = 0; n0 < a0.in.n; n0 += chunk) {
        ConvArgs a = a0;
        const int nn = a0.in.n - n0 < chunk ? a0.in.n - n0 : chunk;
        a.in.n = a.out.n = nn; a.in.p = a0.in.p + (int64_t)n0 * a0.in.h * a0.in.w * a0.in.ld;
        a.out.p = a0.out.p + (int64_t)n0 * a0.out.h * a0.out.w * a0.out.ld;
        if (a0.res_mode) { a.res.n = nn; a.res.p = a0.res.p + (int64_t)n0 * a0.res.h * a0.res.w * a0.res.ld; }
        a.M = nn * a.out.h * a.out.w;
        const int rc = launch_conv_wino_chunk(a, st);
        if (rc) return rc;
    }
    return CSM_OK;
}

static int launch_conv_wino_chunk(const ConvArgs &a0, hipStream_t st) {
    constexpr int WM = 2, WN = 2;
    constexpr size_t lds = (size_t)2 * ((2 * (4 * WM + 2) * 18 + 7) / 8) * 1024 + (size_t)2 * 32768;       // (360 patch entries in either geometry)
    ConvArgs a = a0;
    // block-tile geometry of the eight-wave kernel (speed only): 32 x 8 output pixels, or 16 x 16 where that covers the map with fewer tiles
    static const int geo_force = [] { const char *e = getenv("CSM_WINO_GEO"); return e ? atoi(e) : -1; }();
    const int tx0 = (a.out.w + 31) / 32, ty0 = (a.out.h + 7) / 8, tx1 = (a.out.w + 15) / 16, ty1 = (a.out.h + 15) / 16;
    static const int waves = [] { const char *e = getenv("CSM_WINO_WAVES"); return e && atoi(e) == 4 ? 4 : 8; }();
    const int geo = waves != 8 ? 0 : (geo_force >= 0 ? (geo_force ? 1 : 0) : (tx1 * ty1 < tx0 * ty0 ? 1 : 0));
    const int tiles_x = geo ? tx1 : tx0, tiles_y = geo ? ty1 : ty0;
    a.m_tiles = tiles_x * tiles_y * a.out.n;
    dim3 grid(a.m_tiles, a.cout_g / 64, 1);
#ifdef CSM_WINO_DEV        // development build: CSM_WINO_VARIANT selects an ablation / option instantiation (tools/gpu/r05b.sh)
    const char *ve = getenv("CSM_WINO_VARIANT");
    int variant = ve ? atoi(ve) : 0;
    auto go = [&](auto kern) {
        static KernelPrep prep;
        (void)prep.ensure([&] { return prepare_kernel(kern, 64 * WM * WN, lds); });
        kern<<<grid, 64 * WM * WN, lds, st>>>(a, tiles_x, tiles_y);
        return csm::check_launch("k_conv_wino");
    };
    auto go8 = [&](auto kern) {
        static KernelPrep prep;
        (void)prep.ensure([&] { return prepare_kernel(kern, 512, lds); });
        kern<<<grid, 512, lds, st>>>(a, tiles_x, tiles_y);
        return csm::check_launch("k_conv_wino8");
    };
    if (geo && variant == 0) variant = 300;
    switch (variant) {
        case 300: return go8(&k_conv_wino8<0, 1>);
        case 200: return go8(&k_conv_wino8<0>);
        case 201: return go8(&k_conv_wino8<1>);
        case 202: return go8(&k_conv_wino8<2>);
        case 204: return go8(&k_conv_wino8<4>);
        case 208: return go8(&k_conv_wino8<8>);
        case 215: return go8(&k_conv_wino8<15>);
        case 1: return go(&k_conv_wino<WM, WN, 1, 0>);
        case 2: return go(&k_conv_wino<WM, WN, 2, 0>);
        case 3: return go(&k_conv_wino<WM, WN, 3, 0>);
        case 4: return go(&k_conv_wino<WM, WN, 4, 0>);
        case 8: return go(&k_conv_wino<WM, WN, 8, 0>);
        case 12: return go(&k_conv_wino<WM, WN, 12, 0>);
        case 15: return go(&k_conv_wino<WM, WN, 15, 0>);
        case 100: return go(&k_conv_wino<WM, WN, 0, 1>);
        default: break;
    }
#endif
    // two executions of the same arithmetic (same bits): the eight-wave form (two waves per SIMD) is the default; CSM_WINO_WAVES=4 selects
    // the one-wave-per-SIMD form (A/B measurements, tests)
    if (waves == 8 && geo) {
        static KernelPrep prep8g;
        (void)prep8g.ensure([&] { return prepare_kernel(&k_conv_wino8<0, 1>, 512, lds); });
        k_conv_wino8<0, 1><<<grid, 512, lds, st>>>(a, tiles_x, tiles_y);
        return csm::check_launch("k_conv_wino8");
    }
    if (waves == 8) {
        static KernelPrep prep8;
        (void)prep8.ensure([&] { return prepare_kernel(&k_conv_wino8<0>, 512, lds); });
        k_conv_wino8<0><<<grid, 512, lds, st>>>(a, tiles_x, tiles_y);
        return csm::check_launch("k_conv_wino8");
    }
    static KernelPrep prep;
    (void)prep.ensure([&] { return prepare_kernel(&k_conv_wino<WM, WN>, 64 * WM * WN, lds); });
    k_conv_wino<WM, WN><<<grid, 64 * WM * WN, lds, st>>>(a, tiles_x, tiles_y);
    return csm::check_launch("k_conv_wino");
}

}  // namespace csmconv
