// wino4.hip -- exact-fp32 Winograd F(4x4, 3x3) convolution on the matrix pipe (gfx950), fused in ONE kernel: input transform B^T d B
// by rotating groups of waves through LDS, 36 frequency GEMMs on v_mfma_f32_32x32x2_f32, output transform A^T M A + bias + residual +
// activation in the epilogue.  Arithmetic = the "Winograd F(4x4) contract" of include/csm355.h, restated independently in
// oracle/nets_oracle.c::orc_conv_wino4; this kernel reproduces it bit for bit (every transform value is the contract's fp32 expression,
// every product sum ONE fmaf chain in the direct contract's channel order -- fp32 MFMA is bitwise an fmaf chain over k).
//
// Why: F(2x2, 3x3) (wino.hip) executes 64 multiplications per 4x4 output pixels and channel pair, F(4x4, 3x3) 36 -- the 3x3 layers that
// carry a third of a step's convolution time execute 0.56x the MFMA FLOPs again.
//
// Mapping (CDNA4-first; the F(2x2) kernel's in-register transform does not carry over -- 36 accumulators per (32 tiles x 32 channels)
// are 576 registers, and a 6x6 window transformed per wave would be computed twice and read from LDS 2.25x):
//   * block = 12 waves (three per SIMD, <= 168 registers), ONE block per CU: 32 Winograd tiles = TWO SQUARES of 4 x 4 tiles (16 x 16 output
//     pixels each, consecutive in the launch's list of squares -- they need not be neighbours: a 40 x 40 map is 4.5 blocks per sample
//     where 32 x 16-pixel rectangles need 6) x 64 output channels.  Wave (i, nh) owns frequency ROW i (six 32x32 accumulators = 96
//     registers) of the 32 tiles x output channels [32 nh, 32 nh + 32).
//   * step = 4 input channels = two MFMAs per frequency = 12 MFMA slots per wave; ONE barrier per step, and it waits for LDS traffic
//     only: every fetch is waited for (vmcnt) a step after it was sent, by the wave that sent it.
//   * raw input patches (two of (16 + 2)^2 pixels): LDS-DMA, 8 channels (= 2 steps) per stage, two stages.  A pixel entry is 32 B (two
//     16-byte granules = the channels of k-half 0 / 1) at  psi = 342 square + 19 row + colpos(col)  (columns de-interleaved by col mod 4),
//     granule slot ^ ((row >> 3) & 1):  the stride-4 windows of a tile row are consecutive entries, the odd row pitch alternates the tile
//     rows between the two halves of a 256-byte bank period, and a ds_read_b64 of one channel pair touches every bank at most twice
//     (the minimum for 8-byte reads of 16-byte granules); two consecutive DMA lanes fetch the 32 contiguous bytes of a pixel.  The fused
//     form's two global offsets per lane live in LDS (no register is free for them in the main loop).
//   * transform B^T d B: ONCE per block, through LDS, by the waves of channel half s & 1 during step s for step s + 2 (each wave
//     transforms every other step).  Unit = (tile, frequency row = the wave's own row, k-half): row pass over the six window columns
//     (four ds_read_b64 and three v_pk_fma_f32 per column -- the two channels of a lane are the two halves of a packed operation;
//     ONE code path for all six rows with wave-uniform coefficients), column pass (the contract's T6, packed), six ds_write_b64 into
//     V[f][k-half][tile][2].  The work is cut into twelve slices placed behind the wave's MFMAs; loads run two slices ahead.
//   * A fragments: one conflict-free ds_read_b64 per frequency and step; B fragments: the transformed weights are packed on the host so
//     that a wave's share of a step is three 1-KB LDS-DMA pieces into one of its TWO private 3-KB slots (no barrier: only this wave
//     reads them), sent two steps ahead; a lane picks its four values of two frequencies with one ds_read_b128.  The operands of step
//     s + 1 replace the dead ones of step s from slot 6 on, so the first MFMA of a step issues right behind the barrier.
//   * the main loop exists twice (one copy per channel half = per parity of the steps in which a wave transforms): inside a copy every
//     step is straight-line code, and the compiler's wait counts are exact (a branch per slice made every slice drain the LDS queue).
//   * epilogue: column pass of A^T M A in registers, two accumulator elements per packed operation; the six waves of a channel half
//     exchange the row-pass inputs through LDS in two rounds of four element pairs; unit of the row pass = (tile pair, output column);
//     stores and residual loads go through range-checked buffer descriptors (ragged block tiles need no divergent control flow).
//   * execution forms: the above is the FUSED form (RPB = 6).  Launches with fewer block tiles than CUs (single frames, 23^2-45^2 maps)
//     run ROW-SPLIT (RPB = 3 / 2 frequency rows per block, 2 RPB waves, 6 / RPB x the blocks): same main loop, the epilogue leaves the
//     column-passed s[i][b] in a scratch tensor and k_wino4_rowpass finishes -- the same expressions on the same values, same bits.
// Measured (profiles/r06_*): 1.3-1.5x k_conv_wino8 on the 128/256-channel layers at batch 8, 1.1-1.2x on the 32/64-channel ones; where
// the time goes (ablations): the matrix pipe alone 450 us of 700-770 at 160^2 256->256; fp32 MFMA shares its issue with the
// transform's VALU and the DMA issue (they add up instead of overlapping), the 12-wave barrier costs ~15 %, and the epilogue's
// output leaves as one synchronized burst per round (stores into a 256-KB window cost nothing).
#include "csm_conv.h"
#include <utility>
#include <cstdlib>

using namespace csmconv;

namespace {

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef float f32x2n __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x4n *lds_f4_ptr;
typedef __attribute__((address_space(3))) f32x2n *lds_f2_ptr;

__device__ __forceinline__ f32x4n lds_read4(unsigned byte_addr) { return *(lds_f4_ptr)(size_t)byte_addr; }
__device__ __forceinline__ f32x2n lds_read2(unsigned byte_addr) { return *(lds_f2_ptr)(size_t)byte_addr; }
__device__ __forceinline__ void lds_write4(unsigned byte_addr, f32x4n v) { *(lds_f4_ptr)(size_t)byte_addr = v; }
__device__ __forceinline__ void lds_write2(unsigned byte_addr, f32x2n v) { *(lds_f2_ptr)(size_t)byte_addr = v; }
typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u32x2n *lds_u2_ptr;
__device__ __forceinline__ u32x2n lds_read2u(unsigned byte_addr) { return *(lds_u2_ptr)(size_t)byte_addr; }
__device__ __forceinline__ void lds_write2u(unsigned byte_addr, u32x2n v) { *(lds_u2_ptr)(size_t)byte_addr = v; }

// LDS-DMA piece with a scalar byte offset on the global side: 64 lanes x 16 B -> LDS [lds_byte_addr, +1 KB)
__device__ __forceinline__ void dma16s(unsigned voff, i32x4 rsrc, unsigned soff, unsigned lds_byte_addr) {
    unsigned keep;
    soff = __builtin_amdgcn_readfirstlane(soff);              // (wave-uniform by construction; a literal constant is not a valid soffset operand)
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_byte_addr) : "memory");
}

template <int N> using ic = std::integral_constant<int, N>;
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// raw barrier with its waits in ONE asm block (tools/check_isa_barriers.py)
template <int VM> __device__ __forceinline__ void wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(VM) : "memory");
}

// packed fp32 on the transform's channel pairs: v_pk_fma_f32 / v_pk_add_f32 are the same IEEE operations per component at half the VALU
// issue (the compiler splits __builtin_elementwise_fma back into two v_fma_f32 here, hence the asm).  The coefficient is a wave-uniform
// scalar: an SGPR pair whose LOW half feeds both components (op_sel_hi:[0,1,1])
__device__ __forceinline__ f32x2n pk_fma(f32x2n coef, f32x2n b, f32x2n c) {
    f32x2n d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "s"(coef), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2n pk_add(f32x2n a, f32x2n b) {
    f32x2n d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2n pk_sub(f32x2n a, f32x2n b) {      // a + (-b): exact negation
    f32x2n d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// the contract's 1-D transforms (include/csm355.h), one component
struct T6 { float t0, t1, t2, t3, t4, t5; };
__device__ __forceinline__ T6 bt6(float d0, float d1, float d2, float d3, float d4, float d5) {
    T6 o;
    o.t0 = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
    o.t5 = fmaf(4.0f, d1, fmaf(-5.0f, d3, d5));
    const float a = fmaf(-4.0f, d2, d4), b = fmaf(-4.0f, d1, d3);
    o.t1 = a + b; o.t2 = a - b;
    const float c = d4 - d2, e = d3 - d1;
    o.t3 = fmaf(2.0f, e, c); o.t4 = fmaf(-2.0f, e, c);
    return o;
}
__device__ __forceinline__ f32x4n at6(float m0, float m1, float m2, float m3, float m4, float m5) {
    const float p = m1 + m2, q = m1 - m2, r = m3 + m4, t = m3 - m4;
    f32x4n s;
    s.x = (m0 + p) + r;
    s.y = fmaf(2.0f, t, q);
    s.z = fmaf(4.0f, r, p);
    s.w = fmaf(8.0f, t, q) + m5;
    return s;
}

// A block's 32 Winograd tiles are TWO SQUARES of 4 x 4 tiles (16 x 16 output pixels each), consecutive in the launch's list of squares
// (sample, square row, square column): the squares of a block need not be neighbours -- a 40 x 40 map is 9 squares = 4.5 blocks per sample
// where 32 x 16-pixel rectangles need 6, an 80 x 80 map 12.5 instead of 15.
constexpr int kSQ = 16;                                      // output pixels of a square's side
constexpr int kPS = kSQ + 2;                                 // 18 x 18 patch pixels per square
constexpr int kPitch = 19;                                   // pixel entries per patch row (18 + 1: an ODD pitch spreads the tile rows over the banks)
constexpr int kSqEnt = kPS * kPitch;                         // 342 entries per square
constexpr int kPsi = 2 * kSqEnt;                             // 684 pixel entries (32 B = 8 channels each) of a raw stage
constexpr int kRawPieces = (kPsi * 2 + 63) / 64;             // 22 DMA pieces
constexpr unsigned kRawB = kRawPieces * 1024u;               // 21 504 B per stage
constexpr unsigned kV0 = 2u * kRawB;                         // 43 008
constexpr unsigned kUW = 3072u;                              // a wave's U slot (three pieces); two slots per wave
constexpr unsigned kOob = 0x80000000u;
// LDS map of a block that owns RPB of the six frequency rows (2 RPB waves): [raw 0][raw 1][V 0][V 1][U: 2 RPB waves x 2 slots][raw loader's offsets]
//   RPB = 6: the fused form (12 waves, one block per CU, 156 KB);  RPB = 1 / 2 / 3: the row-split forms (2 / 4 / 6 waves, 61 / 79 / 98 KB)
constexpr unsigned vb_of(int rpb) { return (unsigned)rpb * 6u * 512u; }                         // V[row][j][k-half][32 tiles][2] of one step
constexpr unsigned u0_of(int rpb) { return kV0 + 2u * vb_of(rpb); }
constexpr unsigned lds_of(int rpb) { return u0_of(rpb) + (unsigned)(2 * rpb) * 2u * kUW; }
constexpr int ppw_of(int rpb) { return (kRawPieces + 2 * rpb - 1) / (2 * rpb); }                // raw pieces per wave and stage
constexpr int ppe_of(int rpb) { return (ppw_of(rpb) + 1) & ~1; }                                // ... rounded up to pairs (read as 8-byte words)
// (the fused form has no register left for the loader's offsets and parks them in LDS; the row-split forms keep them in registers --
// two blocks of RPB = 2 then fit a CU)
constexpr unsigned lds_all_of(int rpb) { return lds_of(rpb) + (rpb == 6 ? (unsigned)(2 * rpb) * 64u * (unsigned)ppe_of(rpb) * 4u : 0u); }
static_assert(kPS == 18 && lds_all_of(6) <= 160u * 1024u, "tile shape");

// Pixel entry of patch position (R, C) of a square: psi = R * 19 + colpos(C), colpos = the columns de-interleaved by C mod 4 (0..4 | 5..9 |
// 10..13 | 14..17), so that the stride-4 windows of a tile row are consecutive entries.  psi_k(r, c): the compile-time part for window
// position (r, c) of tile (ty, tx): psi = 19 (4 ty + r) + colbase(c & 3) + tx + (c >> 2).
__host__ __device__ constexpr int colbase(int cr) { return cr == 0 ? 0 : (cr == 1 ? 5 : (cr == 2 ? 10 : 14)); }
__host__ __device__ constexpr int psi_k(int r, int c) { return r * kPitch + colbase(c & 3) + (c >> 2); }

// ABL (development ablations, timing only -- results invalid): 1 = no DMA, 2 = no transform, 4 = no MFMA, 8 = no epilogue stores,
// 16 = no barrier in the main loop, 32 = no operand reads (A / B fragments) in the main loop, 64 = every block stores into the first
// 256 KB of the output (same instructions, no HBM write traffic: is the epilogue's time arithmetic or the write burst?)
// RPB: frequency rows per block.  6 = the FUSED form (a block owns all 36 frequencies of its tile and finishes the outputs itself).
// 1 / 2 / 3 = the ROW-SPLIT forms for launches with fewer block tiles than CUs (single frames, small maps): gridDim.y = cout tiles x 6 / RPB,
// a block owns RPB rows (2 RPB waves, several blocks per CU), does the column pass of A^T M A on them and writes s[i][b] (four values per
// row, tile and channel) to a scratch buffer; k_wino4_rowpass finishes.  Same expressions on the same values: the forms give the same bits.
template <int ABL = 0, int RPB = 6>
__global__ __launch_bounds__(128 * RPB, RPB == 1 ? 1 : (RPB == 6 ? 3 : 2)) void k_conv_wino4(ConvArgs a, int sx_n, int sy_n) {
    constexpr int NW = 2 * RPB, PPW = ppw_of(RPB), PPE = ppe_of(RPB);
    constexpr unsigned kVB = vb_of(RPB), kU0 = u0_of(RPB), kLds = lds_of(RPB);
    extern __shared__ __attribute__((aligned(64))) float lds[];   // [raw 0][raw 1][V 0][V 1][U: NW waves x 2 slots][offsets]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nh = wave / RPB, rl = wave - RPB * nh;          // channel half, frequency row inside the block
    const int li = lane & 31, lh = lane >> 5;
    int mt, ntile, zz;
    block_to_tile(mt, ntile, zz, 1);                          // cout tile slowest: the blocks resident on an XCD share one 64-channel U panel
    const int rg = RPB == 6 ? 0 : ntile % (6 / RPB);          // row group of a row-split block
    if constexpr (RPB != 6) ntile /= (6 / RPB);
    const int wi = rg * RPB + rl;                             // frequency row
    const int ho = a.out.h, wo = a.out.w;
    const int nsteps = a.cin_g >> 2, nstages = nsteps >> 1;   // a raw stage = 8 channels = 2 steps
    // the block's two squares: sample, origin, validity (a launch with an odd number of squares leaves the last block's second one empty)
    int sqn[2], sqy[2], sqx[2];
    bool sqv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sq = 2 * mt + q, per = sx_n * sy_n;
        sqv[q] = sq < a.out.n * per;
        sqn[q] = sq / per;
        const int rem = sq - sqn[q] * per;
        sqy[q] = (rem / sx_n) * kSQ; sqx[q] = (rem % sx_n) * kSQ;
    }

    i32x4 ra, rb;
    {
        uint64_t pa = (uint64_t)a.in.p, pb = (uint64_t)a.w;
        unsigned na = (unsigned)((((int64_t)a.in.n * a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4);
        unsigned nb = (unsigned)((int64_t)(a.cout_g / 64) * nsteps * (12 * kUW));
        auto sg = [](unsigned v) { return (int)__builtin_amdgcn_readfirstlane(v); };      // (pins the descriptors to SGPRs)
        ra = i32x4{sg((unsigned)pa), sg((unsigned)(pa >> 32)), sg(na), 0x00020000};
        rb = i32x4{sg((unsigned)pb), sg((unsigned)(pb >> 32)), sg(nb), 0x00020000};
    }
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    // ---- this wave's U stream: step s at ((ntile nsteps + s) 12 + wave) 3 KB -> slot s & 1 ----
    const unsigned voffU = (unsigned)lane * 16u;
    unsigned u_src = (unsigned)((ntile * nsteps) * 12 + 6 * nh + wi) * kUW;  // next step to fetch
    const unsigned ldsU = lds0 + kU0 + (unsigned)wave * (2u * kUW);
    const unsigned blane = ldsU + voffU;
#pragma unroll
    for (int p = 0; p < 3; ++p) {                             // B(0), B(1): scalar addresses, out before anything else is computed
        if constexpr (!(ABL & 1)) {
            dma16s(voffU, rb, u_src + (unsigned)p * 1024u, ldsU + (unsigned)p * 1024u);
            dma16s(nsteps > 1 ? voffU : kOob, rb, u_src + 12u * kUW + (unsigned)p * 1024u, ldsU + kUW + (unsigned)p * 1024u);
        }
    }
    u_src += 24u * kUW;

    // ---- raw patch loader: wave w owns pieces w, w + NW, ... of a stage.  Granule g = 2 psi + s' sits at LDS position g; it holds channels
    // [4 slot, 4 slot + 4) of the stage's eight, slot = s' ^ ((row >> 2) & 1) ----
    // (fused form: a lane's global offsets live in LDS behind the U slots -- no register is free for them in the main loop, and a scratch
    // reload would put a vmcnt(0) in front of every raw fetch; the row-split forms keep them in registers)
    const unsigned offL = lds0 + kLds + (unsigned)tid * (unsigned)(PPE * 4);
    auto raw_lds = [&](int stage, int q) __attribute__((always_inline)) {
        int pp = wave + NW * q;
        if (pp > kRawPieces - 1) pp = kRawPieces - 1;         // (a piece past the end repeats the last one: same bytes, same place)
        return lds0 + (unsigned)(stage & 1) * kRawB + (unsigned)pp * 1024u;
    };
    unsigned offP[PPE];
    {
#pragma unroll
        for (int q = 0; q < PPE; ++q) {
            int pp = wave + NW * q;
            if (pp > kRawPieces - 1) pp = kRawPieces - 1;
            const int g = 64 * pp + lane;
            const int psi = g >> 1, sl = g & 1;
            const int sq = psi >= kSqEnt ? 1 : 0, rem = psi - sq * kSqEnt, R = rem / kPitch, cp = rem - R * kPitch;
            const int cr = cp < 5 ? 0 : (cp < 10 ? 1 : (cp < 14 ? 2 : 3)), C = 4 * (cp - colbase(cr)) + cr;
            const int slot = sl ^ ((R >> 3) & 1);
            const int iy = (sq ? sqy[1] : sqy[0]) - 1 + R, ix = (sq ? sqx[1] : sqx[0]) - 1 + C;
            const bool v = psi < kPsi && cp < kPS && (sq ? sqv[1] : sqv[0]) && iy >= 0 && iy < a.in.h && ix >= 0 && ix < a.in.w;
            offP[q] = v ? (unsigned)((((sq ? sqn[1] : sqn[0]) * a.in.h + iy) * a.in.w + ix) * a.in.ld + slot * 4) * 4u : kOob;
        }
        if constexpr (RPB == 6) {
#pragma unroll
            for (int q = 0; q < PPE; q += 2) lds_write2u(offL + (unsigned)q * 4u, u32x2n{offP[q], offP[q + 1]});
        }
        if constexpr (!(ABL & 1)) {
#pragma unroll
            for (int q = 0; q < PPW; ++q) dma16s(offP[q], ra, 0u, raw_lds(0, q));
#pragma unroll
            for (int q = 0; q < PPW; ++q) dma16s(nstages > 1 ? offP[q] : kOob, ra, 32u, raw_lds(1, q));
        }
    }

    // ---- fragment / unit addresses ----
    const int tx = li & 3, ty = (li >> 2) & 3, tq = li >> 4;                               // tile li = square tq, tile row ty, tile column tx
    const unsigned ubase = lds0 + (unsigned)(tq * kSqEnt + ty * 4 * kPitch + tx) * 32u + (unsigned)nh * 8u;      // (this wave transforms steps sn = nh mod 2: half nh)
    // granule slot ^ ((R >> 3) & 1) with R = 4 ty + r: window rows r < 4 see (ty >> 1) & 1, rows 4, 5 see ((ty + 1) >> 1) & 1
    const unsigned bxA = ubase + (unsigned)((lh ^ ((ty >> 1) & 1)) << 4), bxB = ubase + (unsigned)((lh ^ (((ty + 1) >> 1) & 1)) << 4);
    const unsigned vlane = lds0 + kV0 + (unsigned)lh * 256u + (unsigned)li * 8u + (unsigned)(rl * 6) * 512u;     // + parity kVB + j 512
    const unsigned vdst = vlane + (unsigned)nh * kVB;
    // transform unit of this lane: frequency ROW wi of tile li, k-half lh, two channels.  ONE code path for all six rows:
    //   t = fmaf(g, fmaf(c1, dP, dQ), fmaf(c2, dR, dS))      (wave-uniform coefficients, four window rows)
    // rows 1..4: (P, Q, R, S) = (1, 3, 2, 4), c1 = c2 = del, (g, del) = (1, -4), (-1, -4), (2, -1), (-2, -1) -- bit for bit the contract's
    //   a + b, a - b, fmaf(2, e, c), fmaf(-2, e, c): fmaf(+-1, x, y) IS y +- x, fmaf(-1, d2, d4) IS d4 - d2;
    // rows 0 / 5:  g = 4, c1 = 0 with P = Q = 0 / 1 (fmaf(0, d, d) IS d for every finite d, signed zeros included), c2 = -5,
    //   (R, S) = (2, 4) / (3, 5): the contract's fmaf(4, d0, fmaf(-5, d2, d4)) / fmaf(4, d1, fmaf(-5, d3, d5)).
    const bool rowB = wi == 0 || wi == 5;
    const float cg = rowB ? 4.0f : (wi == 1 ? 1.0f : (wi == 2 ? -1.0f : (wi == 3 ? 2.0f : -2.0f)));
    const float c1 = rowB ? 0.0f : (wi <= 2 ? -4.0f : -1.0f), c2 = rowB ? -5.0f : c1;
    unsigned rowb[4];                                          // the unit's window rows P, Q, R, S in raw buffer 0
    {
        const int rP = rowB ? (wi == 0 ? 0 : 1) : 1, rQ = rowB ? rP : 3, rR = rowB ? rP + 2 : 2, rS = rR + 2;
        const int rr[4] = {rP, rQ, rR, rS};
#pragma unroll
        for (int k = 0; k < 4; ++k) rowb[k] = ((rr[k] >> 2) ? bxB : bxA) + (unsigned)(rr[k] * kPitch * 32);
    }

    auto sgf = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(unsigned, v))); };
    const f32x2n cg2 = {sgf(cg), 0.0f}, c12 = {sgf(c1), 0.0f}, c22 = {sgf(c2), 0.0f};      // (SGPR pairs: only the low half is read)
    const f32x2n k4 = {sgf(4.0f), 0.0f}, km5 = {sgf(-5.0f), 0.0f}, km4 = {sgf(-4.0f), 0.0f}, k2 = {sgf(2.0f), 0.0f}, km2 = {sgf(-2.0f), 0.0f};
    f32x16 acc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    f32x4n Bq[3];
    f32x2n A[6];
    f32x2n D[2][4], T[6];                                      // transform: two window columns in flight, the row-pass results

    // transform of a step out of raw buffer RB, cut into 12 slices (one per MFMA slot): slices 0..5 = row pass of window column c (and
    // the loads of column c + 2: a load has two slots to return), slices 6..11 = column pass (the contract's T6 on T[0..5], two channels)
    // + the six ds_write_b64 into V[nh]
    auto xf_load = [&](auto RB, auto C) __attribute__((always_inline)) {
        constexpr int c = decltype(C)::value;
        constexpr unsigned imm = (unsigned)decltype(RB)::value * kRawB + (unsigned)(psi_k(0, c) * 32);
#pragma unroll
        for (int k = 0; k < 4; ++k) D[c & 1][k] = lds_read2(rowb[k] + imm);
    };
    auto xf_begin = [&](auto RB) __attribute__((always_inline)) {
        if constexpr (!(ABL & 2)) { xf_load(RB, ic<0>{}); xf_load(RB, ic<1>{}); }
    };
    auto xf_slice = [&](auto RB, auto M) __attribute__((always_inline)) {
        constexpr int m = decltype(M)::value;
        if constexpr (ABL & 2) {
        } else if constexpr (m < 6) {
            constexpr int b = m & 1;
            // (two channels per lane: v_pk_fma_f32 -- the same IEEE fma per component, half the VALU issue)
            T[m] = pk_fma(cg2, pk_fma(c12, D[b][0], D[b][1]), pk_fma(c22, D[b][2], D[b][3]));
            if constexpr (m < 4) xf_load(RB, ic<m + 2>{});
        } else if constexpr (m == 6) {
            lds_write2(vdst + 0u * 512u, pk_fma(k4, T[0], pk_fma(km5, T[2], T[4])));
        } else if constexpr (m == 7) {
            lds_write2(vdst + 5u * 512u, pk_fma(k4, T[1], pk_fma(km5, T[3], T[5])));
        } else if constexpr (m == 8) {                          // a -> D[0][0], b -> D[0][1]
            D[0][0] = pk_fma(km4, T[2], T[4]);
            D[0][1] = pk_fma(km4, T[1], T[3]);
            lds_write2(vdst + 1u * 512u, pk_add(D[0][0], D[0][1]));
        } else if constexpr (m == 9) {
            lds_write2(vdst + 2u * 512u, pk_sub(D[0][0], D[0][1]));
        } else if constexpr (m == 10) {                         // c -> D[0][0], e -> D[0][1]
            D[0][0] = pk_sub(T[4], T[2]);
            D[0][1] = pk_sub(T[3], T[1]);
            lds_write2(vdst + 3u * 512u, pk_fma(k2, D[0][1], D[0][0]));
        } else {
            lds_write2(vdst + 4u * 512u, pk_fma(km2, D[0][1], D[0][0]));
        }
    };

    // ---- prologue: raw stage 0 -> V[0] (waves of channel half 0) and V[1] (half 1); B(0), A(0) into registers ----
    wait_barrier<0>();                                        // raw stages 0, 1 and B(0), B(1) have landed
    xf_begin(ic<0>{});
    static_for<12>([&](auto M) { xf_slice(ic<0>{}, M); });
#pragma unroll
    for (int p = 0; p < 3; ++p) Bq[p] = lds_read4(blane + (unsigned)p * 1024u);
    wait_barrier<0>();                                        // V[0], V[1] complete, B(0) read
#pragma unroll
    for (int j = 0; j < 6; ++j) A[j] = lds_read2(vlane + (unsigned)j * 512u);
    wait_barrier<0>();                                        // A(0) read by everybody: step 0 may overwrite V[0]

    // ---- main loop: step s = 4 input channels = 12 MFMA slots per wave.  Beside the MFMAs: slots 0-2 send B(s + 2) into U slot s & 1
    // (whose contents, B(s), are in registers since step s - 1); slots 1 .. PPW of the EVEN steps send this wave's pieces of raw stage
    // s / 2 + 2 (its buffer was last read in step s - 1 and is first read in step s + 2; the pieces are waited for in step s + 1 and
    // published by that step's barrier); the waves of channel half s & 1 transform the window of step s + 2 into V[s & 1] (whose
    // previous image, step s, is in everybody's registers since the barrier of step s - 1); from slot 6 on the operands of step s + 1
    // replace the dead ones -- V[(s + 1) & 1] was published by the barrier of step s - 1, B(s + 1) was sent a step ago (vmcnt leaves only
    // this step's fetches in flight).  The barrier itself waits for LDS traffic only ----
    // XF: this wave transforms in this step (ONE wave-uniform branch per step selects the variant: straight-line slots, so that the
    // compiler's wait counts stay exact -- a branch per slice made every slice drain the LDS queue)
    auto step = [&](auto Q, auto XF, auto ROLE, int s) __attribute__((always_inline)) {
        constexpr int q = decltype(Q)::value;
        constexpr bool xf = decltype(XF)::value;
        constexpr int u0 = 0;                                  // slots [u0, u0 + 3) send the three U pieces (issuing the two channel halves in different slots measured equal)
        constexpr int rbuf = ((q + 2) >> 1) & 1;               // raw buffer of step s + 2: stage (s + 2) >> 1
        unsigned offq0 = 0u, offq1 = 0u;
        if constexpr (xf) xf_begin(ic<rbuf>{});
        static_for<12>([&](auto M) {
            constexpr int m = decltype(M)::value, half = m / 6, t = (m % 6) / 3, j = 3 * half + m % 3;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 4)) {
                const f32x4n bf = Bq[j >> 1];
                const float bv = (j & 1) ? (t ? bf.w : bf.z) : (t ? bf.y : bf.x);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(t ? A[j].y : A[j].x, bv, acc[j], 0, 0, 0);
            }
            if constexpr (m >= u0 && m < u0 + 3) {
                if constexpr (!(ABL & 1)) dma16s(s + 2 < nsteps ? voffU : kOob, rb, u_src + (unsigned)(m - u0) * 1024u, ldsU + (unsigned)(q & 1) * kUW + (unsigned)(m - u0) * 1024u);
                if constexpr (m == u0 + 2) u_src += 12u * kUW;
            }
            // raw stage s / 2 + 2 (EVEN steps): piece k of this wave goes out in slot k + 1, its offset pair is read in slot k & ~1
            if constexpr ((q & 1) == 0 && m >= 1 && m <= PPW) {
                if constexpr (!(ABL & 1))
                    dma16s((s >> 1) + 2 < nstages ? (RPB != 6 ? offP[m - 1] : (((m - 1) & 1) ? offq1 : offq0)) : kOob, ra, (unsigned)((s >> 1) + 2) * 32u, raw_lds((s >> 1) + 2, m - 1));
            }
            if constexpr (RPB == 6 && (q & 1) == 0 && (m & 1) == 0 && m < PPW) {
                const u32x2n o2 = lds_read2u(offL + (unsigned)m * 4u);
                offq0 = o2.x; offq1 = o2.y;
            }
            // (slice k runs one slot late: the first window column's loads, sent at the top of the step, return under slot 0's MFMA)
            if constexpr (xf && m >= 1) xf_slice(ic<rbuf>{}, ic<m - 1>{});
            if constexpr (xf && m == 11) xf_slice(ic<rbuf>{}, ic<11>{});
            if constexpr (ABL & 32) {
            } else if constexpr (m == 6) {
                // everything sent before this step has landed: B(s + 1), and the raw pieces of step s - 1
                // (issued in this step before this point: the three U pieces and, in even steps, the raw pieces of slots 1..6)
                constexpr int mine = 3 + ((q & 1) == 0 ? (PPW < 6 ? PPW : 6) : 0);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(mine) : "memory");
                Bq[0] = lds_read4(blane + (unsigned)((q + 1) & 1) * kUW);
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) A[jj] = lds_read2(vlane + (unsigned)((q + 1) & 1) * kVB + (unsigned)jj * 512u);
            } else if constexpr (m == 10) {                     // (dead by now: B of j = 2, 3 since slot 9, A of j = 3, 4 since slots 9, 10)
                Bq[1] = lds_read4(blane + (unsigned)((q + 1) & 1) * kUW + 1024u);
#pragma unroll
                for (int jj = 3; jj < 5; ++jj) A[jj] = lds_read2(vlane + (unsigned)((q + 1) & 1) * kVB + (unsigned)jj * 512u);
            } else if constexpr (m == 11) {
                Bq[2] = lds_read4(blane + (unsigned)((q + 1) & 1) * kUW + 2048u);
                A[5] = lds_read2(vlane + (unsigned)((q + 1) & 1) * kVB + 5u * 512u);
            }
        });
        __builtin_amdgcn_sched_barrier(0);
        // everybody: V[s + 2] written, the operands of step s + 1 read, the raw pieces waited for in slot 6 landed
        if constexpr (ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // the whole loop exists twice, once per channel half (= the parity of the steps in which the wave transforms): inside a copy every
    // step is straight-line code.  The last four steps are peeled: their steps 2 and 3 have no step s + 2 to prepare
    auto run = [&](auto ROLE) __attribute__((always_inline)) {
        constexpr int role = decltype(ROLE)::value;
        int s4 = 0;
        for (; s4 + 4 < nsteps; s4 += 4)
            static_for<4>([&](auto Q) { step(Q, std::integral_constant<bool, (decltype(Q)::value & 1) == role>{}, ROLE, s4 + decltype(Q)::value); });
        static_for<4>([&](auto Q) {
            step(Q, std::integral_constant<bool, (decltype(Q)::value & 1) == role && decltype(Q)::value < 2>{}, ROLE, s4 + decltype(Q)::value);
        });
    };
    if (nh == 0) run(ic<0>{}); else run(ic<1>{});
    wait_barrier<0>();        // the trailing (dead) fetches have landed for every wave: raw / V / U become the exchange area

    // ---- epilogue.  Everything it needs is derived from `lane_e`, which the compiler cannot see through: the bias / slope loads and the
    // output addresses would otherwise be hoisted above the main loop, where every register is taken ----
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e) :: "memory");
    const int le_i = lane_e & 31, le_h = lane_e >> 5;
    const int co = ntile * 64 + 32 * nh + le_i;
    const float bias = a.bias ? a.bias[co] : 0.0f;
    const float slope = a.slope ? a.slope[co] : 0.0f;
    if constexpr (RPB != 6) {
        // ---- row-split form: column pass (over j) of this wave's frequency row, s[0..3] per (tile, channel) -> scratch[tile][row][channel][4]
        // (tiles numbered (sample, tile row, tile column) over the whole map; k_wino4_rowpass finishes) ----
        const int tx_n = (wo + 3) >> 2, ty_n = (ho + 3) >> 2;
        float *sp = a.partial + ((int64_t)wi * a.cout_g + co) * 4;
        static_for<16>([&](auto RR) {
            constexpr int r = decltype(RR)::value;
            const f32x4n sv = at6(acc[0][r], acc[1][r], acc[2][r], acc[3][r], acc[4][r], acc[5][r]);
            constexpr int q = r >> 3;                          // element r: square r >> 3, tile row 2 ((r >> 2) & 1) + lh, tile column r & 3
            const int txg = (sqx[q] >> 2) + (r & 3), tyg = (sqy[q] >> 2) + 2 * ((r >> 2) & 1) + le_h;
            if (sqv[q] && txg < tx_n && tyg < ty_n)
                *reinterpret_cast<f32x4n *>(sp + (((int64_t)sqn[q] * ty_n + tyg) * tx_n + txg) * (int64_t)(24 * a.cout_g)) = sv;
        });
        return;
    }
    const int ldo = a.out.ld, ldr = a.res.ld;
    const unsigned xw = lds0 + (unsigned)wave * 8192u + (unsigned)lane_e * 8u;
    const unsigned xr = lds0 + (unsigned)(6 * nh) * 8192u + (unsigned)lane_e * 8u;
    // Stores and residual loads go through range-checked buffer descriptors of the SAMPLE's output / residual view (round h of the
    // exchange = square h of the block): a pixel outside the map gets the offset 2^31 and is dropped (loads return 0) by the hardware --
    // no divergent control flow on ragged squares
    const f32x2n k8 = {sgf(8.0f), 0.0f};
    // the contract's output transform O6 on PAIRS (v_pk_*: the same IEEE operations per component): components = two accumulator
    // elements (column pass) / two tiles (row pass)
    auto at6p = [&](f32x2n m0, f32x2n m1, f32x2n m2, f32x2n m3, f32x2n m4, f32x2n m5, f32x2n (&o)[4]) __attribute__((always_inline)) {
        const f32x2n p = pk_add(m1, m2), q = pk_sub(m1, m2), r = pk_add(m3, m4), t = pk_sub(m3, m4);
        o[0] = pk_add(pk_add(m0, p), r);
        o[1] = pk_fma(k2, t, q);
        o[2] = pk_fma(k4, r, p);
        o[3] = pk_add(pk_fma(k8, t, q), m5);
    };
    // exchange area of a round: [wave][element pair p of the round's four][b][lane][2] (8 KB per wave).  A unit of the row pass = (pair p,
    // output column b): six ds_read_b64, one packed O6 over the frequency rows, eight outputs (two tiles x four output rows a).
    // ACT: compile-time activation (-1 = the run-time switch of apply_act); FULL: interior block; RES: a residual is added
    auto finish = [&](auto ACT, auto FULL, auto RES, int h, int u) __attribute__((always_inline)) {
        constexpr int act_c = decltype(ACT)::value;
        constexpr bool has_res = decltype(RES)::value, is_full = decltype(FULL)::value;
        const int pr = u >> 2, b = u & 3;
        f32x2n S[6], Y[4];
#pragma unroll
        for (int i = 0; i < 6; ++i) S[i] = lds_read2(xr + (unsigned)i * 8192u + (unsigned)u * 512u);
        // elements r0 = 8 h + 2 pr and r0 + 1: square h, tile row 2 ((r0 >> 2) & 1) + lh, tile columns (r0 & 3) + e
        const int r0 = 8 * h + 2 * pr, sn = h ? sqn[1] : sqn[0], sy = h ? sqy[1] : sqy[0], sx = h ? sqx[1] : sqx[0];
        const bool sv = h ? sqv[1] : sqv[0];
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(a.out.p + (int64_t)sn * ho * wo * ldo, 0,
                                                                             sv ? (int)((((int64_t)ho * wo - 1) * ldo + a.out.c) * 4) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr_ = __builtin_amdgcn_make_buffer_rsrc(has_res ? a.res.p + (int64_t)sn * ho * wo * ldr : a.out.p, 0,
                                                                              (has_res && sv) ? (int)((((int64_t)ho * wo - 1) * ldr + a.res.c) * 4) : 0, 0x00020000);
        const int oyb = sy + 4 * (2 * ((r0 >> 2) & 1) + le_h), oxb = sx + 4 * (r0 & 3) + b;
        const unsigned pix = (unsigned)(oyb * wo + oxb);
        unsigned vb = (pix * (unsigned)ldo + (unsigned)co) * 4u;
        const unsigned vbr = (pix * (unsigned)ldr + (unsigned)co) * 4u;
        if constexpr (ABL & 64) vb &= 0x3ffffu;
        at6p(S[0], S[1], S[2], S[3], S[4], S[5], Y);
        static_for<8>([&](auto E) {
            constexpr int e = decltype(E)::value & 1, aa = decltype(E)::value >> 1;
            float v = (e ? Y[aa].y : Y[aa].x) + bias;
            const bool ok = is_full || (oyb + aa < ho && oxb + 4 * e < wo);      // (an empty square's descriptor has no records: everything is dropped)
            const int so = (ABL & 64) ? 0 : (aa * wo + 4 * e) * ldo * 4;
            if constexpr (has_res) {
                const float rv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr_, ok ? (int)vbr : (int)kOob, (aa * wo + 4 * e) * ldr * 4, 0));
                if (a.res_mode == 1) v += rv;
                v = apply_act(v, act_c >= 0 ? act_c : a.act, slope);
                if (a.res_mode == 2) v += rv;
            } else v = apply_act(v, act_c >= 0 ? act_c : a.act, slope);
            if constexpr (!(ABL & 8)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ro, ok ? (int)vb : (int)kOob, so, 0);
        });
    };
    auto finish_act = [&](auto FULL, auto RES, int h, int u) __attribute__((always_inline)) {
        switch (a.act) {                                       // uniform: the common activations get a body without the per-element switch
            case CSM_ACT_NONE: finish(ic<CSM_ACT_NONE>{}, FULL, RES, h, u); break;
            case CSM_ACT_RELU: finish(ic<CSM_ACT_RELU>{}, FULL, RES, h, u); break;
            case CSM_ACT_SILU: finish(ic<CSM_ACT_SILU>{}, FULL, RES, h, u); break;
            default: finish(ic<-1>{}, FULL, RES, h, u); break;
        }
    };
    // column pass (over j) of this wave's frequency row for all 16 accumulator elements FIRST (the 32x32 accumulators are register tuples:
    // they stay allocated as long as any element is live), two elements per packed operation; round 0's s values go to LDS, round 1's
    // wait in registers
    f32x2n sv[8][4];                                           // [element pair][b]
    static_for<8>([&](auto PP) {
        constexpr int r = 2 * decltype(PP)::value;
        at6p(__builtin_shufflevector(acc[0], acc[0], r, r + 1), __builtin_shufflevector(acc[1], acc[1], r, r + 1),
             __builtin_shufflevector(acc[2], acc[2], r, r + 1), __builtin_shufflevector(acc[3], acc[3], r, r + 1),
             __builtin_shufflevector(acc[4], acc[4], r, r + 1), __builtin_shufflevector(acc[5], acc[5], r, r + 1), sv[decltype(PP)::value]);
        if constexpr (decltype(PP)::value < 4) {
#pragma unroll
            for (int b = 0; b < 4; ++b) lds_write2(xw + (unsigned)(decltype(PP)::value * 4 + b) * 512u, sv[decltype(PP)::value][b]);
        }
    });
    static_for<2>([&](auto H) {
        constexpr int h = decltype(H)::value;
        if constexpr (h == 1) {
            static_for<4>([&](auto PP) {
#pragma unroll
                for (int b = 0; b < 4; ++b) lds_write2(xw + (unsigned)(decltype(PP)::value * 4 + b) * 512u, sv[4 + decltype(PP)::value][b]);
            });
        }
        wait_barrier<0>();
        // row pass (over i) + bias / residual / activation: this wave finishes units wi, wi + 6, wi + 12 (< 16) of the round
        const bool full = (h ? sqy[1] : sqy[0]) + kSQ <= ho && (h ? sqx[1] : sqx[0]) + kSQ <= wo;     // an interior square skips the per-pixel range tests
        for (int u = wi; u < 16; u += 6) {
            if (a.res_mode) { if (full) finish_act(std::true_type{}, std::true_type{}, h, u); else finish_act(std::false_type{}, std::true_type{}, h, u); }
            else if (full) finish_act(std::true_type{}, std::false_type{}, h, u);
            else finish_act(std::false_type{}, std::false_type{}, h, u);
        }
        if constexpr (h == 0) wait_barrier<0>();              // the exchange area is rewritten by the second round
    });
}

// Second kernel of the row-split forms: the row pass (over i) of A^T M A on the six rows' s[i][b] a row-split launch left in the scratch
// buffer, + bias / residual / activation: the same expressions on the same values as the fused epilogue.  One thread per (tile, output
// channel), channel fastest: 16-byte coalesced reads, 128-byte segments per pixel on the way out.
__global__ __launch_bounds__(256) void k_wino4_rowpass(ConvArgs a, int tx_n, int ty_n) {
    const int cout = a.cout_g;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, total = (int64_t)a.out.n * ty_n * tx_n * cout;
    if (idx >= total) return;
    const int co = (int)(idx % cout);
    const int64_t tile = idx / cout;
    const int txg = (int)(tile % tx_n), tyg = (int)((tile / tx_n) % ty_n), n = (int)(tile / ((int64_t)tx_n * ty_n));
    const float *sp = a.partial + tile * (int64_t)(24 * cout) + (int64_t)co * 4;
    f32x4n S[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) S[i] = *reinterpret_cast<const f32x4n *>(sp + (int64_t)i * cout * 4);
    f32x4n Y[4];                                              // Y[b] = (y[0][b], y[1][b], y[2][b], y[3][b])
    Y[0] = at6(S[0].x, S[1].x, S[2].x, S[3].x, S[4].x, S[5].x);
    Y[1] = at6(S[0].y, S[1].y, S[2].y, S[3].y, S[4].y, S[5].y);
    Y[2] = at6(S[0].z, S[1].z, S[2].z, S[3].z, S[4].z, S[5].z);
    Y[3] = at6(S[0].w, S[1].w, S[2].w, S[3].w, S[4].w, S[5].w);
    const float bias = a.bias ? a.bias[co] : 0.0f, slope = a.slope ? a.slope[co] : 0.0f;
    const int ho = a.out.h, wo = a.out.w;
    static_for<16>([&](auto E) {
        constexpr int e = decltype(E)::value, dy = e >> 2, dx = e & 3;
        const int oy = 4 * tyg + dy, ox = 4 * txg + dx;
        if (oy < ho && ox < wo) {
            const int64_t m = ((int64_t)n * ho + oy) * wo + ox;
            const f32x4n yb = Y[dx];
            float v = (dy == 0 ? yb.x : (dy == 1 ? yb.y : (dy == 2 ? yb.z : yb.w))) + bias;
            if (a.res_mode == 1) v += a.res.p[m * a.res.ld + co];
            v = apply_act(v, a.act, slope);
            if (a.res_mode == 2) v += a.res.p[m * a.res.ld + co];
            a.out.p[m * a.out.ld + co] = v;
        }
    });
}

}  // namespace

namespace csmconv {

bool wino4_eligible(const ConvArgs &a) {
    const int64_t bytes_in = (((int64_t)a.in.h * a.in.w - 1) * a.in.ld + a.in.c) * 4;           // ONE sample (a launch takes as many samples as fit 2 GiB)
    const int64_t bytes_w = (int64_t)(a.cout_g / 64) * (a.cin_g / 4) * (12 * kUW);
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dil == 1 && a.pad == 1 && a.groups == 1 && a.ksplit <= 1 && (a.cin_g & 31) == 0 &&
           (a.cout_g & 63) == 0 && bytes_in < (1ll << 31) && bytes_w < (1ll << 31) && !(a.in.ld & 3) && !(((uintptr_t)a.in.p | (uintptr_t)a.w) & 15) &&
           a.out.h == a.in.h && a.out.w == a.in.w &&
           (((int64_t)a.out.h * a.out.w - 1) * a.out.ld + a.out.c) * 4 < (1ll << 31) &&                   // (the epilogue's buffer descriptors: one sample)
           (!a.res_mode || (((int64_t)a.out.h * a.out.w - 1) * a.res.ld + a.res.c) * 4 < (1ll << 31));
}

// scratch floats a row-split launch needs for `n` samples of an h x w map with cout channels (24 per tile and channel)
int64_t wino4_scratch_floats(int n, int h, int w, int cout) { return (int64_t)n * ((h + 3) / 4) * ((w + 3) / 4) * 24 * cout; }

template <int ABL, int RPB>
static int launch_form(const ConvArgs &a, int sx_n, int sy_n, hipStream_t st) {
    static KernelPrep prep;
    const auto kern = &k_conv_wino4<ABL, RPB>;
    (void)prep.ensure([&] { return prepare_kernel(kern, 128 * RPB, lds_all_of(RPB)); });
    dim3 grid(a.m_tiles, (a.cout_g / 64) * (6 / RPB), 1);
    kern<<<grid, 128 * RPB, lds_all_of(RPB), st>>>(a, sx_n, sy_n);
    int rc = csm::check_launch("k_conv_wino4");
    if (rc || RPB == 6) return rc;
    const int tx_n = (a.out.w + 3) / 4, ty_n = (a.out.h + 3) / 4;
    const int64_t total = (int64_t)a.out.n * ty_n * tx_n * a.cout_g;
    k_wino4_rowpass<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(a, tx_n, ty_n);
    return csm::check_launch("k_wino4_rowpass");
}

// Which execution form (speed only: the forms give the same bits).  The fused form needs about one block tile per CU to be worth its
// 12-wave blocks; with fewer (one sample of an 80 x 80 map is 15 x 4 block tiles) the K loop of ONE block tile is the launch's whole
// duration, and the row-split forms cut it: 6 / RPB x the blocks of 2 RPB waves.  Times in us from the measured block times
// (profiles/r06_wino4_forms.txt: fused 12 + 1.40 per 4-channel step; RPB = 3 (one block per CU) 8 + 1.03; RPB = 2: 7 + 0.75 alone on a
// CU, 7 + 1.2 when two blocks share it; RPB = 1 never won and is not offered) + the row-pass kernel's round trip.
// CSM_WINO4_FORM = 6 / 3 / 2 / 1 forces a form (tests, measurements).
static int choose_form(const ConvArgs &a) {
    static const int forced = [] { const char *e = getenv("CSM_WINO4_FORM"); return e ? atoi(e) : 0; }();
    if (!a.partial) return 6;
    if (forced == 6 || forced == 3 || forced == 2 || forced == 1) return forced;
    const double ns = a.cin_g / 4, blocks = (double)a.m_tiles * (a.cout_g / 64);
    const double t_row = 3.0 + 2.5 * 4.0 * (double)a.M * a.cout_g / 3.0e6;
    auto ceil_div = [](double b, double s) { return (double)(int64_t)((b + s - 1) / s); };
    const double t6 = ceil_div(blocks, 256) * (12.0 + 1.40 * ns);
    const double t3 = ceil_div(2 * blocks, 256) * (8.0 + 1.03 * ns) + t_row;
    const double b2 = 3 * blocks, full2 = (double)(int64_t)(b2 / 512), rem2 = b2 - 512 * full2;
    const double t2 = full2 * (7.0 + 1.2 * ns) + (rem2 > 256 ? 7.0 + 1.2 * ns : (rem2 > 0 ? 7.0 + 0.75 * ns : 0.0)) + t_row;
    return t6 <= t3 && t6 <= t2 ? 6 : (t2 <= t3 ? 2 : 3);
}

static int launch_conv_wino4_chunk(const ConvArgs &a0, hipStream_t st) {
    ConvArgs a = a0;
    const int tiles_x = (a.out.w + kSQ - 1) / kSQ, tiles_y = (a.out.h + kSQ - 1) / kSQ;      // squares of 16 x 16 output pixels per sample
    a.m_tiles = (tiles_x * tiles_y * a.out.n + 1) / 2;                                     // two squares per block
#ifdef CSM_WINO_DEV
    const char *ve = getenv("CSM_WINO4_VARIANT");
    const int variant = ve ? atoi(ve) : 0;
    switch (variant) {
        case 1: return launch_form<1, 6>(a, tiles_x, tiles_y, st);
        case 2: return launch_form<2, 6>(a, tiles_x, tiles_y, st);
        case 4: return launch_form<4, 6>(a, tiles_x, tiles_y, st);
        case 8: return launch_form<8, 6>(a, tiles_x, tiles_y, st);
        case 3: return launch_form<3, 6>(a, tiles_x, tiles_y, st);
        case 16: return launch_form<16, 6>(a, tiles_x, tiles_y, st);
        case 32: return launch_form<32, 6>(a, tiles_x, tiles_y, st);
        case 35: return launch_form<35, 6>(a, tiles_x, tiles_y, st);
        case 59: return launch_form<59, 6>(a, tiles_x, tiles_y, st);
        case 64: return launch_form<64, 6>(a, tiles_x, tiles_y, st);
        default: break;
    }
#endif
    switch (choose_form(a)) {
        case 1: return launch_form<0, 1>(a, tiles_x, tiles_y, st);
        case 2: return launch_form<0, 2>(a, tiles_x, tiles_y, st);
        case 3: return launch_form<0, 3>(a, tiles_x, tiles_y, st);
        default: return launch_form<0, 6>(a, tiles_x, tiles_y, st);
    }
}

// 32-bit buffer descriptors: a launch covers as many SAMPLES as fit 2 GiB of input view; larger batches are split by sample (independent
// work: the same bits whatever the split).  CSM_WINO_MAX_BYTES lowers the limit (tests).
int launch_conv_wino4(const ConvArgs &a0, hipStream_t st) {
    const char *le = getenv("CSM_WINO_MAX_BYTES");
    const long long lv = le ? atoll(le) : 0;
    const int64_t limit = lv > 0 ? (int64_t)lv : (int64_t)((1ll << 31) - 1);
    const int64_t per_sample = (int64_t)a0.in.h * a0.in.w * a0.in.ld * 4;
    int chunk = (int)(limit / (per_sample > 0 ? per_sample : 1));
    if (chunk < 1) chunk = 1;
    if (chunk >= a0.in.n) return launch_conv_wino4_chunk(a0, st);
    for (int n0 = 0; n0 < a0.in.n; n0 += chunk) {
        ConvArgs a = a0;
        const int nn = a0.in.n - n0 < chunk ? a0.in.n - n0 : chunk;
        a.in.n = a.out.n = nn; a.in.p = a0.in.p + (int64_t)n0 * a0.in.h * a0.in.w * a0.in.ld;
        a.out.p = a0.out.p + (int64_t)n0 * a0.out.h * a0.out.w * a0.out.ld;
        if (a0.res_mode) { a.res.n = nn; a.res.p = a0.res.p + (int64_t)n0 * a0.res.h * a0.res.w * a0.res.ld; }
        if (a0.partial) a.partial = a0.partial + wino4_scratch_floats(n0, a0.out.h, a0.out.w, a0.cout_g);
        a.M = nn * a.out.h * a.out.w;
        const int rc = launch_conv_wino4_chunk(a, st);
        if (rc) return rc;
    }
    return CSM_OK;
}

}  // namespace csmconv
