// grouped.hip -- narrow grouped 3x3 convolutions (ResNeXt conv2: 8 / 16 / 32 channels per group, cin_g == cout_g, stride 1, pad 1) on the
// VECTOR pipe.  (reference layers: depth_modules/leres/leres/Resnext_torch.py:70-117, the grouped conv2 of every bottleneck)
//
// Why not the matrix pipe: the direct kernels run narrow groups block-diagonally inside 32 x 32 x 2 MFMA tiles -- 4x (8-channel groups) or
// 2x (16) of the executed products are zeros, and a group's K is only 72 ... 288 deep, so the tile set-up dominates what is left
// (25 / 47 / 86 TF/s natural at batch 8).  On gfx950 the packed fp32 vector rate EQUALS the fp32 matrix rate (256 FLOP / clk / CU either
// way), and a vector kernel multiplies no zeros.
//
// Arithmetic: the DIRECT contract, unchanged -- one fmaf chain per output, bias first, taps outer (a group has one 32-channel block),
// channels inside a tap in aligned 8-blocks, each in the order 0,4,1,5,2,6,3,7 (oracle/nets_oracle.c::orc_conv; the zeros the block-
// diagonal MFMA form adds do not change a chain).  Same bits as the direct kernels, another weight image (below): csm_op.flags bit 4,
// chosen by the lowering from the layer's shape, tests/test_gpu_nets.py grouped cases bit-exact against the oracle in both forms.
//
// Mapping.  A block = 4 waves = one tile of 8 x (8 PX) output pixels x one 32-channel SLAB (a 128-byte line of every pixel: 4 groups of 8,
// 2 of 16 or 1 of 32).  The (10) x (8 PX + 2) input patch of the slab goes to LDS once (coalesced 128-byte rows; 16-byte slots swizzled by
// (column + row) & 7: the b128 reads of 16 lanes = 2 rows x 8 columns touch all 64 banks once).  Wave w owns the 8 output channels
// 32 slab + 8 w .. + 8 (an OCTET of its group); lane (r, q) owns the pixels (r, q + 8 j), j < PX: 8 PX accumulators = 4 PX register
// pairs.  The octet's weights are wave-uniform: they stream through SGPRs (s_load, 32 floats per half unit, prefetched one ahead) and enter
// v_pk_fma_f32 as the scalar operand pair (w[co], w[co + 1]) against one input value broadcast to both halves (op_sel): per (tap, 8-block)
// 8 ds_read_b128-class reads feed 128 PX / 4 ... packed FMAs -- the kernel is bound by the vector pipe and, for the 8-channel layers at 160^2,
// by the 2 x 210 MB of activations that cross HBM.
// Weight image (host: program.py::pack_grouped_weights): [group][octet][tap][8-block kb][half h][chain position i][4] =
//   w[group CG + 8 octet + 4 h + t][8 kb + (i & 1) 4 + (i >> 1)][ky][kx]
#include "csm_conv.h"
#include <utility>

namespace csmconv {

typedef float f32x2g __attribute__((ext_vector_type(2)));
typedef float f32x4g __attribute__((ext_vector_type(4)));
typedef float f32x16g __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) f32x4g *lds_f4g;
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4g *lds_u4g;

// 16 wave-uniform floats -> SGPRs (asynchronous: the consumer waits on lgkmcnt)
__device__ __forceinline__ f32x16g sload16(const float *p) {
    f32x16g v;
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(v) : "s"(p));
    return v;
}

template <int OFF> __device__ __forceinline__ f32x4g lds_read4_async(unsigned byte_addr) {
    f32x4g v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}

// acc.lo = fma(w.lo, x.lo, acc.lo), acc.hi = fma(w.hi, x.lo, acc.hi)      (w: an SGPR pair)
__device__ __forceinline__ void pk_fma_xlo(f32x2g &acc, f32x2g w, f32x2g x) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "s"(w), "v"(x));
}
// acc.lo = fma(w.lo, x.hi, acc.lo), acc.hi = fma(w.hi, x.hi, acc.hi)
__device__ __forceinline__ void pk_fma_xhi(f32x2g &acc, f32x2g w, f32x2g x) {
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "s"(w), "v"(x));
}

constexpr int kGrpRows = 8, kGrpPH = kGrpRows + 2;
constexpr size_t grouped_lds_bytes(int px) { return (size_t)kGrpPH * (8 * px + 2) * 128; }

template <int CG, int PX>
__global__ __launch_bounds__(256, 3) void k_conv_grouped(ConvArgs a, int tx_n, int ty_n, int nslab) {
    constexpr int KB = CG / 8, OG = CG / 8, TW = 8 * PX, PW = TW + 2, PH = kGrpPH;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane >> 3, q = lane & 7;
    // block -> (pixel tile, slab): slab fastest, so that the eight XCDs (block i -> XCD i % 8) each walk neighbouring tiles of "their" slabs
    const int L = blockIdx.x, slab = L % nslab, tile = L / nslab;
    const int tx = tile % tx_n, t2 = tile / tx_n, ty = t2 % ty_n, n = t2 / ty_n;
    const int H = a.in.h, W = a.in.w;
    const int y0 = ty * kGrpRows - 1, x0 = tx * TW - 1;                     // patch origin in the image
    {   // ---- the slab's input patch -> LDS.  A wave-level load covers 8 consecutive pixels of ONE patch row x the 8 slots of their
        // 128-byte lines: the row and the run of eight are wave-uniform (scalar arithmetic), a lane adds its pixel and slot once.  Pixels
        // outside the image read as zeros through the buffer range check (offset 2^31).  KR runs per row, units dealt round-robin to the waves.
        constexpr int KR = (PW + 7) / 8, NU = PH * KR, UPW = (NU + 3) / 4;
        const int ldi = a.in.ld;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.in.p) + (int64_t)n * H * W * ldi + slab * 32, 0,
                                                                              (int)((((int64_t)H * W - 1) * ldi + 32) * 4), 0x00020000);
        const int pl = lane >> 3, cq = lane & 7;
        const unsigned lane_off = (unsigned)(pl * ldi + 4 * cq) * 4u;          // byte offset of (pixel pl of a run, slot cq)
        u32x4g v[UPW];
#pragma unroll
        for (int t = 0; t < UPW; ++t) {
            const int id = wave + 4 * t, row = id / KR, k = id - row * KR;   // (wave-uniform)
            const int iy = y0 + row, ix = x0 + 8 * k + pl, col = 8 * k + pl;
            const bool rowok = id < NU && iy >= 0 && iy < H;
            const bool ok = rowok && ix >= 0 && ix < W && col < PW;
            const int soff = rowok ? ((iy * W + x0 + 8 * k) * ldi) * 4 : 0;    // (scalar; may be negative at the left edge: the lanes that use it are inside)
            v[t] = __builtin_amdgcn_raw_buffer_load_b128(rin, ok ? (int)lane_off + soff : (int)0x80000000u, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < UPW; ++t) {
            const int id = wave + 4 * t, row = id / KR, k = id - row * KR, col = 8 * k + pl;
            if (id < NU && col < PW)
                *(lds_u4g)(size_t)(lds0 + (unsigned)((row * PW + col) * 8 + (cq ^ ((pl + row) & 7))) * 16u) = v[t];
        }
    }
    const int co0 = slab * 32 + wave * 8;                                    // this wave's eight output channels
    f32x2g acc[PX][4];
    {
        f32x2g b[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) b[p] = a.bias ? f32x2g{a.bias[co0 + 2 * p], a.bias[co0 + 2 * p + 1]} : f32x2g{0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < PX; ++j)
#pragma unroll
            for (int p = 0; p < 4; ++p) acc[j][p] = b[p];
    }
    // The wave's weight stream: half unit (tap, kb, h) = 32 floats = two s_load_dwordx16; the activations of a unit (tap, kb) = 2 PX
    // ds_read_b128.  Both are requested one half unit (16 PX packed FMAs per lane = 64 PX cycles) before their use, by hand: scalar loads
    // return out of order, so every wait is lgkmcnt(0) and a request must be issued BEHIND the wait for the previous one -- the compiler,
    // left to itself, issues each scalar load in front of its first use and stalls on it.  (Loads and waits are volatile asm in program
    // order; nothing touches a destination between its request and the wait -- checked in the ISA.)
    const float *wv = a.w + (int64_t)(slab * 4 + wave) * (9 * KB * 2 * 32);
    const int gs = wave / OG;                                                // the wave's group inside the slab
    f32x16g wA0 = sload16(wv), wA1 = sload16(wv + 16), wB0, wB1;
    f32x4g X[2][2][PX];                                                      // [unit parity][channels 0..3 | 4..7 of the 8-block][pixel]
    auto read_x = [&](int tap, int kb, f32x4g (&da)[PX], f32x4g (&db)[PX]) __attribute__((always_inline)) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const unsigned sw = (unsigned)(q + r + kx + ky) & 7u;
        const unsigned base = lds0 + (unsigned)(((r + ky) * PW + q + kx) * 8) * 16u;
        const unsigned sa = (unsigned)(gs * (CG / 4) + 2 * kb), sb = sa + 1u;
        const unsigned aa = base + ((sa ^ sw) << 4), ab = base + ((sb ^ sw) << 4);
        [&]<int... J>(std::integer_sequence<int, J...>) __attribute__((always_inline)) {
            ((da[J] = lds_read4_async<J * 1024>(aa), db[J] = lds_read4_async<J * 1024>(ab)), ...);
        }(std::make_integer_sequence<int, PX>{});
    };
    auto half_unit = [&](int h, const f32x16g &w0, const f32x16g &w1, const f32x4g (&xa)[PX], const f32x4g (&xb)[PX]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {                                        // chain position i = channel (i & 1) 4 + (i >> 1) of the 8-block
            const int c4 = i >> 1;                                           // component of xa (i even) / xb (i odd)
            const f32x16g &wq = i < 4 ? w0 : w1;
            const int o = (i & 3) * 4;
            const f32x2g wp0 = {wq[o], wq[o + 1]}, wp1 = {wq[o + 2], wq[o + 3]};
#pragma unroll
            for (int j = 0; j < PX; ++j) {
                const f32x4g xv = (i & 1) ? xb[j] : xa[j];
                const f32x2g xp = c4 < 2 ? f32x2g{xv.x, xv.y} : f32x2g{xv.z, xv.w};
                if (c4 & 1) { pk_fma_xhi(acc[j][2 * h], wp0, xp); pk_fma_xhi(acc[j][2 * h + 1], wp1, xp); }
                else        { pk_fma_xlo(acc[j][2 * h], wp0, xp); pk_fma_xlo(acc[j][2 * h + 1], wp1, xp); }
            }
        }
    };
    __syncthreads();
    read_x(0, 0, X[0][0], X[0][1]);
    [&]<int... U>(std::integer_sequence<int, U...>) __attribute__((always_inline)) {
        ([&] {
            constexpr int u = U;
            constexpr bool last = u + 1 == 9 * KB;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // wA and the activations of unit u have landed
            wB0 = sload16(wv + (2 * u + 1) * 32); wB1 = sload16(wv + (2 * u + 1) * 32 + 16);
            half_unit(0, wA0, wA1, X[u & 1][0], X[u & 1][1]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // wB
            if constexpr (!last) {
                wA0 = sload16(wv + (2 * u + 2) * 32); wA1 = sload16(wv + (2 * u + 2) * 32 + 16);
                read_x((u + 1) / KB, (u + 1) % KB, X[(u + 1) & 1][0], X[(u + 1) & 1][1]);
            }
            half_unit(1, wB0, wB1, X[u & 1][0], X[u & 1][1]);
        }(), ...);
    }(std::make_integer_sequence<int, 9 * KB>{});
    // ---- epilogue: two 16-byte stores per pixel.  ReLU / no activation without a residual (every ResNeXt conv2) is straight-line code;
    // anything else runs the direct kernels' epilogue arithmetic (residual before / after the activation) in a ROLLED loop over the lane's
    // pixels (one copy of the activation switch per channel instead of PX: it is the rare path)
    const int oy = ty * kGrpRows + r;
    const bool plain = (a.act == CSM_ACT_RELU || a.act == CSM_ACT_NONE) && !a.res_mode, relu = a.act == CSM_ACT_RELU;
    if (plain) {
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int ox = tx * TW + q + 8 * j;
            if (oy >= a.out.h || ox >= a.out.w) continue;
            const int64_t m = ((int64_t)n * a.out.h + oy) * a.out.w + ox;
            f32x4g v0 = {acc[j][0].x, acc[j][0].y, acc[j][1].x, acc[j][1].y}, v1 = {acc[j][2].x, acc[j][2].y, acc[j][3].x, acc[j][3].y};
            if (relu) {
                v0 = f32x4g{fmaxf(v0.x, 0.0f), fmaxf(v0.y, 0.0f), fmaxf(v0.z, 0.0f), fmaxf(v0.w, 0.0f)};
                v1 = f32x4g{fmaxf(v1.x, 0.0f), fmaxf(v1.y, 0.0f), fmaxf(v1.z, 0.0f), fmaxf(v1.w, 0.0f)};
            }
            float *dst = a.out.p + m * a.out.ld + co0;
            *reinterpret_cast<f32x4g *>(dst) = v0;
            *reinterpret_cast<f32x4g *>(dst + 4) = v1;
        }
    } else {
        float sl[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) sl[t] = (a.act == CSM_ACT_PRELU && a.slope) ? a.slope[co0 + t] : 0.0f;
#pragma nounroll
        for (int j = 0; j < PX; ++j) {
            f32x2g s[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                s[p] = acc[0][p];
#pragma unroll
                for (int jj = 1; jj < PX; ++jj) if (j == jj) s[p] = acc[jj][p];
            }
            const int ox = tx * TW + q + 8 * j;
            if (oy >= a.out.h || ox >= a.out.w) continue;
            const int64_t m = ((int64_t)n * a.out.h + oy) * a.out.w + ox;
            float v[8] = {s[0].x, s[0].y, s[1].x, s[1].y, s[2].x, s[2].y, s[3].x, s[3].y}, rv[8] = {};
            if (a.res_mode) {
                const f32x4g r0 = *reinterpret_cast<const f32x4g *>(a.res.p + m * a.res.ld + co0), r1 = *reinterpret_cast<const f32x4g *>(a.res.p + m * a.res.ld + co0 + 4);
                rv[0] = r0.x; rv[1] = r0.y; rv[2] = r0.z; rv[3] = r0.w; rv[4] = r1.x; rv[5] = r1.y; rv[6] = r1.z; rv[7] = r1.w;
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float x = v[t];
                if (a.res_mode == 1) x += rv[t];
                x = apply_act(x, a.act, sl[t]);
                if (a.res_mode == 2) x += rv[t];
                v[t] = x;
            }
            float *dst = a.out.p + m * a.out.ld + co0;
            *reinterpret_cast<f32x4g *>(dst) = f32x4g{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4g *>(dst + 4) = f32x4g{v[4], v[5], v[6], v[7]};
        }
    }
}

bool grouped_eligible(const ConvArgs &a) {
    const int C = a.groups * a.cin_g;
    return a.groups > 1 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.dil == 1 && a.pad == 1 && a.cin_g == a.cout_g &&
           (a.cin_g == 8 || a.cin_g == 16 || a.cin_g == 32) && C % 32 == 0 && a.ksplit == 1 && a.in.c == C && a.out.c == C &&
           a.out.h == a.in.h && a.out.w == a.in.w && !(a.in.ld & 3) && !(a.out.ld & 3) && !(((uintptr_t)a.in.p | (uintptr_t)a.out.p) & 15) &&
           (int64_t)a.in.h * a.in.w * a.in.ld * 4 < (1ll << 31) &&          // one sample's input view is addressed with 32-bit byte offsets (buffer loads)
           (!a.res_mode || (!(a.res.ld & 3) && !(((uintptr_t)a.res.p) & 15)));
}

template <int CG, int PX> static int launch_grouped(const ConvArgs &a, hipStream_t st) {
    static KernelPrep prep;
    constexpr size_t lds = grouped_lds_bytes(PX);
    prep.ensure([&] { return prepare_kernel(k_conv_grouped<CG, PX>, 256, lds); });
    const int tx_n = (a.out.w + 8 * PX - 1) / (8 * PX), ty_n = (a.out.h + kGrpRows - 1) / kGrpRows, nslab = a.groups * a.cin_g / 32;
    const int64_t blocks = (int64_t)tx_n * ty_n * a.out.n * nslab;
    if (blocks <= 0 || blocks > 0x7fffffff) { csm::set_error("k_conv_grouped: %lld blocks", (long long)blocks); return CSM_ERR_ARG; }
    k_conv_grouped<CG, PX><<<dim3((unsigned)blocks), 256, lds, st>>>(a, tx_n, ty_n, nslab);
    return csm::check_launch("k_conv_grouped");
}

int launch_conv_grouped(const ConvArgs &a, hipStream_t st) {
    // pixels per lane (tile width 32 or 40): whichever pads the row less; 40 on a tie (more work per weight load)
    static const int force = [] { const char *e = getenv("CSM_GROUPED_PX"); return e ? atoi(e) : 0; }();
    const int w4 = (a.out.w + 31) / 32 * 32, w5 = (a.out.w + 39) / 40 * 40;
    const bool five = force ? force == 5 : w5 <= w4;
    switch (a.cin_g) {
        case 8: return five ? launch_grouped<8, 5>(a, st) : launch_grouped<8, 4>(a, st);
        case 16: return five ? launch_grouped<16, 5>(a, st) : launch_grouped<16, 4>(a, st);
        case 32: return five ? launch_grouped<32, 5>(a, st) : launch_grouped<32, 4>(a, st);
    }
    csm::set_error("k_conv_grouped: %d channels per group", a.cin_g);
    return CSM_ERR_ARG;
}

}  // namespace csmconv
