// tokens.hip -- the transformer ops of the layer-program executor (CSM_OP_LAYERNORM / ATTENTION / TOKENS / DEPTH_TO_SPACE,
// include/csm355.h): what the MiDaS DPT-BEiT core of ZoeDepth needs next to the convolution engine of nets.hip.  Every nn.Linear of the
// network is a 1x1 convolution on that engine; this file holds the reductions (LayerNorm over channels, softmax over keys) and the
// attention core  softmax(q k^T + relative position bias) v  on the exact-fp32 matrix pipe.
//
// The attention kernel streams the keys with a running max / sum (no N x N matrix): see k_attention.
#include "csm_common.h"
#include "csm_tokens.h"
#include <mutex>
#include <utility>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- LayerNorm over the channels of every pixel: one wave per row, two passes (mean, then the centred second moment) -------------------
__global__ __launch_bounds__(256) void k_layernorm(const float *__restrict__ in, int in_ld, float *__restrict__ out, int out_ld, int64_t rows,
                                                   int c, const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ eps_p) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *x = in + row * in_ld;
    float s = 0.0f;
    for (int i = lane * 4; i < c; i += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i);
        s += (v.x + v.y) + (v.z + v.w);
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.0f;
    for (int i = lane * 4; i < c; i += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i);
        const float a = v.x - mean, b = v.y - mean, d = v.z - mean, e = v.w - mean;
        q += (a * a + b * b) + (d * d + e * e);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)c + eps_p[0]);
    float *y = out + row * out_ld;
    for (int i = lane * 4; i < c; i += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(x + i);
        const float4 g = *reinterpret_cast<const float4 *>(gamma + i), b = *reinterpret_cast<const float4 *>(beta + i);
        float4 r;
        r.x = (v.x - mean) * rstd * g.x + b.x; r.y = (v.y - mean) * rstd * g.y + b.y;
        r.z = (v.z - mean) * rstd * g.z + b.z; r.w = (v.w - mean) * rstd * g.w + b.w;
        *reinterpret_cast<float4 *>(y + i) = r;
    }
}

// ---- attention core: softmax(q k^T + relative position bias) v, streamed over the keys with a running max / sum (no N x N matrix, no
// limit on the sequence length) --------------------------------------------------------------------------------------------------------------
// D = head dimension (32, 64 or 128).  qkv rows: [q (heads*D) | k (heads*D) | v (heads*D)], pitch ld floats; q carries the 1/sqrt(D) scale.
// A block = 4 waves = (2 tiles of 32 queries) x (2 halves of the key range); the two halves of a query tile are merged at the end
// (m = max, l and o rescaled -- the usual split-softmax identity).  Per iteration the block stages one 32-key tile of K and of V for EACH
// half into LDS with coalesced 256-B row loads (row pitch D + 4 floats: conflict-free fragment reads), then every wave runs, for its 32
// queries x 32 keys:
//   s^T = K q^T   on v_mfma_f32_32x32x2_f32 with A = K rows, B = q (kept in registers): a lane holds ONE query (its column) and 16 of the 32
//                 keys (rows (r & 3) + 8 (r >> 2) + 4 lh), lane ^ 32 the other 16 -- row statistics need one cross-lane exchange, not a tree;
//   + bias        BEiT's relative position bias, index = C(query) - K(key) computed from the token grid (the keys' part staged per tile);
//   p = exp(s - m_new), l and o rescaled by exp(m_old - m_new): per-lane scalars, because a lane's o registers belong to its own query;
//   o^T += V^T p^T with A = V (staged tile, a lane reads V[key][its dimension]) and B = p STRAIGHT FROM THE SCORE REGISTERS (MFMA step r
//                 pairs the keys of register r in the two lane halves): P never goes through LDS.
// 64 MFMAs per wave and tile, none wasted.  Reductions are tolerance-level against the oracle (see the header).
// The relative position bias of a (32 queries) x (32 keys) tile pair comes from a WINDOW of the head's table: 32 consecutive tokens span
// at most R = 31 / gw + 2 grid rows, so yi - yj takes at most 2 R - 1 values and the pair needs the (2 R - 1) (2 gw - 1) consecutive table
// entries starting at row yq0 - yk0 - R + gh (249 floats at 42 x 42).  Every wave fetches the window of its NEXT key tile into its own LDS
// region by DMA (buffer_load_dword ... lds: no registers, out-of-table entries read as 0 through the buffer range check) behind the softmax
// of the current one, and the 16 bias values of a lane are LDS reads at  base(query) - kterm(key).  (Gathering them from the table in
// global memory -- 16 scattered loads per lane and tile -- kept the texture path as busy as the matrix pipe: 216 -> 150 us per layer at
// 1765 tokens.)  wcap = floats of one wave's window (a multiple of 64), 0 = window too large for the LDS budget: gather from global memory.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma4(unsigned voff, i32x4_t rsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_byte_addr) : "memory");
}

template <int D, bool WIN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_attention(const float *__restrict__ qkv, int ld, float *__restrict__ out, int out_ld, int N, int heads,
                                                   const float *__restrict__ table /* [heads][T] or null */, int gh, int gw, int wcap) {
    constexpr int NCT = D / 32, PITCH = D + 4, TILE = 32 * PITCH;       // floats of one staged K or V tile
    extern __shared__ __attribute__((aligned(16))) float lds[];          // [half][K tile | V tile], [half][32] key terms of the bias index, [wave][wcap] bias windows
    // half h: K tile at lds + 2 h TILE, V tile behind it
    int *kterm = reinterpret_cast<int *>(lds + 4 * TILE);               // [2][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int qt = wave & 1, kh = wave >> 1;
    const int head = blockIdx.y, b = blockIdx.z, C = heads * D;
    const int q0 = (blockIdx.x * 2 + qt) * 32, qi = q0 + li;
    const float *base = qkv + (int64_t)b * N * ld;
    const float *Q = base + head * D, *K = base + C + head * D, *V = base + 2 * C + head * D;
    const int T = (2 * gh - 1) * (2 * gw - 1) + 3;
    const float *tab = table ? table + (int64_t)head * T : nullptr;
    // the query's part of the bias index: idx(i, j) = Ci - Kj for patch tokens, Kj = yj (2 gw - 1) + xj
    int Ci = 0, yi = 0, xi = 0;
    if (tab && qi >= 1) { yi = (qi - 1) / gw; xi = (qi - 1) - yi * gw; Ci = yi * (2 * gw - 1) + xi + (gh - 1) * (2 * gw - 1) + gw - 1; }
    // bias window of this wave (see above)
    const int gwd = tab ? gw : 1;                                      // (no table: gh = gw = 0 -- the index arithmetic below is unused, keep it defined)
    const int W2 = 2 * gw - 1, R = 31 / gwd + 2, WN = (2 * R - 1) * W2, yq0 = q0 >= 1 ? (q0 - 1) / gwd : 0;
    constexpr bool win = WIN;                                          // (the launcher: WIN <=> table && wcap > 0)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    float *wbuf = lds + 4 * TILE + 64 + wave_u * wcap;
    const unsigned wbuf_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)lds + (unsigned)(4 * TILE + 64 + wave_u * wcap) * 4u;
    i32x4_t rtab = {0, 0, 0, 0};
    float b_cp = 0.0f, b_pc = 0.0f, b_cc = 0.0f;                       // class token -> patch, patch -> class token, class -> class
    if (tab) {
        const uint64_t pt = (uint64_t)tab;
        rtab = i32x4_t{(int)(unsigned)pt, (int)(unsigned)(pt >> 32), (int)((unsigned)(T - 3) * 4u), 0x00020000};
        b_cp = tab[T - 3]; b_pc = tab[T - 2]; b_cc = tab[T - 1];
    }
    auto fetch_window = [&](int tl) __attribute__((always_inline)) {   // the window of key tile tl -> wbuf (asynchronous: vmcnt)
        const int j0 = tl * 32, yk0 = j0 >= 1 ? (j0 - 1) / gwd : 0;
        const int g0 = (yq0 - yk0 - R + gh) * W2;
        for (int u = 0; u * 64 < wcap; ++u) {
            const int e = u * 64 + lane;
            dma4(e < WN ? (unsigned)((g0 + e) * 4) : 0x80000000u, rtab, wbuf_addr + (unsigned)u * 256u);
        }
    };

    float4 qf[D / 8];
    {
        const float *qrow = Q + (int64_t)min(qi, N - 1) * ld;
#pragma unroll
        for (int kb = 0; kb < D / 8; ++kb) qf[kb] = *reinterpret_cast<const float4 *>(qrow + 8 * kb + 4 * lh);
    }
    f32x16 o[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] = 0.0f;
    float m_run = -3.0e38f, l_run = 0.0f;

    const int NT = (N + 31) >> 5, half = (NT + 1) >> 1;
    // the K / V tiles of iteration it + 1 are fetched into registers BEFORE the MFMAs of iteration it (their round trip runs under the
    // compute) and stored to LDS behind the barrier that ends it: per thread 2 halves x (K + V) x 32 x D / 4 / 256 float4
    constexpr int NPF = 2 * 32 * (D / 4) / 256;
    // (native vectors: HIP's float4 is a struct, its global -> private -> LDS copies stay memcpys that SROA does not split, and the
    // arrays end up in scratch)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 pk[NPF], pv[NPF];
    auto prefetch = [&](int itn) __attribute__((always_inline)) {
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ([&] {
                constexpr int e_ = U * 256;
                const int e = tid + e_;
                const int hs = e / (32 * (D / 4)), rem = e - hs * (32 * (D / 4)), row = rem / (D / 4), c4 = rem - row * (D / 4);
                const int key = min((hs * half + itn) * 32 + row, N - 1);
                pk[U] = *reinterpret_cast<const f32x4 *>(K + (int64_t)key * ld + 4 * c4);
                pv[U] = *reinterpret_cast<const f32x4 *>(V + (int64_t)key * ld + 4 * c4);
            }(), ...);
        }(std::make_integer_sequence<int, NPF>{});
    };
    prefetch(0);
    if (win && kh * half < NT) fetch_window(kh * half);
    for (int it = 0; it < half; ++it) {
        // ---- stage the two halves' K and V tiles (rows beyond N repeat row N - 1: finite values, their probabilities are forced to 0)
        __syncthreads();                                                // everybody is done with the previous tiles
        [&]<int... U>(std::integer_sequence<int, U...>) {
            ([&] {
                const int e = tid + U * 256;
                const int hsel = e / (32 * (D / 4)), rem = e - hsel * (32 * (D / 4)), row = rem / (D / 4), c4 = rem - row * (D / 4);
                *reinterpret_cast<f32x4 *>(lds + 2 * hsel * TILE + row * PITCH + 4 * c4) = pk[U];
                *reinterpret_cast<f32x4 *>(lds + (2 * hsel + 1) * TILE + row * PITCH + 4 * c4) = pv[U];
            }(), ...);
        }(std::make_integer_sequence<int, NPF>{});
        if (tid < 64) {
            const int j = ((tid >> 5) * half + it) * 32 + (tid & 31);
            int kt = 0;
            if (j >= 1) { const int yj = (j - 1) / gw, xj = (j - 1) - yj * gw; kt = yj * (2 * gw - 1) + xj; }
            kterm[tid] = kt;
        }
        __syncthreads();
        prefetch(min(it + 1, half - 1));
        const int tile = kh * half + it;
        if (tile >= NT) continue;                                       // (odd tile counts: the second half has one tile fewer)
        const float *Kt = lds + 2 * kh * TILE, *Vt = Kt + TILE;
        const int yk0 = tile >= 1 ? (tile * 32 - 1) / gwd : 0;
        const int wbase = (yi - yq0 + yk0 + R - 1) * W2 + xi + gw - 1;  // window index of (this query, key) = wbase - kterm(key)
        // ---- s^T = K q^T
        f32x16 sacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < D / 8; ++kb) {
            const float4 kf = *reinterpret_cast<const float4 *>(Kt + li * PITCH + 8 * kb + 4 * lh);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[kb].x, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[kb].y, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[kb].z, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[kb].w, sacc, 0, 0, 0);
        }
        // ---- bias, padded keys, running max.  Interior tile pairs (no class token, no padding on either side: all but the first / last
        // tiles) take a path without per-element conditions: 16 + 16 LDS reads issued together, one add and one max per score
        float mt = -3.0e38f;
        if (win) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this tile's window has landed (issued one tile ago, behind the K / V prefetch)
        const bool interior = tile > 0 && q0 > 0 && tile * 32 + 32 <= N && q0 + 32 <= N;      // (wave-uniform)
        if (win && interior) {
            int kt[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) kt[r] = kterm[kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
            for (int r = 0; r < 16; ++r) { sacc[r] += wbuf[wbase - kt[r]]; mt = fmaxf(mt, sacc[r]); }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2) + 4 * lh, j = tile * 32 + krow;
                float v = sacc[r];
                if (tab) {
                    float bv;
                    if (win) {
                        const float w = wbuf[min(max(wbase - kterm[kh * 32 + krow], 0), WN - 1)];   // (clamped: padded queries / keys carry meaningless indices)
                        bv = qi == 0 ? (j == 0 ? b_cc : b_cp) : (j == 0 ? b_pc : w);
                    } else {
                        int idx;
                        if (qi == 0) idx = j == 0 ? T - 1 : T - 3;
                        else if (j == 0) idx = T - 2;
                        else idx = Ci - kterm[kh * 32 + krow];
                        bv = tab[min(max(idx, 0), T - 1)];
                    }
                    v += bv;
                }
                if (j >= N) v = -3.0e38f;
                sacc[r] = v;
                mt = fmaxf(mt, v);
            }
        }
        if (win && it + 1 < half && tile + 1 < NT) {                    // the next tile's window, under the rest of this tile
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the reads of this one have returned)
            fetch_window(tile + 1);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        // p = 2^((s - m) log2 e) on the raw v_exp_f32 (arguments <= 0: a result below 2^-126 is 0 either way at the tolerance of this op);
        // a padded key's score is -3e38: its p is exactly 0 without a test
        // ((s - m) log2 e with the subtraction FIRST: the two products rounded separately lose |s| 6e-8 1.44 absolute -- fine for logits
        // of a few units, 1e-3 relative for logits in the thousands, and p could exceed 1)
        float lt = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f((sacc[r] - m_new) * 1.44269504088896341f);
            sacc[r] = pv;
            lt += pv;
        }
        lt += __shfl_xor(lt, 32, 64);
        // the running maximum settles after a few tiles: rescale l and o only when some query of the wave moved (wave-uniform test; alpha
        // is exactly 1 otherwise)
        if (__builtin_amdgcn_ballot_w64(m_new != m_run) != 0) {
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * 1.44269504088896341f);
            l_run *= alpha;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
        }
        l_run += lt;
        m_run = m_new;
        // ---- o^T += V^T p^T
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2) + 4 * lh;
                o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vt[krow * PITCH + 32 * ct + li], sacc[r], o[ct], 0, 0, 0);
            }
        }
    }
    // ---- merge the two key halves of each query tile (the second half hands (m, l, o) over through LDS), normalise, store ----------------
    __syncthreads();
    float *X = lds + qt * (64 * (16 * NCT + 2));                        // per query tile: [lane][16 NCT + 2]
    if (kh == 1) {
        float *x = X + lane * (16 * NCT + 2);
        x[0] = m_run; x[1] = l_run;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) x[2 + 16 * ct + r] = o[ct][r];
    }
    __syncthreads();
    if (kh == 0) {
        const float *x = X + lane * (16 * NCT + 2);
        const float m1 = x[0], l1 = x[1];
        const float m = fmaxf(m_run, m1);
        const float a0 = exp2f((m_run - m) * 1.44269504088896341f), a1 = exp2f((m1 - m) * 1.44269504088896341f);
        const float inv = 1.0f / (l_run * a0 + l1 * a1);
        if (qi < N) {
            float *orow = out + ((int64_t)b * N + qi) * out_ld + head * D;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {                          // registers 4g .. 4g + 3 = dimensions 32 ct + 8 g + 4 lh + {0, 1, 2, 3}
                    float4 v;
                    v.x = (o[ct][4 * g + 0] * a0 + x[2 + 16 * ct + 4 * g + 0] * a1) * inv;
                    v.y = (o[ct][4 * g + 1] * a0 + x[2 + 16 * ct + 4 * g + 1] * a1) * inv;
                    v.z = (o[ct][4 * g + 2] * a0 + x[2 + 16 * ct + 4 * g + 2] * a1) * inv;
                    v.w = (o[ct][4 * g + 3] * a0 + x[2 + 16 * ct + 4 * g + 3] * a1) * inv;
                    *reinterpret_cast<float4 *>(orow + 32 * ct + 8 * g + 4 * lh) = v;
                }
        }
    }
}

// ---- token plumbing -----------------------------------------------------------------------------------------------------------------------
// mode 0: out row 0 = cls, row 1 + i = patch i.  mode 1: out[i] = (token 1 + i | token 0).  mode 2: out[i] = token 1 + i.   c % 4 == 0.
__global__ __launch_bounds__(256) void k_tokens(int mode, const float *__restrict__ in, int in_ld, float *__restrict__ out, int out_ld, int n, int np,
                                                int c, const float *__restrict__ cls) {
    const int c4 = c >> 2;
    const int64_t rows_out = mode == 0 ? (int64_t)n * (np + 1) : (int64_t)n * np;
    const int per_row = mode == 1 ? 2 * c4 : c4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows_out * per_row) return;
    const int64_t row = idx / per_row;
    const int q = (int)(idx - row * per_row);
    float4 v;
    if (mode == 0) {
        const int64_t b = row / (np + 1), t = row - b * (np + 1);
        v = t == 0 ? *reinterpret_cast<const float4 *>(cls + 4 * q) : *reinterpret_cast<const float4 *>(in + (b * np + t - 1) * in_ld + 4 * q);
    } else {
        const int64_t b = row / np, i = row - b * np;
        const float *src = q < c4 ? in + (b * (np + 1) + 1 + i) * in_ld + 4 * q : in + (b * (np + 1)) * in_ld + 4 * (q - c4);
        v = *reinterpret_cast<const float4 *>(src);
    }
    *reinterpret_cast<float4 *>(out + row * out_ld + 4 * q) = v;
}

__global__ __launch_bounds__(256) void k_depth_to_space(const float *__restrict__ in, int in_ld, float *__restrict__ out, int out_ld, int n, int h, int w,
                                                        int k, int c) {
    const int c4 = c >> 2;
    const int64_t total = (int64_t)n * h * k * w * k * c4;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int q = (int)(idx % c4);
    int64_t p = idx / c4;
    const int ox = (int)(p % (w * k)); p /= (w * k);
    const int oy = (int)(p % (h * k));
    const int64_t b = p / (h * k);
    const int y = oy / k, ky = oy - y * k, x = ox / k, kx = ox - x * k;
    const float4 v = *reinterpret_cast<const float4 *>(in + ((b * h + y) * w + x) * in_ld + (ky * k + kx) * c + 4 * q);
    *reinterpret_cast<float4 *>(out + ((b * h * k + oy) * (int64_t)(w * k) + ox) * out_ld + 4 * q) = v;
}

}  // namespace

namespace csm {

int launch_layernorm(const float *in, int in_ld, float *out, int out_ld, int64_t rows, int c, const float *gamma, const float *beta,
                     const float *eps /* device pointer */, hipStream_t st) {
    if ((c & 3) || (in_ld & 3) || (out_ld & 3) || !gamma || !beta || !eps) { set_error("layernorm: channels %% 4 == 0 and gamma / beta required"); return CSM_ERR_ARG; }
    k_layernorm<<<(unsigned)((rows + 3) / 4), 256, 0, st>>>(in, in_ld, out, out_ld, rows, c, gamma, beta, eps);
    return check_launch("k_layernorm");
}

static int g_attention_options = 0;      // csm_debug_attention_options: bit 0 = no bias window (the large-grid path), for tests

template <int D> static int launch_attention_t(const float *qkv, int ld, float *out, int out_ld, int n, int N, int heads, const float *table, int gh,
                                               int gw, hipStream_t st) {
    constexpr int kTiles = 4 * 32 * (D + 4) + 64, kExchange = 2 * 64 * (16 * (D / 32) + 2);
    // bias windows: (2 R - 1) (2 gw - 1) floats per wave, rounded up to whole 64-lane DMA pieces; at most 2 K floats per wave (grids up to
    // ~340 tokens wide), else the kernel gathers from the table in global memory
    int wcap = 0;
    if (table && !(g_attention_options & 1)) {
        const int R = 31 / gw + 2, WN = (2 * R - 1) * (2 * gw - 1);
        wcap = (WN + 63) / 64 * 64;
        if (wcap > 2048) wcap = 0;
    }
    size_t fl = (size_t)kTiles + 4 * (size_t)wcap;
    if (fl < (size_t)kExchange) fl = kExchange;
    const size_t lds = sizeof(float) * fl;
    const dim3 grid((unsigned)((N + 63) / 64), (unsigned)heads, (unsigned)n);
    // the dynamic-LDS limit is raised when a launch needs more than the kernel was last prepared for (per device; checked: an
    // oversized request used to surface only as a generic launch error)
    auto prepare = [&](const void *fn, size_t &have) -> int {
        if (lds <= have) return CSM_OK;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("attention: %zu bytes of LDS requested: %s", lds, hipGetErrorString(e)); return CSM_ERR_HIP; }
        have = lds;
        return CSM_OK;
    };
    static std::mutex prep_mutex;
    static size_t have_w[32] = {}, have_g[32] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 31;
    if (wcap > 0) {
        { std::lock_guard<std::mutex> lk(prep_mutex); int rc = prepare(reinterpret_cast<const void *>(&k_attention<D, true>), have_w[dev]); if (rc) return rc; }
        k_attention<D, true><<<grid, 256, lds, st>>>(qkv, ld, out, out_ld, N, heads, table, gh, gw, wcap);
    } else {
        { std::lock_guard<std::mutex> lk(prep_mutex); int rc = prepare(reinterpret_cast<const void *>(&k_attention<D, false>), have_g[dev]); if (rc) return rc; }
        k_attention<D, false><<<grid, 256, lds, st>>>(qkv, ld, out, out_ld, N, heads, table, gh, gw, 0);
    }
    return check_launch("k_attention");
}

int launch_attention(const float *qkv, int ld, float *out, int out_ld, int n, int N, int heads, int d, const float *table, int gh, int gw,
                     hipStream_t st) {
    if (N < 1) { set_error("attention: empty sequence"); return CSM_ERR_ARG; }
    if ((ld & 3) || (out_ld & 3) || (((uintptr_t)qkv | (uintptr_t)out) & 15)) { set_error("attention: qkv / out must be 16-byte aligned"); return CSM_ERR_ARG; }
    if (table && (gh < 1 || gw < 1)) { set_error("attention: relative position bias needs a token grid (gh, gw >= 1), got %d x %d", gh, gw); return CSM_ERR_ARG; }
    if (table && gh * gw + 1 != N) { set_error("attention: relative position bias needs N == gh * gw + 1 (%d x %d vs %d)", gh, gw, N); return CSM_ERR_ARG; }
    switch (d) {
        case 32: return launch_attention_t<32>(qkv, ld, out, out_ld, n, N, heads, table, gh, gw, st);
        case 64: return launch_attention_t<64>(qkv, ld, out, out_ld, n, N, heads, table, gh, gw, st);
        case 128: return launch_attention_t<128>(qkv, ld, out, out_ld, n, N, heads, table, gh, gw, st);
        default: set_error("attention: head dimension %d not built (32, 64, 128)", d); return CSM_ERR_ARG;
    }
}

int launch_tokens(int mode, const float *in, int in_ld, float *out, int out_ld, int n, int np, int c, const float *cls, hipStream_t st) {
    if ((c & 3) || (in_ld & 3) || (out_ld & 3) || mode < 0 || mode > 2 || (mode == 0 && !cls)) { set_error("tokens: bad mode / channels"); return CSM_ERR_ARG; }
    const int64_t total = (mode == 0 ? (int64_t)n * (np + 1) : (int64_t)n * np) * (mode == 1 ? 2 : 1) * (c >> 2);
    k_tokens<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(mode, in, in_ld, out, out_ld, n, np, c, cls);
    return check_launch("k_tokens");
}

int launch_depth_to_space(const float *in, int in_ld, float *out, int out_ld, int n, int h, int w, int k, int c, hipStream_t st) {
    if ((c & 3) || (in_ld & 3) || (out_ld & 3) || k < 1) { set_error("depth_to_space: channels %% 4 == 0"); return CSM_ERR_ARG; }
    const int64_t total = (int64_t)n * h * k * w * k * (c >> 2);
    k_depth_to_space<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(in, in_ld, out, out_ld, n, h, w, k, c);
    return check_launch("k_depth_to_space");
}

}  // namespace csm

// test aid (not stable ABI): bit 0 = attention gathers the relative position bias from the table in global memory even when the window
// fits the LDS budget (the path grids wider than ~340 tokens take)
extern "C" int csm_debug_attention_options(int options) {
    csm::g_attention_options = options;
    return CSM_OK;
}
