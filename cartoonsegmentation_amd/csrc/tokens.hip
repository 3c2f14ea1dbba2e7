// tokens.hip -- the transformer ops of the layer-program executor (CSM_OP_LAYERNORM / ATTENTION / TOKENS / DEPTH_TO_SPACE,
// include/csm355.h): what the MiDaS DPT-BEiT core of ZoeDepth needs next to the convolution engine of nets.hip.  Every nn.Linear of the
// network is a 1x1 convolution on that engine; this file holds the reductions (LayerNorm over channels, softmax over keys) and the
// attention core  softmax(q k^T + relative position bias) v  on the exact-fp32 matrix pipe.
//
// Attention kernel, one block (4 waves) per (image, head, 32-query tile):
//   1. scores: the tile's 32 x N score block is computed 32 keys at a time on v_mfma_f32_32x32x2_f32 (A = the wave's q fragments, kept in
//      registers; B = k rows straight from global memory: a lane reads 16 B of one key row) -- key tiles round-robin over the waves -- and
//      written to LDS with BEiT's relative position bias added (index computed arithmetically from the token grid: no N x N table);
//   2. softmax over each row in LDS (a wave per 8 rows, shuffle reductions), padded keys get probability 0;
//   3. out = P V on the matrix pipe: waves = (32-wide slice of the head dimension) x (a part of the key range), the parts summed in a
//      fixed order through LDS.  P fragments come from LDS (row pitch N + 4 floats: conflict-free ds_read_b128), V rows from global.
// LDS = QT x (N + 4) floats with QT = 32 query rows per block up to N = 1216 tokens and QT = 16 up to N = 2496 (ZoeDepth's 672 x 672
// input: 42 x 42 + 1 = 1765 tokens; the 32-row MFMA then carries every query twice -- the attention core is a tenth of the network's
// FLOPs); longer sequences are refused.
// Reductions here are tolerance-level against the oracle (not bit-exact): see the header.
#include "csm_common.h"
#include "csm_tokens.h"

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

// ---- attention core ------------------------------------------------------------------------------------------------------------------------
// D = head dimension (32, 64 or 128).  qkv rows: [q (heads*D) | k (heads*D) | v (heads*D)], pitch ld floats.
template <int D, int QT>
__global__ __launch_bounds__(256) void k_attention(const float *__restrict__ qkv, int ld, float *__restrict__ out, int out_ld, int N, int heads,
                                                   const float *__restrict__ table, int gh, int gw) {
    constexpr int NCOL = D / 32, KPARTS = 4 / NCOL;                    // PV: column tiles of the head dimension x parts of the key range
    extern __shared__ __attribute__((aligned(16))) float S[];          // [QT][pitch]
    const int Npad = (N + 31) & ~31, pitch = Npad + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int q0 = blockIdx.x * QT, head = blockIdx.y, b = blockIdx.z;
    const int qrow = li & (QT - 1);                                     // QT = 16: MFMA rows 16 .. 31 repeat rows 0 .. 15 (results discarded)
    const int C = heads * D;
    const float *base = qkv + (int64_t)b * N * ld;
    const float *Q = base + head * D, *K = base + C + head * D, *V = base + 2 * C + head * D;

    // ---- 1. scores ------------------------------------------------------------------------------------------------------------------
    float4 qf[D / 8];
    {
        const int qi = min(q0 + qrow, N - 1);
#pragma unroll
        for (int kb = 0; kb < D / 8; ++kb) qf[kb] = *reinterpret_cast<const float4 *>(Q + (int64_t)qi * ld + 8 * kb + 4 * lh);
    }
    const int T = (2 * gh - 1) * (2 * gw - 1) + 3;
    for (int kt = wave; kt * 32 < Npad; kt += 4) {
        const int kj = min(kt * 32 + li, N - 1);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kb = 0; kb < D / 8; ++kb) {
            const float4 kf = *reinterpret_cast<const float4 *>(K + (int64_t)kj * ld + 8 * kb + 4 * lh);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kb].x, kf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kb].y, kf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kb].z, kf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[kb].w, kf.w, acc, 0, 0, 0);
        }
        // lane holds key column li of the tile, query rows (r & 3) + 8 (r >> 2) + 4 lh
        const int j = kt * 32 + li;
        const int yj = j > 0 ? (j - 1) / gw : 0, xj = j > 0 ? (j - 1) - yj * gw : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh, i = q0 + row;
            if (row >= QT) continue;
            float v = acc[r];
            if (j >= N) v = -3.0e38f;                                   // padded key: probability 0
            else if (table && i < N) {
                int idx;
                if (i == 0) idx = j == 0 ? T - 1 : T - 3;
                else if (j == 0) idx = T - 2;
                else { const int yi = (i - 1) / gw, xi = (i - 1) - yi * gw; idx = (yi - yj + gh - 1) * (2 * gw - 1) + (xi - xj + gw - 1); }
                v += table[(int64_t)idx * heads + head];
            }
            S[row * pitch + j] = v;
        }
    }
    __syncthreads();
    // ---- 2. softmax over the keys: wave w owns rows (QT/4) w .. (QT/4) w + QT/4 - 1 ---------------------------------------------------
    for (int rr = 0; rr < QT / 4; ++rr) {
        float *row = S + ((QT / 4) * wave + rr) * pitch;
        float m = -3.0e38f;
        for (int j = lane; j < Npad; j += 64) m = fmaxf(m, row[j]);
        m = wave_max(m);
        float s = 0.0f;
        for (int j = lane; j < Npad; j += 64) { const float e = j < N ? expf(row[j] - m) : 0.0f; row[j] = e; s += e; }
        const float inv = 1.0f / wave_sum(s);
        for (int j = lane; j < Npad; j += 64) row[j] *= inv;
    }
    __syncthreads();
    // ---- 3. out = P V: wave -> (column tile ct of the head dimension, part kp of the key range); parts are added in order through LDS
    const int ct = wave % NCOL, kp = wave / NCOL;
    const int nkb = Npad / 8, kb0 = (int)((int64_t)kp * nkb / KPARTS), kb1 = (int)((int64_t)(kp + 1) * nkb / KPARTS);
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
    const float *Vc = V + 32 * ct + li;
    for (int kb = kb0; kb < kb1; ++kb) {
        const float4 pf = *reinterpret_cast<const float4 *>(S + qrow * pitch + 8 * kb + 4 * lh);
        const int k0 = 8 * kb + 4 * lh;
        const float v0 = Vc[(int64_t)min(k0, N - 1) * ld], v1 = Vc[(int64_t)min(k0 + 1, N - 1) * ld];
        const float v2 = Vc[(int64_t)min(k0 + 2, N - 1) * ld], v3 = Vc[(int64_t)min(k0 + 3, N - 1) * ld];
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.x, v0, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.y, v1, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.z, v2, o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(pf.w, v3, o, 0, 0, 0);
    }
    __syncthreads();                                                    // everybody is done reading P: the score buffer becomes the exchange area
    float *X = S;                                                       // [KPARTS - 1][NCOL][32 rows][33]  (fits: QT * pitch >= 16 * 36 ... checked by the launcher)
    if (kp > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) X[(((kp - 1) * NCOL + ct) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = o[r];
    }
    __syncthreads();
    if (kp == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh, i = q0 + row;
            float v = o[r];
            for (int p = 1; p < KPARTS; ++p) v += X[(((p - 1) * NCOL + ct) * 32 + row) * 33 + li];
            if (row < QT && i < N) out[((int64_t)b * N + i) * out_ld + head * D + 32 * ct + li] = v;
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

template <int D, int QT> static int launch_attention_t(const float *qkv, int ld, float *out, int out_ld, int n, int N, int heads, const float *table,
                                                       int gh, int gw, hipStream_t st) {
    constexpr int kExchange = (4 / (D / 32) - 1) * (D / 32) * 32 * 33;        // floats of the partial-sum exchange area of phase 3
    size_t lds = (size_t)QT * (((N + 31) & ~31) + 4) * sizeof(float);
    if (lds < kExchange * sizeof(float)) lds = kExchange * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_attention<D, QT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k_attention<D, QT><<<dim3((unsigned)((N + QT - 1) / QT), (unsigned)heads, (unsigned)n), 256, lds, st>>>(qkv, ld, out, out_ld, N, heads, table, gh, gw);
    return check_launch("k_attention");
}

int launch_attention(const float *qkv, int ld, float *out, int out_ld, int n, int N, int heads, int d, const float *table, int gh, int gw,
                     hipStream_t st) {
    if (N < 1 || N > 2496) { set_error("attention: 1 <= tokens <= 2496 (the score tile of a block lives in LDS), got %d", N); return CSM_ERR_ARG; }
    if ((ld & 3) || (((uintptr_t)qkv) & 15)) { set_error("attention: qkv must be 16-byte aligned"); return CSM_ERR_ARG; }
    if (table && gh * gw + 1 != N) { set_error("attention: relative position bias needs N == gh * gw + 1 (%d x %d vs %d)", gh, gw, N); return CSM_ERR_ARG; }
    const bool wide = N <= 1216;
    switch (d) {
        case 32: return wide ? launch_attention_t<32, 32>(qkv, ld, out, out_ld, n, N, heads, table, gh, gw, st)
                             : launch_attention_t<32, 16>(qkv, ld, out, out_ld, n, N, heads, table, gh, gw, st);
        case 64: return wide ? launch_attention_t<64, 32>(qkv, ld, out, out_ld, n, N, heads, table, gh, gw, st)
                             : launch_attention_t<64, 16>(qkv, ld, out, out_ld, n, N, heads, table, gh, gw, st);
        case 128: return wide ? launch_attention_t<128, 32>(qkv, ld, out, out_ld, n, N, heads, table, gh, gw, st)
                              : launch_attention_t<128, 16>(qkv, ld, out, out_ld, n, N, heads, table, gh, gw, st);
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
