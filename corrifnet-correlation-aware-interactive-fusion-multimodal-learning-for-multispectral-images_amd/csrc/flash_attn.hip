// Flash-style multi-head self-attention for the CorrIFNet transformers (mmvit4.py:305-315), fp32 on the gfx950 matrix cores.
//
//   O = dropout(softmax(Q K^T * scale)) V        head dim 64, N in {512, 2048} (any multiple of 128)
//
// The [B, heads, N, N] score / probability tensors are never written: forward keeps one 64-key tile of S^T in MFMA accumulators,
// backward recomputes it from Q, K and the saved row log-sum-exp.  All products are v_mfma_f32_32x32x2_f32 (exact fp32 fma chains).
//
// Layout trick (no LDS round trip for P).  The 32x32 MFMA gives lane (c = lane & 31, h = lane >> 5) the 16 values
// (row = (r & 3) + 8 (r >> 2) + 4 h, column c) of the result, and takes as B operand B[k = h][column c] from the same lane.  So a
// score tile whose COLUMNS are the queries (S^T = K Q^T) can be fed straight back as the B operand of O^T += V^T P^T: the MFMA
// (g, e) contracts over the key pair {8 g + e, 8 g + 4 + e}, lane (c, h) supplies its own register r = 4 g + e for it and the A
// operand reads V[8 g + 4 h + e][d] from LDS - the k order only has to agree between A and B.  The same holds for dQ^T += K^T dS^T
// (columns = queries) and, with the roles swapped (columns = keys, S = Q K^T), for dV^T += dO^T P and dK^T += Q^T dS.
//
// Dropout mask = the Philox4x32-10 stream of corrif_dropout over the flat [B, heads, N, N] index (counter = offset/4 + index/4,
// element = index % 4), so the fused kernels reproduce the unfused softmax + dropout passes bit for bit in the mask.  With queries
// as columns the four consecutive keys of one counter are the lane's own registers 4 g .. 4 g + 3.  The forward draws the mask once
// and leaves it as keep BITS ([B*heads][N][N/32] words, 1/32 of a score tensor); the backward kernels read the bits.
//
// Deterministic: no atomics; dQ and (dK, dV) come from two kernels that each own their output rows.
#include "common.h"

namespace corrif_flash {       // named (not anonymous): the kernels keep a greppable name in rocprofv3 traces

constexpr int HD = 64;             // head dimension
constexpr int TP = HD + 4;         // LDS row pitch (floats): 272-byte rows keep float4 fragment reads conflict-free
constexpr int KT = 64;             // rows of the streamed operand tile (keys in fwd / dQ, queries in dK-dV)

__device__ __forceinline__ void philox_round_(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox_(uint64_t ctr, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round_(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = c[e];
}
__device__ __forceinline__ float keep_factor(uint32_t word, float pdrop, float inv_keep) {
    const float u = (float)(word >> 8) * (1.0f / 16777216.0f);      // the 24-bit uniform of corrif_dropout
    return u >= pdrop ? inv_keep : 0.f;
}

struct FlashArgs {
    const float* qkv;     // [B][N][3*C], q | k | v, head h at columns h*64
    float* out;           // fwd: O [B][N][C]
    float* lse;           // [B*heads][N]  row log-sum-exp of the scaled scores
    const float* dout;    // bwd: dO [B][N][C]
    const float* dvec;    // bwd: D = rowsum(dO * O)  [B*heads][N]
    float* dqkv;          // bwd: [B][N][3*C]
    uint32_t* mask;       // dropout keep bits [B*heads][N][N/32] (bit j of word kb <-> key 32 kb + j), written by fwd, read by bwd
    int N, heads, C;
    float scale, pdrop, inv_keep;
    uint64_t seed, offset4;
};

// ---- cooperative tile load: rows [r0, r0 + 64) x 64 floats of a [.. ][pitch] matrix -> LDS [64][TP]
__device__ __forceinline__ void load_tile_regs(const float* __restrict__ src, int64_t pitch, int tid, f32x4 (&v)[4]) {
    const int c4 = tid & 15, r = tid >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(src + (int64_t)(r + 16 * i) * pitch + c4 * 4);
}
__device__ __forceinline__ void store_tile_lds(float* lds, int tid, const f32x4 (&v)[4]) {
    const int c4 = tid & 15, r = tid >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(lds + (r + 16 * i) * TP + c4 * 4) = v[i];
}

// acc += tile[rows 32 t .. 32 t + 31][0..63] . frag^T (one 32-row block of the LDS tile)
__device__ __forceinline__ void block_times_resident(const float* tile, int t, const f32x4 (&frag)[8], int lane, f32x16& acc) {
    const int i = lane & 31, h4 = (lane >> 5) * 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(tile + (32 * t + i) * TP + s * 8 + h4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], frag[s][e], acc, 0, 0, 0);
    }
}
// out[dt] += (tile rows 32 t ..)^T . w : contraction over that 32-row block, w[r] = this lane's register for row 32 t + 8 (r >> 2) + 4 h + (r & 3)
__device__ __forceinline__ void blockT_times_regs(const float* tile, int t, const f32x16& w, int lane, f32x16 (&out)[2]) {
    const int i = lane & 31, h4 = (lane >> 5) * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * t + 8 * (r >> 2) + h4 + (r & 3);
        const float a0 = tile[row * TP + i], a1 = tile[row * TP + 32 + i];
        out[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w[r], out[0], 0, 0, 0);
        out[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w[r], out[1], 0, 0, 0);
    }
}
__device__ __forceinline__ void zero1(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}
__device__ __forceinline__ void zero2(f32x16 (&a)[2]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[t][r] = 0.f;
}
// resident fragment of row `row` (64 floats at p): float4 chunks 8 s + 4 h
__device__ __forceinline__ void load_resident(const float* __restrict__ p, int lane, f32x4 (&f)[8]) {
    const int h4 = (lane >> 5) * 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = *reinterpret_cast<const f32x4*>(p + s * 8 + h4);
}
// store the transposed accumulator pair (rows = head dim, column = this lane's token) as 8 float4 at p[0..63]
__device__ __forceinline__ void store_cols(float* __restrict__ p, int lane, const f32x16 (&a)[2], float mul) {
    const int h4 = (lane >> 5) * 4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {a[dt][4 * g] * mul, a[dt][4 * g + 1] * mul, a[dt][4 * g + 2] * mul, a[dt][4 * g + 3] * mul};
            *reinterpret_cast<f32x4*>(p + 32 * dt + 8 * g + h4) = v;
        }
}

// =============================================================================================== forward
// grid (N / 128, B * heads); wave w owns queries q0 + 32 w ..; the workgroup streams 64-key K / V tiles through LDS.
// (the dropout variant would need one spilled register to fit three waves per SIMD: scratch memory has to be set up on the queue
// at the first launch, which showed up as a 18 ms outlier - two waves, no scratch)
template <bool DROP>
__global__ __launch_bounds__(256, DROP ? 2 : 3) void flash_fwd_kernel(FlashArgs p) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * TP];
    __shared__ __attribute__((aligned(16))) float Vs[KT * TP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, b = z / p.heads, hh = z - b * p.heads;
    const int64_t ld = 3 * (int64_t)p.C;
    const float* __restrict__ base = p.qkv + (int64_t)b * p.N * ld + hh * HD;
    const int n = blockIdx.x * 128 + wave * 32 + (lane & 31);          // this lane's query (column)
    const int h4 = (lane >> 5) * 4;

    f32x4 qf[8];
    load_resident(base + (int64_t)n * ld, lane, qf);
    f32x16 o[2];
    zero2(o);
    float m_run = -INFINITY, l_run = 0.f;       // l_run: this half's partial row sum (halves are combined at the end)
    const uint64_t row4 = ((uint64_t)z * p.N + n) * (uint64_t)(p.N / 4);        // counter of (query n, key 0)

    f32x4 kr[4], vr[4];
    load_tile_regs(base + p.C, ld, tid, kr);
    load_tile_regs(base + 2 * p.C, ld, tid, vr);
    for (int k0 = 0; k0 < p.N; k0 += KT) {
        store_tile_lds(Ks, tid, kr);
        store_tile_lds(Vs, tid, vr);
        __syncthreads();
        if (k0 + KT < p.N) {
            load_tile_regs(base + (int64_t)(k0 + KT) * ld + p.C, ld, tid, kr);
            load_tile_regs(base + (int64_t)(k0 + KT) * ld + 2 * p.C, ld, tid, vr);
        }
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {                  // online softmax per 32-key block: one score accumulator live at a time
            f32x16 s;
            zero1(s);
            block_times_resident(Ks, t, qf, lane, s);  // S^T block: rows = keys, column = query n
            float mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] *= p.scale; mx = fmaxf(mx, s[r]); }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = expf(m_run - m_new);   // first block: exp(-inf) = 0
            m_run = m_new;
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = expf(s[r] - m_new); sum += s[r]; }
            l_run = l_run * alpha + sum;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            if constexpr (DROP) {
                uint32_t bits = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t w[4];
                    philox_(p.offset4 + row4 + (uint64_t)((k0 + 32 * t + 8 * g + h4) >> 2), p.seed, w);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float kf = keep_factor(w[e], p.pdrop, p.inv_keep);
                        s[4 * g + e] *= kf;
                        bits |= (kf != 0.f ? 1u : 0u) << (8 * g + h4 + e);
                    }
                }
                // the mask is drawn once: the backward kernels read these bits instead of re-running Philox (10 rounds per 4 elements,
                // ~0.7 ms per pass over the [32, 8, 2048, 2048] index space)
                bits |= (uint32_t)__shfl_xor((int)bits, 32);
                if (lane < 32) p.mask[((int64_t)z * p.N + n) * (p.N >> 5) + ((k0 >> 5) + t)] = bits;
            }
            blockT_times_regs(Vs, t, s, lane, o);      // O^T += V^T P'^T
        }
        __syncthreads();
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    store_cols(p.out + ((int64_t)b * p.N + n) * p.C + hh * HD, lane, o, 1.0f / l_tot);
    if (lane < 32) p.lse[(int64_t)z * p.N + n] = m_run + logf(l_tot);
}

// =============================================================================================== D = rowsum(dO * O)
__global__ __launch_bounds__(256) void flash_dvec_kernel(const float* __restrict__ o, const float* __restrict__ go, float* __restrict__ dvec,
                                                         int B, int N, int heads) {
    // one 16-lane group per (b, n, head): 64 floats = 16 float4
    const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int c4 = threadIdx.x & 15;
    const int64_t total = (int64_t)B * N * heads;
    float v = 0.f;
    if (gid < total) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(o + gid * HD + c4 * 4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(go + gid * HD + c4 * 4);
        v = a[0] * g[0] + a[1] * g[1] + a[2] * g[2] + a[3] * g[3];
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (gid < total && c4 == 0) {
        const int64_t bn = gid / heads;
        const int hh = (int)(gid - bn * heads);
        const int64_t b = bn / N, n = bn - b * N;
        dvec[((int64_t)b * heads + hh) * N + n] = v;
    }
}

// =============================================================================================== backward: dQ
// grid (N / 128, B * heads); wave owns 32 queries (columns), streams 64-key K / V tiles.
template <bool DROP>
__global__ __launch_bounds__(256, 3) void flash_bwd_dq_kernel(FlashArgs p) {
    __shared__ __attribute__((aligned(16))) float Ks[KT * TP];
    __shared__ __attribute__((aligned(16))) float Vs[KT * TP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, b = z / p.heads, hh = z - b * p.heads;
    const int64_t ld = 3 * (int64_t)p.C;
    const float* __restrict__ base = p.qkv + (int64_t)b * p.N * ld + hh * HD;
    const int n = blockIdx.x * 128 + wave * 32 + (lane & 31);
    const int h4 = (lane >> 5) * 4;

    f32x4 qf[8], gf[8];
    load_resident(base + (int64_t)n * ld, lane, qf);
    load_resident(p.dout + ((int64_t)b * p.N + n) * p.C + hh * HD, lane, gf);
    const float lse = p.lse[(int64_t)z * p.N + n], dv = p.dvec[(int64_t)z * p.N + n];
    f32x16 dq[2];
    zero2(dq);
    for (int k0 = 0; k0 < p.N; k0 += KT) {
        f32x4 kr[4], vr[4];
        load_tile_regs(base + (int64_t)k0 * ld + p.C, ld, tid, kr);
        load_tile_regs(base + (int64_t)k0 * ld + 2 * p.C, ld, tid, vr);
        store_tile_lds(Ks, tid, kr);
        store_tile_lds(Vs, tid, vr);
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            f32x16 s, dp;
            zero1(s);
            zero1(dp);
            uint32_t bits = 0;
            if constexpr (DROP) bits = p.mask[((int64_t)z * p.N + n) * (p.N >> 5) + ((k0 >> 5) + t)];
            block_times_resident(Ks, t, qf, lane, s);      // S^T
            block_times_resident(Vs, t, gf, lane, dp);     // dP'^T = V dO^T
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float pr = expf(s[r] * p.scale - lse);
                    float d = dp[r];
                    if constexpr (DROP) d = (bits >> (8 * g + h4 + e)) & 1u ? d * p.inv_keep : 0.f;
                    s[r] = p.scale * pr * (d - dv);             // dS^T
                }
            }
            blockT_times_regs(Ks, t, s, lane, dq);         // dQ^T += K^T dS^T
        }
        __syncthreads();
    }
    store_cols(p.dqkv + ((int64_t)b * p.N + n) * ld + hh * HD, lane, dq, 1.0f);
}

// =============================================================================================== backward: dK, dV
// grid (N / 128, B * heads); wave owns 32 keys (columns), streams 64-query Q / dO tiles (+ their lse / D entries).
template <bool DROP>
__global__ __launch_bounds__(256, 2) void flash_bwd_dkv_kernel(FlashArgs p) {
    __shared__ __attribute__((aligned(16))) float Qs[KT * TP];
    __shared__ __attribute__((aligned(16))) float Gs[KT * TP];
    __shared__ __attribute__((aligned(16))) float Ls[KT];
    __shared__ __attribute__((aligned(16))) float Ds[KT];
    __shared__ __attribute__((aligned(16))) uint32_t Ms[4 * KT];       // keep bits of (query row, this wave's 32-key block)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.y, b = z / p.heads, hh = z - b * p.heads;
    const int64_t ld = 3 * (int64_t)p.C;
    const float* __restrict__ base = p.qkv + (int64_t)b * p.N * ld + hh * HD;
    const float* __restrict__ gbase = p.dout + (int64_t)b * p.N * p.C + hh * HD;
    const int m = blockIdx.x * 128 + wave * 32 + (lane & 31);          // this lane's key (column)
    const int h4 = (lane >> 5) * 4;

    f32x4 kf[8], vf[8];
    load_resident(base + (int64_t)m * ld + p.C, lane, kf);
    load_resident(base + (int64_t)m * ld + 2 * p.C, lane, vf);
    f32x16 dk[2], dvv[2];
    zero2(dk);
    zero2(dvv);
    const int kb = blockIdx.x * 4 + wave, bit = lane & 31;             // this wave's 32-key block / this lane's bit in its words
    for (int q0 = 0; q0 < p.N; q0 += KT) {
        f32x4 qr[4], gr[4];
        load_tile_regs(base + (int64_t)q0 * ld, ld, tid, qr);
        load_tile_regs(gbase + (int64_t)q0 * p.C, p.C, tid, gr);
        store_tile_lds(Qs, tid, qr);
        store_tile_lds(Gs, tid, gr);
        if (tid < KT) {
            Ls[tid] = p.lse[(int64_t)z * p.N + q0 + tid];
            Ds[tid] = p.dvec[(int64_t)z * p.N + q0 + tid];
        }
        if constexpr (DROP) Ms[tid] = p.mask[((int64_t)z * p.N + q0 + lane) * (p.N >> 5) + kb];      // thread (wave, lane): row q0 + lane
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < 2; ++t) {
            f32x16 s, dp;
            zero1(s);
            zero1(dp);
            block_times_resident(Qs, t, kf, lane, s);      // S: rows = queries, column = key m
            block_times_resident(Gs, t, vf, lane, dp);     // dP' = dO V^T
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rbase = 32 * t + 8 * g + h4;                 // tile rows rbase .. rbase + 3 <-> registers 4 g .. 4 g + 3
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[rbase]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&Ds[rbase]);
                uint32_t w[4] = {0, 0, 0, 0};
                if constexpr (DROP) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = Ms[wave * KT + rbase + e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    float kf_ = 1.f;
                    if constexpr (DROP) kf_ = (w[e] >> bit) & 1u ? p.inv_keep : 0.f;
                    const float pr = expf(s[r] * p.scale - l4[e]);
                    const float d = dp[r] * kf_;
                    dp[r] = pr * kf_;                                   // P'
                    s[r] = p.scale * pr * (d - d4[e]);                  // dS
                }
            }
            blockT_times_regs(Gs, t, dp, lane, dvv);       // dV^T += dO^T P'
            blockT_times_regs(Qs, t, s, lane, dk);         // dK^T += Q^T dS
        }
        __syncthreads();
    }
    store_cols(p.dqkv + ((int64_t)b * p.N + m) * ld + p.C + hh * HD, lane, dk, 1.0f);
    store_cols(p.dqkv + ((int64_t)b * p.N + m) * ld + 2 * p.C + hh * HD, lane, dvv, 1.0f);
}

bool flash_ok(const void* qkv, int32_t B, int32_t N, int32_t heads, float pdrop, uint64_t offset) {
    if (!qkv || B <= 0 || heads <= 0 || N <= 0 || !(pdrop >= 0.f && pdrop < 1.f)) return false;
    if ((int64_t)B * heads > 65535) return false;
    if (N % 128 || (offset & 3) || ((uintptr_t)qkv & 15)) return false;
    return true;
}

}  // namespace corrif_flash
using namespace corrif_flash;

extern "C" int corrif_flash_attn_supported(int32_t N, int32_t head_dim) { return head_dim == HD && N > 0 && N % 128 == 0; }

extern "C" size_t corrif_flash_attn_mask_bytes(int32_t B, int32_t N, int32_t heads, float pdrop) {
    if (!(pdrop > 0.f) || B <= 0 || N <= 0 || heads <= 0) return 0;
    return (size_t)B * heads * N * (N / 32) * sizeof(uint32_t);
}

extern "C" int corrif_flash_attn_fwd(const float* qkv, float* out, float* lse, uint32_t* mask, int32_t B, int32_t N, int32_t heads, float scale,
                                     float pdrop, uint64_t seed, uint64_t offset, void* stream) {
    if (!flash_ok(qkv, B, N, heads, pdrop, offset) || !out || !lse || (pdrop > 0.f && !mask)) return CORRIF_EINVAL;
    if (((uintptr_t)out & 15)) return CORRIF_EUNSUPPORTED;
    FlashArgs a{};
    a.qkv = qkv; a.out = out; a.lse = lse; a.mask = mask; a.N = N; a.heads = heads; a.C = heads * HD;
    a.scale = scale; a.pdrop = pdrop; a.inv_keep = 1.0f / (1.0f - pdrop); a.seed = seed; a.offset4 = offset / 4;
    dim3 grid(N / 128, B * heads);
    if (pdrop > 0.f) hipLaunchKernelGGL(flash_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(flash_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

extern "C" int corrif_flash_attn_bwd(const float* qkv, const float* out, const float* lse, const uint32_t* mask, const float* dout, float* dvec,
                                     float* dqkv, int32_t B, int32_t N, int32_t heads, float scale, float pdrop, void* stream) {
    if (!flash_ok(qkv, B, N, heads, pdrop, 0) || !out || !lse || !dout || !dvec || !dqkv || (pdrop > 0.f && !mask)) return CORRIF_EINVAL;
    if (((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15)) return CORRIF_EUNSUPPORTED;
    FlashArgs a{};
    a.qkv = qkv; a.lse = const_cast<float*>(lse); a.mask = const_cast<uint32_t*>(mask); a.dout = dout; a.dvec = dvec; a.dqkv = dqkv;
    a.N = N; a.heads = heads; a.C = heads * HD;
    a.scale = scale; a.pdrop = pdrop; a.inv_keep = 1.0f / (1.0f - pdrop);
    hipStream_t s = (hipStream_t)stream;
    const int64_t groups = (int64_t)B * N * heads;
    hipLaunchKernelGGL(flash_dvec_kernel, dim3((unsigned)((groups * 16 + 255) / 256)), dim3(256), 0, s, out, dout, dvec, B, N, heads);
    CORRIF_CHECK_LAUNCH();
    dim3 grid(N / 128, B * heads);
    if (pdrop > 0.f) {
        hipLaunchKernelGGL(flash_bwd_dkv_kernel<true>, grid, dim3(256), 0, s, a);
        CORRIF_CHECK_LAUNCH();
        hipLaunchKernelGGL(flash_bwd_dq_kernel<true>, grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(flash_bwd_dkv_kernel<false>, grid, dim3(256), 0, s, a);
        CORRIF_CHECK_LAUNCH();
        hipLaunchKernelGGL(flash_bwd_dq_kernel<false>, grid, dim3(256), 0, s, a);
    }
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
