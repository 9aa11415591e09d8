// split_lab: can the bf16 matrix cores carry the fp32 GEMMs?  (diagnostic tool, not product)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/split_lab.hip -o gpurun_out/split_lab && gpurun_out/split_lab
// C[M,N] = A[M,K] . B[N,K]^T with fp32 operands and an fp32 result, two main loops of the same 128x128 / 4-wave structure:
//   F32 : v_mfma_f32_32x32x2_f32 (the product kernel's loop, csrc/igemm.hip)
//   S6  : every fp32 operand x is split EXACTLY into three bf16 terms x = h + m + l (round-to-nearest, 8 + 8 + 8 significand bits) while
//         it is staged to LDS; the product x*y is the six bf16 MFMA products hh' + (hm' + mh') + (hl' + mm' + lh') accumulated in fp32
//         (dropped: ml' + lm' + ll' <= 2^-25 |xy|, below half an fp32 ulp of the product).  v_mfma_f32_32x32x16_bf16 runs at 16x the
//         fp32-input rate, so six of them per K = 16 are 2.67x the fp32 MFMA peak.
// Reported per shape: time, fp32-equivalent TF/s (2MNK / t), and the relative L2 error of 256 result rows against an fp64 sum.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
    uint32_t q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// ---------------------------------------------------------------------------------------------------------------
template <int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_f32_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                             int M, int N, int K) {
    constexpr int BM = 128, BN = 128, BKT = 32, PITCH = BKT + 4, AI = 4;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t tiles_n = N / BN;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kc = tid & 7, ar = tid >> 3;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ra[AI], rb[AI];
    const int nk = K / BKT;
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(A + (int64_t)(m0 + ar + 32 * i) * K + kt * BKT + kc * 4);
            rb[i] = *reinterpret_cast<const f32x4*>(B + (int64_t)(n0 + ar + 32 * i) * K + kt * BKT + kc * 4);
        }
    };
    float* As = lds;
    float* Bs = As + BM * PITCH;
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            *reinterpret_cast<f32x4*>(&As[(ar + 32 * i) * PITCH + kc * 4]) = ra[i];
            *reinterpret_cast<f32x4*>(&Bs[(ar + 32 * i) * PITCH + kc * 4]) = rb[i];
        }
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BKT / 8; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f32x4*>(&As[((wm * 2 + i) * 32 + frow) * PITCH + kk * 8 + fk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * 2 + j) * 32 + frow) * PITCH + kk * 8 + fk]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + (wn * 2 + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(int64_t)row * N + col] = acc[i][j][r];
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// x = h + m + l, each a bf16 (round to nearest even); exact for every finite fp32 whose l does not underflow.
__device__ __forceinline__ void split3(const f32x4 x, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const __bf16 hh = (__bf16)x[e];
        const float r1 = x[e] - (float)hh;
        const __bf16 mm = (__bf16)r1;
        const float r2 = r1 - (float)mm;
        h[e] = hh;
        m[e] = mm;
        l[e] = (__bf16)r2;
    }
}

// NACC: 1 = one fp32 accumulator per output element for all six products; 2 = hh' in one, the five small products in a second
// (summed in the epilogue); PRODS: 6, or 3 (hh' + hm' + mh': "bf16x3", ~2^-17 per product - shown for contrast only)
// PBV: bytes per LDS row of one plane (planes apart), or, negative, bytes per row with the three planes of a row side by side
template <int MINW, int NACC, int PRODS, int ABL = 0, int PBV = 80>
__global__ __launch_bounds__(256, MINW) void gemm_split_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                               float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 128, BN = 128, BKT = 32, AI = 4;
    constexpr bool SWZ = PBV == 64;                    // 64-byte rows (no padding), 16-byte chunk index XOR ((row >> 1) + (row >> 3)) & 3
    constexpr int PB = PBV > 0 ? PBV : -PBV;           // bytes per LDS row
    constexpr int PSTEP = PBV > 0 ? BM * PB : 64;      // byte distance between the planes of one operand
    constexpr int PLANE = PBV > 0 ? BM * PB : BM * PB / 3;      // so that 3 * PLANE = one operand's bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t tiles_n = N / BN;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kc = tid & 7, ar = tid >> 3;
    f32x16 acc[NACC][2][2];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;
    f32x4 ra[AI], rb[AI];
    const int nk = K / BKT;
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(A + (int64_t)(m0 + ar + 32 * i) * K + kt * BKT + kc * 4);
            rb[i] = *reinterpret_cast<const f32x4*>(B + (int64_t)(n0 + ar + 32 * i) * K + kt * BKT + kc * 4);
        }
    };
    constexpr int NBUF = ABL == 4 ? 2 : 1;
    auto store_tile = [&](int buf) {
        unsigned char* const As = ldsb + buf * 6 * PLANE;                    // planes h, m, l of A, then of B
        unsigned char* const Bs = As + 3 * PLANE;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            bf16x4 h, m, l;
            const int srow = ar + 32 * i;
            const int off = SWZ ? srow * 64 + ((((kc >> 1) ^ ((srow >> 1) + (srow >> 3))) & 3) << 4) + (kc & 1) * 8 : srow * PB + kc * 8;
            if (ABL == 1) { for (int e = 0; e < 4; ++e) h[e] = (__bf16)ra[i][e]; m = h; l = h; } else split3(ra[i], h, m, l);
            *reinterpret_cast<bf16x4*>(As + off) = h;
            *reinterpret_cast<bf16x4*>(As + PSTEP + off) = m;
            if (PRODS == 6) *reinterpret_cast<bf16x4*>(As + 2 * PSTEP + off) = l;
            if (ABL == 1) { for (int e = 0; e < 4; ++e) h[e] = (__bf16)rb[i][e]; m = h; l = h; } else split3(rb[i], h, m, l);
            *reinterpret_cast<bf16x4*>(Bs + off) = h;
            *reinterpret_cast<bf16x4*>(Bs + PSTEP + off) = m;
            if (PRODS == 6) *reinterpret_cast<bf16x4*>(Bs + 2 * PSTEP + off) = l;
        }
    };
    const int frow = lane & 31, fk = (lane >> 5) * 16;       // byte offset of this lane's 8 bf16 inside a 16-deep k slab
    constexpr int NP = PRODS == 6 ? 3 : 2;
    auto compute = [&](int buf) {
        const unsigned char* const As = ldsb + buf * 6 * PLANE;
        const unsigned char* const Bs = As + 3 * PLANE;
#pragma unroll
        for (int kk = 0; kk < BKT / 16; ++kk) {
            bf16x8 a[2][NP], b[2][NP];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    a[i][p] = *reinterpret_cast<const bf16x8*>(As + p * PSTEP + (SWZ ? ((wm * 2 + i) * 32 + frow) * 64 + ((((kk * 2 + (lane >> 5)) ^ ((frow >> 1) + (frow >> 3))) & 3) << 4)
                                                                                      : ((wm * 2 + i) * 32 + frow) * PB + kk * 32 + fk));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    b[j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * PSTEP + (SWZ ? ((wn * 2 + j) * 32 + frow) * 64 + ((((kk * 2 + (lane >> 5)) ^ ((frow >> 1) + (frow >> 3))) & 3) << 4)
                                                                                      : ((wn * 2 + j) * 32 + frow) * PB + kk * 32 + fk));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16& s = acc[NACC - 1][i][j];
                    if constexpr (PRODS == 6) {
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], s, 0, 0, 0);
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], s, 0, 0, 0);
                        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], s, 0, 0, 0);
                    }
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], s, 0, 0, 0);
                    acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[0][i][j], 0, 0, 0);
                }
        }
    };
    if constexpr (ABL == 3) {
        bf16x8 a0, b0;
        for (int e = 0; e < 8; ++e) { a0[e] = (__bf16)A[tid + e]; b0[e] = (__bf16)B[tid + e]; }
        for (int kt = 0; kt < nk; ++kt) {
#pragma unroll
            for (int s = 0; s < 12; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[(s & 1) * (NACC - 1)][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[(s & 1) * (NACC - 1)][i][j], 0, 0, 0);
            asm volatile("" : "+v"(a0), "+v"(b0));
        }
    } else if constexpr (ABL == 4) {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tile(kt + 1);
            compute(kt & 1);
            if (kt + 1 < nk) store_tile((kt + 1) & 1);
            __syncthreads();
        }
    } else {
        load_tile(0);
        if (ABL == 2) { store_tile(0); __syncthreads(); }
        for (int kt = 0; kt < nk; ++kt) {
            if (ABL != 2) { store_tile(0); __syncthreads(); }
            if (kt + 1 < nk) load_tile(kt + 1);
            compute(0);
            if (ABL != 2) __syncthreads();
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + (wn * 2 + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[0][i][j][r];
                if (NACC == 2) v += acc[1][i][j][r];
                C[(int64_t)row * N + col] = v;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// P: K tiles of 16, LDS double-buffered, ONE barrier per tile, and the staging of tile k+1 (split + LDS stores) issued between the MFMA
// groups of tile k by the same wave (an MFMA occupies the issue port for 8 of its 32 cycles; the split work rides in the other 24).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(const float x0, const float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}
template <int MINW, bool PIN>
__global__ __launch_bounds__(256, MINW) void gemm_pipe_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                              int M, int N, int K) {
    constexpr int BM = 128, BN = 128, BKT = 16;
    constexpr int PB = 48;                             // 16 bf16 + 16 B pad: 12 dwords, 8 rows cover all 32 banks for the b128 reads
    constexpr int PLANE = 128 * PB, BUF = 6 * PLANE;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t tiles_n = N / BN;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kc = tid & 3, ar = tid >> 2;             // chunk of 4 k, rows ar and ar + 64
    f32x16 acc[2][2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;
    const int nk = K / BKT;
    f32x4 g[2][4];                                     // [set][A0, A1, B0, B1]
    auto load_tile = [&](int kt, f32x4 (&r)[4]) {
        r[0] = *reinterpret_cast<const f32x4*>(A + (int64_t)(m0 + ar) * K + kt * BKT + kc * 4);
        r[1] = *reinterpret_cast<const f32x4*>(A + (int64_t)(m0 + ar + 64) * K + kt * BKT + kc * 4);
        r[2] = *reinterpret_cast<const f32x4*>(B + (int64_t)(n0 + ar) * K + kt * BKT + kc * 4);
        r[3] = *reinterpret_cast<const f32x4*>(B + (int64_t)(n0 + ar + 64) * K + kt * BKT + kc * 4);
    };
    auto stage_chunk = [&](int buf, int c, const f32x4 v) {      // c: 0, 1 = A rows ar, ar + 64; 2, 3 = B
        unsigned char* base = ldsb + buf * BUF + (c >> 1) * 3 * PLANE + (ar + 64 * (c & 1)) * PB + kc * 8;
        uint32_t h0, m0_, l0, h1, m1, l1;
        split3_pair(v[0], v[1], h0, m0_, l0);
        split3_pair(v[2], v[3], h1, m1, l1);
        *reinterpret_cast<u32x2*>(base) = u32x2{h0, h1};
        *reinterpret_cast<u32x2*>(base + PLANE) = u32x2{m0_, m1};
        *reinterpret_cast<u32x2*>(base + 2 * PLANE) = u32x2{l0, l1};
    };
    const int frow = lane & 31, fk = (lane >> 5) * 16;
    auto step = [&](int kt, f32x4 (&cur)[4], f32x4 (&nxt)[4]) {
        // cur: registers of tile kt + 1 (loaded one iteration ago), nxt: receives tile kt + 2
        const int buf = kt & 1;
        if (kt + 2 < nk) load_tile(kt + 2, nxt);
        const unsigned char* As = ldsb + buf * BUF;
        const unsigned char* Bs = As + 3 * PLANE;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(As + p * PLANE + ((wm * 2 + i) * 32 + frow) * PB + fk);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(Bs + p * PLANE + ((wn * 2 + j) * 32 + frow) * PB + fk);
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            const int i = ij >> 1, j = ij & 1;
            if (kt + 1 < nk) stage_chunk(buf ^ 1, ij, cur[ij]);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
            f32x16& s = acc[1][i][j];
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], s, 0, 0, 0);
            acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[0][i][j], 0, 0, 0);
            if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    load_tile(0, g[0]);
#pragma unroll
    for (int c = 0; c < 4; ++c) stage_chunk(0, c, g[0][c]);
    if (nk > 1) load_tile(1, g[1]);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {               // nk even (K multiple of 32)
        step(kt, g[1], g[0]);
        step(kt + 1, g[0], g[1]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + (wn * 2 + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(int64_t)row * N + col] = acc[0][i][j][r] + acc[1][i][j][r];
            }
    }
}

// full 24-bit significands, magnitudes over ~3 binades, both signs
__global__ void fill_kernel(float* p, int64_t n, uint32_t seed, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        uint32_t y = x * 747796405u + 2891336453u;
        y ^= y >> 17;
        const float u = ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f) - 0.5f;         // (-0.5, 0.5), 24 bits
        const float v = 0.25f + (float)(y >> 8) * (1.0f / 16777216.0f);                  // [0.25, 1.25)
        p[i] = scale < 0.f ? fabsf(u) * v * v * v * -scale : u * v * v * v * scale;
    }
}

// fp64 truth for the first R rows
__global__ void ref_kernel(const float* __restrict__ A, const float* __restrict__ B, double* __restrict__ C, int R, int N, int K) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)R * N) return;
    const int r = (int)(idx / N), c = (int)(idx % N);
    double s = 0;
    for (int k = 0; k < K; ++k) s += (double)A[(int64_t)r * K + k] * (double)B[(int64_t)c * K + k];
    C[idx] = s;
}
__global__ void err_kernel(const float* __restrict__ C, const double* __restrict__ Ref, double* __restrict__ out, int R, int N) {
    // one block: sum (c - ref)^2 and sum ref^2
    __shared__ double s1[256], s2[256], s3[256], s4[256];
    double a = 0, b = 0, c = 0, e = 0;
    for (int64_t i = threadIdx.x; i < (int64_t)R * N; i += 256) {
        const double d = (double)C[i] - Ref[i];
        a += d * d;
        b += Ref[i] * Ref[i];
        c += d;                      // signed: a systematic bias of the accumulation shows here
        e += fabs(Ref[i]);
    }
    s1[threadIdx.x] = a; s2[threadIdx.x] = b; s3[threadIdx.x] = c; s4[threadIdx.x] = e;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { s1[threadIdx.x] += s1[threadIdx.x + s]; s2[threadIdx.x] += s2[threadIdx.x + s]; s3[threadIdx.x] += s3[threadIdx.x + s]; s4[threadIdx.x] += s4[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = s1[0]; out[1] = s2[0]; out[2] = s3[0]; out[3] = s4[0]; }
}

struct Shape { int M, N, K; };

template <typename F>
static float time_it(F f, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / reps;
}

int main() {
    Shape shapes[] = {{4096, 4096, 4096}, {25088, 256, 2304}, {6272, 512, 4608}, {100352, 128, 1152}, {25088, 1024, 256}, {401408, 256, 64},
                      {-4096, 256, 2304}, {-4096, 256, 256}};      // M < 0: all-positive operands (every product has the same sign: accumulation bias shows)
    const int R = 256;
    for (auto& sh0 : shapes) {
        Shape sh = sh0;
        const bool positive = sh.M < 0;
        if (positive) sh.M = -sh.M;
        const int M = sh.M, N = sh.N, K = sh.K;
        float *A, *B, *C;
        double *Ref, *err;
        CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&B, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
        CK(hipMalloc(&Ref, (size_t)R * N * 8)); CK(hipMalloc(&err, 32));
        fill_kernel<<<2048, 256>>>(A, (int64_t)M * K, 1u, positive ? -4.0f : 4.0f);
        fill_kernel<<<2048, 256>>>(B, (int64_t)N * K, 7u, positive ? -0.25f : 0.25f);
        ref_kernel<<<(unsigned)(((int64_t)R * N + 255) / 256), 256>>>(A, B, Ref, R, N, K);
        CK(hipDeviceSynchronize());
        const unsigned tiles = (unsigned)(M / 128) * (N / 128);
        const double fl = 2.0 * M * N * K;
        printf("shape M %d N %d K %d  tiles %u%s\n", M, N, K, tiles, positive ? "  (all-positive operands)" : "");
        auto report = [&](const char* name, float ms) {
            double h[4];
            err_kernel<<<1, 256>>>(C, Ref, err, R, N);
            CK(hipMemcpy(h, err, 32, hipMemcpyDeviceToHost));
            printf("  %-58s %8.3f ms  %6.1f TF/s (fp32-equivalent)   rel L2 error vs fp64 %.3e   signed mean error / mean |ref| %+.3e\n", name, ms, fl / ms / 1e9, sqrt(h[0] / h[1]), h[2] / h[3]);
            fflush(stdout);
        };
#define RUN(NAME, KERNEL, LDSB)                                                                          \
    {                                                                                                    \
        CK(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB)); \
        CK(hipMemset(C, 0, (size_t)M * N * 4));                                                          \
        float ms = time_it([&] { hipLaunchKernelGGL(KERNEL, dim3(tiles), dim3(256), LDSB, 0, A, B, C, M, N, K); }, 10); \
        CK(hipGetLastError());                                                                           \
        report(NAME, ms);                                                                                \
    }
        RUN("F32  v_mfma_f32_32x32x2_f32, launch_bounds(256,2)", (gemm_f32_kernel<2>), 256 * 36 * 4);
        RUN("S6   six bf16 products, one accumulator", (gemm_split_kernel<2, 1, 6>), 6 * 128 * 80);
        RUN("S6   six bf16 products, one accumulator, (256,3)", (gemm_split_kernel<3, 1, 6>), 6 * 128 * 80);
        RUN("S6x2 six bf16 products, hh' apart from the small five", (gemm_split_kernel<2, 2, 6>), 6 * 128 * 80);
        RUN("S3   three bf16 products (contrast: not fp32-grade)", (gemm_split_kernel<2, 1, 3>), 6 * 128 * 80);
        RUN("S6x2 ablation: convert only, no split arithmetic", (gemm_split_kernel<2, 2, 6, 1>), 6 * 128 * 80);
        RUN("S6x2 ablation: no LDS stores / barriers in the loop", (gemm_split_kernel<2, 2, 6, 2>), 6 * 128 * 80);
        RUN("S6x2 ablation: MFMA only", (gemm_split_kernel<2, 2, 6, 3>), 6 * 128 * 80);
        RUN("S6x2 double-buffered LDS, one barrier per tile (1 WG/CU)", (gemm_split_kernel<1, 2, 6, 4>), 2 * 6 * 128 * 80);
        RUN("S6x2 64-byte rows, chunk index swizzled by the row", (gemm_split_kernel<2, 2, 6, 0, 64>), 6 * 128 * 64);
        RUN("S6x2 64-byte rows swizzled, (256,3)", (gemm_split_kernel<3, 2, 6, 0, 64>), 6 * 128 * 64);
        RUN("S6x2 row pitch  96 B", (gemm_split_kernel<2, 2, 6, 0, 96>), 6 * 128 * 96);
        RUN("S6x2 row pitch 112 B", (gemm_split_kernel<2, 2, 6, 0, 112>), 6 * 128 * 112);
        RUN("S6x2 row pitch 144 B (1 WG/CU by LDS)", (gemm_split_kernel<2, 2, 6, 0, 144>), 6 * 128 * 144);
        RUN("P    K tiles of 16, double-buffered, staging between MFMA groups (pinned)", (gemm_pipe_kernel<2, true>), 2 * 6 * 128 * 48);
        RUN("P    same, compiler's own order", (gemm_pipe_kernel<2, false>), 2 * 6 * 128 * 48);
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(Ref)); CK(hipFree(err));
    }
    return 0;
}
