// Weight gradients (contraction over rows) of the implicit-GEMM family: fp32-input MFMA kernels, their split-bf16 twins, the slab
// reduction and the host side of corrif_wgrad.  (Split off igemm.hip so that it compiles in parallel with the forward family.)
#include "igemm_fwd.h"


// ------------------------------------------------------------------------------------------------
// W-type: contraction over rows.  LDS tiles are [32 rows][BM] and [32 rows][BN]; MFMA lane (i, h)
// reads element [2s+h][i] of each (ds_read_b32, lanes consecutive -> conflict free).
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int VEC, bool GEMM>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PA = BM + 4, PB = BN + 4;
    constexpr int CA = BM / 4, CB = BN / 4;            // float4 chunks per row
    constexpr int AI = 32 * CA / 256, BI = 32 * CB / 256;
    static_assert(AI >= 1 && BI >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) float As[32 * PA];
    __shared__ __attribute__((aligned(16))) float Bs[32 * PB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const uint32_t tiles_n = (p.N + BN - 1) / BN;
    // 1-D grid over (row slab | batch) x tiles, remapped so that one XCD runs all the tiles of a slab: they re-read the same
    // rows of A and B, which then come from that XCD's L2 instead of eight separate HBM / Infinity-Cache fetches.
    const uint32_t tiles_mn = ((p.M + BM - 1) / BM) * tiles_n;
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t zz = lin / tiles_mn, tile = lin - zz * tiles_mn;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    int split = 0, z = 0;
    if (p.splits > 1) { z = (int)zz / p.splits; split = (int)zz - z * p.splits; } else z = (int)zz;
    const int zo = z / p.Zi, zi = z - zo * p.Zi;
    const float* __restrict__ A = p.A + zo * p.sA_o + zi * p.sA_i;
    const float* __restrict__ B = p.B + zo * p.sB_o + zi * p.sB_i;
    const int r_begin = split * p.rows_per_split;
    const int r_end = min(p.R, r_begin + p.rows_per_split);

    // B staging role: column chunk jc fixed per thread -> tap / channel decode once
    const int jc = tid % CB, br = tid / CB;              // rows br + (256/CB)*i
    const int j = n0 + jc * 4;
    const bool jin = j < p.N;
    int td[VEC == 1 ? 4 : 1], th[VEC == 1 ? 4 : 1], tw[VEC == 1 ? 4 : 1], cch = j;
    bool tok[VEC == 1 ? 4 : 1];
    td[0] = th[0] = tw[0] = 0;
    tok[0] = true;
    if (!GEMM && jin) {
#pragma unroll
        for (int e = 0; e < (VEC == 1 ? 4 : 1); ++e) {
            int tap = (VEC == 1) ? j + e : j / p.Cs;
            if (VEC != 1) cch = j - tap * p.Cs;
            tok[e] = (VEC != 1) || tap < p.g.ntaps;
            td[e] = (int)fdiv((uint32_t)tap, p.g.dKhw);
            int rem = tap - td[e] * (int)p.g.dKhw.d;
            th[e] = (int)fdiv((uint32_t)rem, p.g.dKw);
            tw[e] = rem - th[e] * (int)p.g.dKw.d;
        }
    }
    const int ac = tid % CA, arow = tid / CA;
    const int am = m0 + ac * 4;
    const bool ain = am < p.M;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jj][r] = 0.f;

    f32x4 ra[AI], rb[BI];
    uint32_t okA = 0, okB = 0;            // validity bits of ra[] / rb[]; loads are unconditional, zero fill happens at the LDS store
    auto load_tile = [&](int r0) {
        okA = okB = 0;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = r0 + arow + (256 / CA) * i;
            const bool ok = ain && row < r_end;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (ok ? (int64_t)row * p.lda + am : 0));
            okA |= (uint32_t)ok << i;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int row = r0 + br + (256 / CB) * i;
            const bool rin = jin && row < r_end;
            if constexpr (GEMM) {
                rb[i] = *reinterpret_cast<const f32x4*>(B + (rin ? (int64_t)row * p.ldb + j : 0));
                okB |= (uint32_t)rin << i;
            } else {
                uint32_t n, pk;
                int vox;
                decode_row((uint32_t)(rin ? row : 0), p.g, n, pk);
                if constexpr (VEC == 4) {
                    const bool ok = gather_voxel(pk, td[0], th[0], tw[0], p.g, vox) && rin;
                    rb[i] = *reinterpret_cast<const f32x4*>(B + (ok ? (int64_t)n * p.g.sample_pitch + (int64_t)vox * p.ldb + cch : 0));
                    okB |= (uint32_t)ok << i;
                } else {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = gather_voxel(pk, td[e], th[e], tw[e], p.g, vox) && rin && tok[e];
                        const float x = B[ok ? (int64_t)n * p.g.sample_pitch + (int64_t)vox * p.ldb : 0];
                        v[e] = ok ? x : 0.f;
                    }
                    rb[i] = v;
                    okB |= 1u << i;
                }
            }
        }
    };
    auto store_tile = [&]() {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(arow + (256 / CA) * i) * PA + ac * 4]) = (okA >> i) & 1 ? ra[i] : zero4;
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(br + (256 / CB) * i) * PB + jc * 4]) = (okB >> i) & 1 ? rb[i] : zero4;
    };

    const int fi = lane & 31, fh = lane >> 5;
    if (r_begin < r_end) {
        load_tile(r_begin);
        for (int r0 = r_begin; r0 < r_end; r0 += 32) {
            store_tile();
            __syncthreads();
            if (r0 + 32 < r_end) load_tile(r0 + 32);
            // fragments of k-pair s+2 are requested before the MFMAs of pair s are issued (two pairs of LDS latency cover)
            float a[3][TM], b[3][TN];
            auto frag = [&](int s, float (&fa)[TM], float (&fb)[TN]) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = As[(2 * s + fh) * PA + (wm * TM + i) * 32 + fi];
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) fb[jj] = Bs[(2 * s + fh) * PB + (wn * TN + jj) * 32 + fi];
            };
            frag(0, a[0], b[0]);
            frag(1, a[1], b[1]);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (s + 2 < 16) frag(s + 2, a[(s + 2) % 3], b[(s + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);          // keep the reads ahead: the scheduler otherwise sinks them next to their use
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s % 3][i], b[s % 3][jj], acc[i][jj], 0, 0, 0);
            }
            __syncthreads();
        }
    }

    float* __restrict__ C;
    int64_t ldc;
    if (p.splits > 1) { C = p.ws + ((int64_t)z * p.splits + split) * p.M * p.N; ldc = p.N; }
    else { C = p.C + zo * p.sC_o + zi * p.sC_i; ldc = p.ldc; }
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
        const int col = n0 + (wn * TN + jj) * 32 + (lane & 31);
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.M) C[(int64_t)row * ldc + col] = acc[i][jj][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------
// The same contraction on the bf16 matrix cores (SPLIT, see gemm_mainloop): both operands are contracted over their ROWS, so the
// bf16 fragments (8 consecutive r per lane) need the tiles transposed: a thread loads AI (BI) ADJACENT rows of one float4 column
// chunk, splits them, and stores per column the AI values as one 2*AI-byte piece of the [column][r] plane - the transposition costs no
// instruction, only the choice of which registers go into one store.  Row group is the fastest index across lanes: 16 (8) lanes fill
// one 64-byte LDS row, so the stores are conflict-free; the global loads are 64-byte pieces of 16 (8) different rows per wave.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, bool GEMM>
__global__ __launch_bounds__(256, 2) void wgrad_split_kernel(WgradArgs p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int CA = BM / 4, CB = BN / 4;            // float4 chunks per row
    constexpr int AI = 32 * CA / 256, BI = 32 * CB / 256;      // adjacent rows per thread (2 for a 64-wide operand, 4 for 128)
    static_assert(AI >= 2 && BI >= 2, "tile too small for the transposing store");
    constexpr int NGA = 32 / AI, NGB = 32 / BI;
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * (BM + BN) * SPLIT_PB];
    unsigned char* const Ab = lds;
    unsigned char* const Bb = lds + 3 * BM * SPLIT_PB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const uint32_t tiles_n = (p.N + BN - 1) / BN;
    const uint32_t tiles_mn = ((p.M + BM - 1) / BM) * tiles_n;
    const uint32_t lin = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t zz = lin / tiles_mn, tile = lin - zz * tiles_mn;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    int split = 0, z = 0;
    if (p.splits > 1) { z = (int)zz / p.splits; split = (int)zz - z * p.splits; } else z = (int)zz;
    const int zo = z / p.Zi, zi = z - zo * p.Zi;
    const float* __restrict__ A = p.A + zo * p.sA_o + zi * p.sA_i;
    const float* __restrict__ B = p.B + zo * p.sB_o + zi * p.sB_i;
    const int r_begin = split * p.rows_per_split;
    const int r_end = min(p.R, r_begin + p.rows_per_split);

    const int jc = tid / NGB, bg = tid % NGB;          // B: column chunk jc, rows bg * BI + i
    const int j = n0 + jc * 4;
    const bool jin = j < p.N;
    int td = 0, th = 0, tw = 0, cch = j;
    if (!GEMM && jin) {
        const int tap = j / p.Cs;
        cch = j - tap * p.Cs;
        td = (int)fdiv((uint32_t)tap, p.g.dKhw);
        const int rem = tap - td * (int)p.g.dKhw.d;
        th = (int)fdiv((uint32_t)rem, p.g.dKw);
        tw = rem - th * (int)p.g.dKw.d;
    }
    const int ac = tid / NGA, ag = tid % NGA;          // A: column chunk ac, rows ag * AI + i
    const int am = m0 + ac * 4;
    const bool ain = am < p.M;

    f32x16 acc[TM][TN], sml[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][jj][r] = 0.f; sml[i][jj][r] = 0.f; }

    f32x4 ra[AI], rb[BI];
    uint32_t okA = 0, okB = 0;
    auto load_tile = [&](int r0) {
        okA = okB = 0;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int row = r0 + ag * AI + i;
            const bool ok = ain && row < r_end;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (ok ? (int64_t)row * p.lda + am : 0));
            okA |= (uint32_t)ok << i;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const int row = r0 + bg * BI + i;
            const bool rin = jin && row < r_end;
            if constexpr (GEMM) {
                rb[i] = *reinterpret_cast<const f32x4*>(B + (rin ? (int64_t)row * p.ldb + j : 0));
                okB |= (uint32_t)rin << i;
            } else {
                uint32_t n, pk;
                int vox;
                decode_row((uint32_t)(rin ? row : 0), p.g, n, pk);
                const bool ok = gather_voxel(pk, td, th, tw, p.g, vox) && rin;
                rb[i] = *reinterpret_cast<const f32x4*>(B + (ok ? (int64_t)n * p.g.sample_pitch + (int64_t)vox * p.ldb + cch : 0));
                okB |= (uint32_t)ok << i;
            }
        }
    };
    auto store_tile = [&]() {
        auto stage = [&](auto masked) {
            constexpr bool MK = decltype(masked)::value;
            auto va = [&](int i, int e) { return (!MK || ((okA >> i) & 1)) ? ra[i][e] : 0.f; };
            auto vb = [&](int i, int e) { return (!MK || ((okB >> i) & 1)) ? rb[i][e] : 0.f; };
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int off = (ac * 4 + e) * SPLIT_PB + ag * AI * 2;
                if constexpr (AI == 4) split3_store4(Ab, BM * SPLIT_PB, off, va(0, e), va(1, e), va(2, e), va(3, e));
                else split3_store2(Ab, BM * SPLIT_PB, off, va(0, e), va(1, e));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int off = (jc * 4 + e) * SPLIT_PB + bg * BI * 2;
                if constexpr (BI == 4) split3_store4(Bb, BN * SPLIT_PB, off, vb(0, e), vb(1, e), vb(2, e), vb(3, e));
                else split3_store2(Bb, BN * SPLIT_PB, off, vb(0, e), vb(1, e));
            }
        };
        constexpr uint32_t FA = (1u << AI) - 1, FB = (1u << BI) - 1;
        if (__all(okA == FA && okB == FB)) stage(std::false_type{});      // interior tiles: no zero-fill selects (wave-uniform branch)
        else stage(std::true_type{});
    };

    const int frow = lane & 31, fb = (lane >> 5) * 16;
    if (r_begin < r_end) {
        load_tile(r_begin);
        for (int r0 = r_begin; r0 < r_end; r0 += 32) {
            store_tile();
            __syncthreads();
            if (r0 + 32 < r_end) load_tile(r0 + 32);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[TM][3], b[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        a[i][q] = *reinterpret_cast<const bf16x8*>(Ab + q * BM * SPLIT_PB + ((wm * TM + i) * 32 + frow) * SPLIT_PB + kk * 32 + fb);
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        b[jj][q] = *reinterpret_cast<const bf16x8*>(Bb + q * BN * SPLIT_PB + ((wn * TN + jj) * 32 + frow) * SPLIT_PB + kk * 32 + fb);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int jj = 0; jj < TN; ++jj) {
                        f32x16& t = sml[i][jj];
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[jj][0], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[jj][2], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[jj][1], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[jj][0], t, 0, 0, 0);
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[jj][1], t, 0, 0, 0);
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[jj][0], acc[i][jj], 0, 0, 0);
                    }
            }
            __syncthreads();
        }
    }

    float* __restrict__ C;
    int64_t ldc;
    if (p.splits > 1) { C = p.ws + ((int64_t)z * p.splits + split) * p.M * p.N; ldc = p.N; }
    else { C = p.C + zo * p.sC_o + zi * p.sC_i; ldc = p.ldc; }
#pragma unroll
    for (int jj = 0; jj < TN; ++jj) {
        const int col = n0 + (wn * TN + jj) * 32 + (lane & 31);
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < p.M) C[(int64_t)row * ldc + col] = acc[i][jj][r] + sml[i][jj][r];
            }
    }
}

// out[i] = sum_j ws[j*n + i] in a fixed order: block = 64 consecutive elements x 4 slab lanes (coalesced 256-B rows)
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ ws, float* __restrict__ out, int64_t n, int count,
                                                          int N, int64_t ldc, int64_t out_zstride) {
    __shared__ float red[4][64];
    const int e = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    ws += (int64_t)blockIdx.y * count * n;          // blockIdx.y = batch index of a grouped weight gradient (slabs ws[z][split][n])
    out += (int64_t)blockIdx.y * out_zstride;
    float s = 0.f;
    if (i < n)
        for (int j = rl; j < count; j += 4) s += ws[(int64_t)j * n + i];
    red[rl][e] = s;
    __syncthreads();
    if (rl == 0 && i < n) {
        s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
        if (N > 0) { int64_t r = i / N; out[r * ldc + (i - r * N)] = s; }
        else out[i] = s;
    }
}

#ifndef IGEMM_PROBE_ONLY
// Row splits of a weight gradient: tiles * splits workgroups should fill the resident workgroup slots of the chip evenly (the kernels
// run 5 workgroups per CU: 96-102 VGPRs, 26 KB of LDS), but every split also costs a slab of M x N floats written and read back by the
// reduction.  Cost model: time ~ flops / (fill x 85 TFLOP/s) + 2 x splits x M x N x 4 B / 3 TB/s; the split count with the smallest
// estimate wins (it used to be ceil(2048 / tiles): 1.6 waves of workgroups for most encoder shapes, e.g. 2088 workgroups on 1280 slots
// for the e4 3x3 gradients; measured 60 -> 73 TF/s on the e2 1x1 gradient, 82 -> 87 on e4's, whole step -0.5 %).
static int pick_splits(int R, int M, int N, int BM, int BN, int Z) {
    const int64_t tiles = (int64_t)((M + BM - 1) / BM) * ((N + BN - 1) / BN) * (Z > 1 ? Z : 1);
    int64_t maxs = R / 256;                 // at least 8 K-tiles of work per split
    if (maxs < 1) maxs = 1;
    if (maxs > 4096) maxs = 4096;
    // the split-bf16 kernels (BM >= 64) hold 46 KB of LDS: 3 workgroups per CU at ~120 TFLOP/s; the fp32-input ones 5 at ~85
    const bool split = BM >= 64;
    const int64_t slots = (split ? (BM == 128 && BN == 128 ? 2 : 3) : 5) * 256;
    const double t_full = 2.0 * R * (double)M * N / (split ? 120e12 : 85e12), t_slab = 2.0 * (double)M * N * 4.0 / 3e12;
    int64_t best = 1;
    double best_t = 1e30;
    for (int64_t sp = 1; sp <= maxs && (sp == 1 || sp * tiles <= 6 * slots); ++sp) {
        const int64_t wgs = sp * tiles, waves = (wgs + slots - 1) / slots;
        const double t = t_full * (double)(waves * slots) / (double)wgs + (sp > 1 ? sp * t_slab : 0.0);
        if (t < best_t - 1e-12) { best_t = t; best = sp; }
    }
    return (int)best;
}
static void wgrad_tile(int M, int N, int& BM, int& BN, bool split = true) {
    BM = (M <= 32) ? 32 : 64;
    BN = (M <= 16 && !(M & 3)) ? 256 : 128;     // M <= 16: the 4x4x1 small-M kernel, 256-wide J tiles
    if (M >= 128 && N <= 64) { BM = 128; BN = 64; }      // N <= 64 (dV = P^T dO, 1x1 convs from 64 channels): a 128-wide J tile is half empty (50 -> 65 TF/s)
    // split-bf16 loop: a 64 x 64 wave tile halves the staging (split) work and the LDS fragment reads per MFMA of the 32 x 64 one
    if (split && M >= 128 && !(M & 127) && N >= 128) { BM = 128; BN = 128; }
}

extern "C" size_t corrif_wgrad_workspace(const CorrifWgrad* p) {
    if (!p || p->splits <= 1) return 0;
    return (size_t)(p->Z > 1 ? p->Z : 1) * (size_t)p->splits * (size_t)p->M * (size_t)p->N * sizeof(float);
}
extern "C" int corrif_wgrad_plan(int32_t R, int32_t M, int32_t N, int32_t Z) {
    int BM, BN;
    wgrad_tile(M, N, BM, BN);
    if (BN == 256) BM = 16;
    return pick_splits(R, M, N, BM, BN, Z);
}

extern "C" int corrif_wgrad_is_split(const CorrifWgrad* p) {
    if (!p || p->f32_mfma || p->Cs == 1) return 0;
    int BM, BN;
    wgrad_tile(p->M, p->N, BM, BN, true);
    return BN != 256 && BM >= 64;
}

extern "C" int corrif_slab_reduce(const float* ws, float* out, int64_t n, int32_t count, void* stream) {
    if (!ws || !out || n <= 0 || count <= 0) return CORRIF_EINVAL;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, ws, out, n,
                       count, 0, (int64_t)0, (int64_t)0);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

extern "C" int corrif_wgrad(const CorrifWgrad* p, void* stream) {
    if (!p || !p->A || !p->B || !p->C) return CORRIF_EINVAL;
    if (p->R <= 0 || p->M <= 0 || p->N <= 0 || p->splits < 1 || p->Z < 1 || p->Zi < 1) return CORRIF_EINVAL;
    const bool scalar = p->Cs == 1;
    if ((p->M & 3) || (p->N & 3) || p->Cs <= 0 || (p->lda & 3)) return CORRIF_EUNSUPPORTED;
    if (!scalar && ((p->Cs & 3) || (p->ldb & 3) || ((uintptr_t)p->B & 15) || (p->sB_o & 3) || (p->sB_i & 3) ||
                    (p->g.src_batch_pitch & 3)))
        return CORRIF_EUNSUPPORTED;
    if (scalar && p->g.is_gemm) return CORRIF_EUNSUPPORTED;
    if (((uintptr_t)p->A & 15) || (p->sA_o & 3) || (p->sA_i & 3)) return CORRIF_EUNSUPPORTED;
    if (p->splits > 1 && (!p->ws || (p->Z != 1 && p->Zi != 1))) return CORRIF_EINVAL;
    if (!geom_ok(p->g)) return CORRIF_EINVAL;
    if (!p->g.is_gemm && !scalar && p->N != p->g.kd * p->g.kh * p->g.kw * p->Cs) return CORRIF_EINVAL;
    if (scalar && (p->g.ntaps <= 0 || p->g.ntaps > p->g.kd * p->g.kh * p->g.kw || p->N < p->g.ntaps)) return CORRIF_EINVAL;
    if (p->Z > 65535 || p->splits > 65535 || (int64_t)p->Z * p->splits > 65535) return CORRIF_EUNSUPPORTED;
    WgradArgs a;
    a.A = p->A; a.B = p->B; a.C = p->C; a.ws = p->ws;
    a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc;
    a.R = p->R; a.M = p->M; a.N = p->N; a.Cs = p->Cs; a.splits = p->splits; a.Zi = p->Zi; a.Z = p->Z;
    int rps = (p->R + p->splits - 1) / p->splits;
    a.rows_per_split = (rps + 31) / 32 * 32;
    a.sA_o = p->sA_o; a.sA_i = p->sA_i; a.sB_o = p->sB_o; a.sB_i = p->sB_i; a.sC_o = p->sC_o; a.sC_i = p->sC_i;
    a.g = make_devgeom(p->g, p->ldb);
    hipStream_t s = (hipStream_t)stream;
    int BM, BN;
    wgrad_tile(p->M, p->N, BM, BN, !p->f32_mfma);
    if (scalar) { BM = 64; BN = 128; }
    if (BN == 256) BM = 16;
    uint32_t tiles = (uint32_t)((p->M + BM - 1) / BM) * (uint32_t)((p->N + BN - 1) / BN);
    const uint32_t nz = (uint32_t)(p->splits > 1 ? p->splits * p->Z : p->Z);
    if ((uint64_t)tiles * nz >= (1ull << 31)) return CORRIF_EUNSUPPORTED;
    dim3 grid(tiles * nz, 1, 1);
    if (scalar) {
        grid.x = (uint32_t)((p->M + 63) / 64) * (uint32_t)((p->N + 127) / 128) * nz;
        hipLaunchKernelGGL((wgrad_kernel<64, 128, 2, 2, 1, false>), grid, dim3(256), 0, s, a);
    } else if (BN == 256) {
        int rc = launch_smallm_wgrad(a, (int)nz, s);
        if (rc != CORRIF_OK) return rc;
    } else if (BM == 128 && BN == 128) {
        if (a.g.is_gemm) hipLaunchKernelGGL((wgrad_split_kernel<128, 128, 2, 2, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_split_kernel<128, 128, 2, 2, false>), grid, dim3(256), 0, s, a);
    } else if (BM == 128 && BN == 64) {
        if (!p->f32_mfma) {
            if (a.g.is_gemm) hipLaunchKernelGGL((wgrad_split_kernel<128, 64, 4, 1, true>), grid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((wgrad_split_kernel<128, 64, 4, 1, false>), grid, dim3(256), 0, s, a);
        } else if (a.g.is_gemm) hipLaunchKernelGGL((wgrad_kernel<128, 64, 4, 1, 4, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_kernel<128, 64, 4, 1, 4, false>), grid, dim3(256), 0, s, a);
    } else if (BM == 32) {
        if (a.g.is_gemm) hipLaunchKernelGGL((wgrad_kernel<32, 128, 1, 4, 4, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_kernel<32, 128, 1, 4, 4, false>), grid, dim3(256), 0, s, a);
    } else if (!p->f32_mfma) {
        if (a.g.is_gemm) hipLaunchKernelGGL((wgrad_split_kernel<64, 128, 2, 2, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_split_kernel<64, 128, 2, 2, false>), grid, dim3(256), 0, s, a);
    } else {
        if (a.g.is_gemm) hipLaunchKernelGGL((wgrad_kernel<64, 128, 2, 2, 4, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((wgrad_kernel<64, 128, 2, 2, 4, false>), grid, dim3(256), 0, s, a);
    }
    CORRIF_CHECK_LAUNCH();
    if (p->splits > 1) {
        int64_t n = (int64_t)p->M * p->N;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)p->Z), dim3(256), 0, s, p->ws, p->C, n, p->splits,
                           p->N, p->ldc, p->sC_o);
        CORRIF_CHECK_LAUNCH();
    }
    return CORRIF_OK;
}
#endif  // IGEMM_PROBE_ONLY

