// Implicit-GEMM convolution / GEMM family on the gfx950 FP32-input matrix cores.
//
//   gemm_fwd_kernel  : C[M,N] = act(gather(A)[M,K] . B^T + bias + addend)    (forward conv, data gradient,
//                      nn.Linear, batched q.k^T / p.v)
//   wgrad_kernel     : C[M,N] = sum_r A[r][M] * gather(B)[r][N]               (weight gradients, P^T.dO, dS^T.Q)
//
// Both use v_mfma_f32_32x32x2_f32 (exact f32 fma chain; 64 cycles/SIMD each) on 32x32 register tiles.
// A wave's A/B fragments come from LDS; because the MFMA k-index only has to agree between A and B,
// lane (i, h) reads ONE float4 = k {4h..4h+3} of its row and feeds element j to MFMA j (4 MFMAs per
// pair of ds_read_b128).  Activations are channels-last so a tap's channel run is contiguous: the
// gather is done while staging global -> registers -> LDS, never as an im2col buffer.
#pragma once
#include <type_traits>
#include "common.h"
#include "igemm_args.h"


// GEMM / BL (= is_gemm / b_layout) are compile-time: one straight-line K loop per variant lets the compiler keep all tile
// loads in flight together (run-time variants shared basic blocks and forced vmcnt drains at the joins).
//
// The kernel body is split in two device functions so that three kernels can share it:
//   gemm_mainloop : acc += sum over the K tiles [kt0, kt1) of one BM x BN output tile
//   gemm_epilogue : bias / addend / activation / fused norm statistics / row map, float4 stores through an LDS transpose
//   gemm_fwd_kernel      one workgroup per output tile (grids that fill the chip many times over)
//   gemm_sk_kernel       "stream-K": a persistent grid of G = CUs x resident workgroups; workgroup g owns the g-th equal share of
//                        the (tile, K tile) iteration space, so every CU gets the same number of MFMA iterations whatever the tile
//                        count (392 tiles of an e4 conv on 512 slots ran at 392/512 = 0.77 occupancy; several shapes of the encoder
//                        sat at ceil() losses of 15-25 %).  Pieces that cover a whole tile run the normal epilogue; partial pieces
//                        write their raw accumulators to a slab (at most 2 per workgroup),
//   gemm_sk_fixup_kernel sums the slabs of every split tile in a FIXED order (ascending workgroup) and runs the same epilogue:
//                        deterministic, no atomics, no inter-workgroup wait inside a launch.
template <int BM, int BN, int WM, int WN>
struct GemmCfg {
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int SP = TN * 32 + 4;                       // epilogue staging row pitch (floats)
};

// SPLIT ("bf16x6"): the main loop of the 128-row tiles runs on the bf16 matrix cores with fp32-grade results.  While a tile is staged to
// LDS every fp32 operand x is split EXACTLY into three bf16 terms x = h + m + l (round to nearest: 8 + 8 + 8 significand bits); the
// product x*y is then the six bf16 products  hh' + (hm' + mh') + (hl' + mm' + lh')  accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The
// three dropped terms ml' + lm' + ll' are <= 2^-25 |xy| - less than half an fp32 ulp of the product, i.e. less than what one fp32 fma
// rounds away.  The leading products hh' accumulate in `acc`, the five small ones in a second accumulator that is folded in once at the
// end, so the K sum sees ONE fp32 rounding per 16 products instead of 16: measured against fp64 (tools/split_lab.hip, full 24-bit random
// operands) the relative L2 error is 0.36x that of the v_mfma_f32_32x32x2_f32 chain at K = 2304-4608 (3.1e-7 vs 8.6e-7) and the kernel is
// 1.4-1.5x faster, because six 32-cycle bf16 MFMAs per K = 16 replace eight 64-cycle fp32-input MFMAs.  This also retires the two-level
// "KS" accumulation of round 3 (the fp32 chain's sqrt(K) error growth was the source of the 1.5x end-to-end gradient-error excess).
// LDS image: per operand three planes (h, m, l) of [rows][32 bf16], row pitch 80 B (20 dwords: 8 consecutive rows cover all 32 banks for
// the ds_read_b128 fragment loads and for the 8-byte staging stores).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define SPLIT_PB 80
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two values at a time: one v_cvt_pk_bf16_f32 per term and pair; h / m / l receive the packed bf16 pairs (x0's term in the low half)
__device__ __forceinline__ void split3_pair(const float x0, const float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{x0, x1}, bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);      // exact
    m = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{r0, r1}, bf16x2));
    const float s0 = r0 - __builtin_bit_cast(float, m << 16), s1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);      // exact, <= 8 bits
    l = __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{s0, s1}, bf16x2));
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// four values that are consecutive along k in one LDS row: three 8-byte stores (planes h, m, l)
__device__ __forceinline__ void split3_store4(unsigned char* base, const int plane_bytes, const int off, const float v0, const float v1,
                                              const float v2, const float v3) {
    uint32_t h0, m0, l0, h1, m1, l1;
    split3_pair(v0, v1, h0, m0, l0);
    split3_pair(v2, v3, h1, m1, l1);
    *reinterpret_cast<u32x2*>(base + off) = u32x2{h0, h1};
    *reinterpret_cast<u32x2*>(base + plane_bytes + off) = u32x2{m0, m1};
    *reinterpret_cast<u32x2*>(base + 2 * plane_bytes + off) = u32x2{l0, l1};
}
__device__ __forceinline__ void split3_store2(unsigned char* base, const int plane_bytes, const int off, const float v0, const float v1) {
    uint32_t h, m, l;
    split3_pair(v0, v1, h, m, l);
    *reinterpret_cast<uint32_t*>(base + off) = h;
    *reinterpret_cast<uint32_t*>(base + plane_bytes + off) = m;
    *reinterpret_cast<uint32_t*>(base + 2 * plane_bytes + off) = l;
}
template <int BM, int BN, int WM, int WN, int VEC, bool GEMM, int BL, int SPLIT = 0>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& p, const float* __restrict__ A, const float* __restrict__ B, float* lds,
                                              const int m0, const int n0, const int kt0, const int kt1,
                                              f32x16 (&acc)[BM / WM / 32][BN / WN / 32]) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(!SPLIT || (VEC == 4 && (BM == 128 || BM == 64) && (BN == 128 || BN == 64)), "split-bf16 main loop: float4 loader, 64- / 128-wide tiles");
    constexpr int AI = BM / 32, BI = BN / 32;      // float4 chunks per thread per K tile
    constexpr int PBT = BN + 4;                           // row pitch of the K-major B tile (BL == 1)
    float* const As = lds;
    float* const Bs = lds + BM * LDS_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // ---- per-thread staging roles: A chunk column kc (fixed), rows ar + 32*i
    const int kc = tid & 7, ar = tid >> 3;
    // per row: GEMM mode needs nothing but the row number; gather mode keeps the sample index and the packed (d,h,w) of the row
    // (32-bit each: the 64-bit base is re-derived at the load, registers are what limits the workgroups per CU)
    uint32_t a_n[AI], a_pack[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        int row = m0 + ar + 32 * i;
        a_n[i] = 0;
        a_pack[i] = 0xFFFFFFFFu;
        if (row < p.M) {
            if constexpr (GEMM) {
                a_pack[i] = 0;
            } else {
                uint32_t n, pk;
                decode_row((uint32_t)row, p.g, n, pk);
                a_n[i] = n;
                a_pack[i] = pk;
            }
        }
    }

    f32x4 ra[AI], rb[BI];
    int a_off[AI];                        // cached per-row source offsets (floats inside the sample) of tap cur_tap; -1 = zero padding
    uint32_t okA = 0, okB = 0;            // validity bits of ra[] / rb[] (applied at the LDS store)
    // (tap, float4 chunk) of this thread's chunk in the NEXT tile to load (tiles load in order, starting at K tile kt0)
    int cur_tap = -1, nx_tap = 0, nx_c4 = kc;
    if constexpr (!GEMM && VEC == 4) {
        if (kt0 > 0) {
            const int q0 = kt0 * 8 + kc;
            nx_tap = q0 / p.Cs4;
            nx_c4 = q0 - nx_tap * p.Cs4;
        }
    }
#pragma unroll
    for (int i = 0; i < AI; ++i) a_off[i] = -1;

    auto load_tile = [&](int kt) {
        // ---------------- A (gathered) ----------------
        const int q = kt * 8 + kc;          // global float4 chunk index along K
        const int k = q * 4;
        bool kin = k < p.K;
        if constexpr (VEC == 4) {
            int c = k;
            if (!GEMM && kin) {
                // K runs (tap, channel): the voxel offsets of this thread's rows only change when its chunk crosses into a new
                // tap, so they are cached and re-derived on tap change (every Cs/32 K tiles), not per tile.
                while (nx_c4 >= p.Cs4) { nx_c4 -= p.Cs4; ++nx_tap; }
                c = nx_c4 * 4;
                if (nx_tap != cur_tap) {
                    cur_tap = nx_tap;
                    const int tap = p.ntap_sel ? p.tap_sel[nx_tap] : nx_tap;
                    const int td = (int)fdiv((uint32_t)tap, p.g.dKhw);
                    const int rem = tap - td * (int)p.g.dKhw.d;
                    const int th = (int)fdiv((uint32_t)rem, p.g.dKw);
                    const int tw = rem - th * (int)p.g.dKw.d;
#pragma unroll
                    for (int i = 0; i < AI; ++i) {
                        int vox;
                        a_off[i] = (a_pack[i] != 0xFFFFFFFFu && gather_voxel(a_pack[i], td, th, tw, p.g, vox)) ? vox * (int)p.lda : -1;
                    }
                }
                nx_c4 += 8;
            }
            // Loads are UNCONDITIONAL (masked-off lanes read the 16 bytes at A, always mapped) and the zero fill happens at the
            // LDS store: a load under a branch makes the compiler drain vmcnt at every join, serialising the round trips.
            okA = 0;
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                const bool ok = kin && (GEMM ? a_pack[i] != 0xFFFFFFFFu : a_off[i] >= 0);
                const int64_t off = GEMM ? (int64_t)(m0 + ar + 32 * i) * p.lda + k : (int64_t)a_n[i] * p.g.sample_pitch + a_off[i] + c;
                ra[i] = *reinterpret_cast<const f32x4*>(A + (ok ? off : 0));
                okA |= (uint32_t)ok << i;
            }
        } else {   // Cs == 1 (the stem): the 4 k's of a chunk are 4 different taps, scalar gathers
            int td[4], th[4], tw[4];
            bool tok[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int tap = k + e;
                tok[e] = tap < p.g.ntaps;
                td[e] = (int)fdiv((uint32_t)tap, p.g.dKhw);
                int rem = tap - td[e] * (int)p.g.dKhw.d;
                th[e] = (int)fdiv((uint32_t)rem, p.g.dKw);
                tw[e] = rem - th[e] * (int)p.g.dKw.d;
            }
            okA = ~0u;                     // this path zero-fills per element below
#pragma unroll
            for (int i = 0; i < AI; ++i) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int vox;
                    const bool ok = kin && a_pack[i] != 0xFFFFFFFFu && tok[e] && gather_voxel(a_pack[i], td[e], th[e], tw[e], p.g, vox);
                    const float x = A[ok ? (int64_t)a_n[i] * p.g.sample_pitch + (int64_t)vox * p.lda : 0];      // unconditional load, select afterwards
                    v[e] = ok ? x : 0.f;
                }
                ra[i] = v;
            }
        }
        // ---------------- B ----------------
        okB = 0;
        if constexpr (BL == 0) {              // [N][K]
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                const int n = n0 + ar + 32 * i;
                const bool ok = kin && n < p.N;
                rb[i] = *reinterpret_cast<const f32x4*>(B + (ok ? (int64_t)n * p.ldb + k : 0));
                okB |= (uint32_t)ok << i;
            }
        } else {                            // [K][N]: float4 along n
            constexpr int CN = BN / 4;
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                int nc, kk;
                if constexpr (SPLIT) {      // BI ADJACENT k rows per thread (transposed in registers at the LDS store), k group fastest across lanes
                    constexpr int NG = BK / BI;
                    nc = tid / NG;
                    kk = (tid % NG) * BI + i;
                } else {
                    const int cidx = tid + 256 * i;
                    nc = cidx % CN;
                    kk = cidx / CN;
                }
                int kg = kt * BK + kk, n = n0 + nc * 4;
                const bool ok = kg < p.K && n < p.N;
                rb[i] = *reinterpret_cast<const f32x4*>(B + (ok ? (int64_t)kg * p.ldb + n : 0));
                okB |= (uint32_t)ok << i;
            }
        }
    };
    // split-bf16 LDS image (bytes): A planes h, m, l ([BM][SPLIT_PB]) then B planes h, m, l ([BN][SPLIT_PB])
    unsigned char* const Ab = reinterpret_cast<unsigned char*>(lds);
    unsigned char* const Bb = Ab + 3 * BM * SPLIT_PB;
    auto store_tile = [&]() {
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (SPLIT) {
            // zero fill (K / M / N edges, padding taps) costs one select per register: interior tiles - the common case - take a
            // wave-uniform branch around all of them (56 of ~250 vector instructions per K tile; the loop is issue-bound)
            auto stage = [&](auto masked) {
                constexpr bool MK = decltype(masked)::value;
#pragma unroll
                for (int i = 0; i < AI; ++i) {
                    const f32x4 v = (!MK || ((okA >> i) & 1)) ? ra[i] : zero4;
                    split3_store4(Ab, BM * SPLIT_PB, (ar + 32 * i) * SPLIT_PB + kc * 8, v[0], v[1], v[2], v[3]);
                }
                if constexpr (BL == 0) {
#pragma unroll
                    for (int i = 0; i < BI; ++i) {
                        const f32x4 v = (!MK || ((okB >> i) & 1)) ? rb[i] : zero4;
                        split3_store4(Bb, BN * SPLIT_PB, (ar + 32 * i) * SPLIT_PB + kc * 8, v[0], v[1], v[2], v[3]);
                    }
                } else {
                    // [K][N] operand: this thread holds BI adjacent k rows x 4 columns; per column the BI values are one 2*BI-byte store
                    constexpr int NG = BK / BI;
                    const int nc = tid / NG, kgp = tid % NG;
                    f32x4 v[BI];
#pragma unroll
                    for (int i = 0; i < BI; ++i) v[i] = (!MK || ((okB >> i) & 1)) ? rb[i] : zero4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int off = (nc * 4 + e) * SPLIT_PB + kgp * BI * 2;
                        if constexpr (BI == 4) split3_store4(Bb, BN * SPLIT_PB, off, v[0][e], v[1][e], v[2][e], v[3][e]);
                        else split3_store2(Bb, BN * SPLIT_PB, off, v[0][e], v[1][e]);
                    }
                }
            };
            constexpr uint32_t FA = (1u << AI) - 1, FB = (1u << BI) - 1;
            if (__all((okA & FA) == FA && (okB & FB) == FB)) stage(std::false_type{});
            else stage(std::true_type{});
            return;
        }
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(ar + 32 * i) * LDS_PITCH + kc * 4]) = (okA >> i) & 1 ? ra[i] : zero4;
        if constexpr (BL == 0) {
#pragma unroll
            for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(&Bs[(ar + 32 * i) * LDS_PITCH + kc * 4]) = (okB >> i) & 1 ? rb[i] : zero4;
        } else {
            // [K][N] operand: the LDS tile stays K-major ([32][BN + 4], float4 stores without bank conflicts); the MFMA B
            // fragments are then 4 ds_read_b32 per 8 k instead of one ds_read_b128 (a transposing scalar store was 8-way conflicted)
            constexpr int CN = BN / 4;
#pragma unroll
            for (int i = 0; i < BI; ++i) {
                int cidx = tid + 256 * i;
                int nc = cidx % CN, kk = cidx / CN;
                *reinterpret_cast<f32x4*>(&Bs[kk * PBT + nc * 4]) = (okB >> i) & 1 ? rb[i] : zero4;
            }
        }
    };

    const int frow = lane & 31, fk = (lane >> 5) * 4;
    // SPLIT == 2: the five small products of every element in their own accumulator (acc carries hh'); SPLIT == 1: all six in acc
    // (the persistent stream-K kernel has no registers for a second set; its K sum still sees 6 roundings per 16 products instead of 16)
    f32x16 sml[SPLIT == 2 ? TM : 1][SPLIT == 2 ? TN : 1];
    if constexpr (SPLIT == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sml[i][j][r] = 0.f;
    }
    load_tile(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < kt1) load_tile(kt + 1);      // global loads fly while the MFMAs run
        if constexpr (SPLIT) {
            const int fb = (lane >> 5) * 16;      // byte offset of this lane's 8 k inside a 16-deep slab (lane (r, h): k = 8h .. 8h+7)
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8 a[TM][3], b[TN][3];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        a[i][q] = *reinterpret_cast<const bf16x8*>(Ab + q * BM * SPLIT_PB + ((wm * TM + i) * 32 + frow) * SPLIT_PB + kk * 32 + fb);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        b[j][q] = *reinterpret_cast<const bf16x8*>(Bb + q * BN * SPLIT_PB + ((wn * TN + j) * 32 + frow) * SPLIT_PB + kk * 32 + fb);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        f32x16& t = SPLIT == 2 ? sml[i][j] : acc[i][j];
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], t, 0, 0, 0);      // l h'
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], t, 0, 0, 0);      // h l'
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], t, 0, 0, 0);      // m m'
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], t, 0, 0, 0);      // m h'
                        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], t, 0, 0, 0);      // h m'
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);      // h h'
                    }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 8; ++kk) {
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i] = *reinterpret_cast<const f32x4*>(&As[((wm * TM + i) * 32 + frow) * LDS_PITCH + kk * 8 + fk]);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (BL == 0) {
                        b[j] = *reinterpret_cast<const f32x4*>(&Bs[((wn * TN + j) * 32 + frow) * LDS_PITCH + kk * 8 + fk]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) b[j][e] = Bs[(kk * 8 + fk + e) * PBT + (wn * TN + j) * 32 + frow];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    if constexpr (SPLIT == 2) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += sml[i][j];
    }
}

// ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5): a direct store is 16*TM*TN
// scalar store instructions per lane (store-issue bound when K is short).  Instead every wave transposes one 32 x (32*TN)
// row block at a time through its private LDS staging area and writes float4 per lane: 4x fewer store instructions and
// 128*TN-byte contiguous row segments.  (The caller has fenced the A/B tiles with a __syncthreads() before calling.)
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, float* __restrict__ C, float* lds, const int m0, const int n0,
                                              f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const int zo) {
    // per-group epilogue operands of a grouped launch (zo = outer batch index; all strides are 0 for an ordinary launch)
    const float* __restrict__ const bias = p.bias ? p.bias + zo * p.zs_bias : nullptr;
    const float* __restrict__ const addend = p.addend ? p.addend + zo * p.zs_add : nullptr;
    const float* __restrict__ const addend2 = p.addend2 ? p.addend2 + zo * p.zs_add2 : nullptr;
    const float* __restrict__ const bs_x = p.bs_x ? p.bs_x + zo * p.zs_bsx : nullptr;
    const float* __restrict__ const bs_y = p.bs_y ? p.bs_y + zo * p.zs_bsy : nullptr;
    double* __restrict__ const stats_part = p.stats_part ? p.stats_part + zo * p.zs_stats : nullptr;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int SP = GemmCfg<BM, BN, WM, WN>::SP;
    constexpr int CQ = TN * 8;                    // float4 chunks per staged row
    constexpr int RPP = 64 / CQ;                  // rows per pass
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    float* const stg = lds + wave * 32 * SP;
    const int rr = lane / CQ, cq = lane % CQ;
    const int col = n0 + wn * TN * 32 + cq * 4;   // this lane's 4 output columns (fixed for all passes)
    const bool vec_ok = !(p.ldc & 3) && !((uintptr_t)C & 15) && (!addend || (!(p.ld_add & 3) && !((uintptr_t)addend & 15))) &&
                        (!addend2 || (!(p.ld_add2 & 3) && !((uintptr_t)addend2 & 15)));
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < p.N) bv[e] = bias[col + e];
    }
    // fused norm statistics of this lane's 4 columns, in double from the first add on: E[x^2] - E[x]^2 cancels in nearly constant
    // channels, and the reference (ATen on the CPU) accumulates its batch statistics in double too.  (fp32 per-lane partials of <= 32
    // values were tried in round 3: -1 ms per step, but a channel with var / mean^2 ~ 1e-5 then carries a 1 % error in rstd - the MMVit2
    // 32 x 32 fixture lost a gradient-norm bracket to it.)
    double ssum[4] = {0, 0, 0, 0}, ssq[4] = {0, 0, 0, 0};
    f32x4 bmu = {0.f, 0.f, 0.f, 0.f}, brs = {0.f, 0.f, 0.f, 0.f};      // backward-statistics mode: mean / rstd of this lane's 4 columns
    if (bs_x) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < p.N) { bmu[e] = p.bs_mean[zo * p.zs_bsstat + col + e]; brs[e] = p.bs_rstd[zo * p.zs_bsstat + col + e]; }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stg[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * SP + j * 32 + (lane & 31)] = acc[i][j][r];
        // same wave wrote and reads: LDS ops of one wave complete in order, no barrier needed
#pragma unroll
        for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int lr = ps * RPP + rr;
            const int row = m0 + (wm * TM + i) * 32 + lr;
            f32x4 v = *reinterpret_cast<const f32x4*>(&stg[lr * SP + cq * 4]);
            if (row >= p.M || col >= p.N) continue;
            v += bv;
            if (stats_part && !bs_x) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const double sv = (double)(p.stats_relu ? fmaxf(v[e], 0.f) : v[e]);
                    ssum[e] += sv;
                    ssq[e] += sv * sv;
                }
            }
            int64_t orow = row;
            if (p.out_map) {       // scatter to the strided sub-grid this GEMM's rows enumerate
                uint32_t n, pk;
                decode_row((uint32_t)row, p.g, n, pk);
                const int od = (int)(pk >> 20) * p.om_d + p.oo_d, oh = (int)((pk >> 10) & 1023) * p.om_h + p.oo_h,
                          ow = (int)(pk & 1023) * p.om_w + p.oo_w;
                orow = (((int64_t)n * p.OD + od) * p.OH + oh) * p.OW + ow;
            }
            float* __restrict__ dst = C + orow * p.ldc + col;
            if (vec_ok && col + 3 < p.N) {
                if (addend) v += *reinterpret_cast<const f32x4*>(addend + (int64_t)row * p.ld_add + col);
                if (addend2) v += *reinterpret_cast<const f32x4*>(addend2 + (int64_t)row * p.ld_add2 + col);
                if (p.act == CORRIF_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.act == CORRIF_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
                }
                if (bs_x) {      // v is the complete gradient w.r.t. the producing BatchNorm's output (addends included)
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(bs_x + (int64_t)row * p.bs_ldx + col);
                    f32x4 yv = {1.f, 1.f, 1.f, 1.f};
                    if (bs_y) yv = *reinterpret_cast<const f32x4*>(bs_y + (int64_t)row * p.bs_ldy + col);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gm = yv[e] > 0.f ? v[e] : 0.f;
                        const float xh = (xv[e] - bmu[e]) * brs[e];
                        ssum[e] += (double)gm;
                        ssq[e] += (double)gm * (double)xh;
                    }
                }
                *reinterpret_cast<f32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col + e >= p.N) break;
                    float x = v[e];
                    if (addend) x += addend[(int64_t)row * p.ld_add + col + e];
                    if (addend2) x += addend2[(int64_t)row * p.ld_add2 + col + e];
                    if (p.act == CORRIF_ACT_RELU) x = fmaxf(x, 0.f);
                    else if (p.act == CORRIF_ACT_GELU) x = gelu_erf(x);
                    if (bs_x) {
                        const float yv = bs_y ? bs_y[(int64_t)row * p.bs_ldy + col + e] : 1.f;
                        const float gm = yv > 0.f ? x : 0.f;
                        ssum[e] += (double)gm;
                        ssq[e] += (double)gm * (double)((bs_x[(int64_t)row * p.bs_ldx + col + e] - bmu[e]) * brs[e]);
                    }
                    dst[e] = x;
                }
            }
        }
    }
    if (stats_part) {       // TM == 2: the wave's rows are one 64-row block; sum over the RPP lanes that share a column chunk
#pragma unroll
        for (int off = CQ; off < 64; off <<= 1)
#pragma unroll
            for (int e = 0; e < 4; ++e) { ssum[e] += __shfl_xor(ssum[e], off); ssq[e] += __shfl_xor(ssq[e], off); }
        const int row0 = m0 + wm * TM * 32;
        if (lane < CQ && row0 < p.M) {
            const int g = row0 / p.stats_rpg, chunk = (row0 - g * p.stats_rpg) >> 6;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < p.N) {
                    double* o = stats_part + (((int64_t)g * p.N + col + e) * p.stats_chunks + chunk) * 2;
                    o[0] = ssum[e];
                    o[1] = ssq[e];
                }
        }
    }
}

template <int BM, int BN, int WM, int WN, int BL>
struct GemmLds {
    static constexpr int TN = BN / WN / 32;
    static constexpr int SP = TN * 32 + 4;
    static constexpr int PBT = BN + 4;
    static constexpr int B_FLOATS = BL == 0 ? BN * LDS_PITCH : BK * PBT;
    // one LDS array: [A tile | B tile] during the K loop, re-used as the per-wave output staging area in the epilogue
    static constexpr int FLOATS = BM * LDS_PITCH + B_FLOATS > 4 * 32 * SP ? BM * LDS_PITCH + B_FLOATS : 4 * 32 * SP;
    static constexpr int SPLIT_FLOATS = 3 * (BM + BN) * SPLIT_PB / 4 > 4 * 32 * SP ? 3 * (BM + BN) * SPLIT_PB / 4 : 4 * 32 * SP;
};

template <int BM, int BN, int WM, int WN, int VEC, bool GEMM, int BL, int SPLIT = 0>
__global__ __launch_bounds__(256, SPLIT ? 2 : 1) void gemm_fwd_kernel(GemmArgs p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    __shared__ __attribute__((aligned(16))) float lds[SPLIT ? GemmLds<BM, BN, WM, WN, BL>::SPLIT_FLOATS : GemmLds<BM, BN, WM, WN, BL>::FLOATS];
    const uint32_t tiles_n = (p.N + BN - 1) / BN;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int z = blockIdx.z, zo = z / p.Zi, zi = z - zo * p.Zi;
    const float* __restrict__ A = p.A + zo * p.sA_o + zi * p.sA_i;
    const float* __restrict__ B = p.B + zo * p.sB_o + zi * p.sB_i;
    float* __restrict__ C = p.C + zo * p.sC_o + zi * p.sC_i;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gemm_mainloop<BM, BN, WM, WN, VEC, GEMM, BL, SPLIT>(p, A, B, lds, m0, n0, 0, (p.K + BK - 1) / BK, acc);
    gemm_epilogue<BM, BN, WM, WN>(p, C, lds, m0, n0, acc, zo);
}

// unit boundary of stream-K workgroup g: floor(g * U / G)
__device__ __forceinline__ int64_t sk_bound(int64_t g, int64_t U, int G) { return g * U / G; }

template <int BM, int BN, int WM, int WN, int VEC, bool GEMM, int BL, int SPLIT = 0>
__global__ __launch_bounds__(256, SPLIT ? 2 : 1) void gemm_sk_kernel(GemmArgs p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    __shared__ __attribute__((aligned(16))) float lds[SPLIT ? GemmLds<BM, BN, WM, WN, BL>::SPLIT_FLOATS : GemmLds<BM, BN, WM, WN, BL>::FLOATS];
    const int G = (int)gridDim.x;
    const int g = (int)xcd_remap(blockIdx.x, gridDim.x);      // neighbouring ranges (shared operand panels) on one XCD's L2
    const int nk = p.sk_nk;
    const int64_t U = (int64_t)p.sk_tiles * nk;
    const int64_t u0 = sk_bound(g, U, G), u1 = sk_bound(g + 1, U, G);
    const uint32_t tiles_n = (p.N + BN - 1) / BN;
    const int tid = threadIdx.x;
    for (int64_t u = u0; u < u1;) {
        const int t = (int)(u / nk);
        const int k0 = (int)(u - (int64_t)t * nk);
        const int k1 = (int)min((int64_t)nk, k0 + (u1 - u));
        const int z = t / p.sk_tiles_mn, tile = t - z * p.sk_tiles_mn;
        const int zo = z / p.Zi, zi = z - zo * p.Zi;
        const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
        const float* __restrict__ A = p.A + zo * p.sA_o + zi * p.sA_i;
        const float* __restrict__ B = p.B + zo * p.sB_o + zi * p.sB_i;
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        gemm_mainloop<BM, BN, WM, WN, VEC, GEMM, BL, SPLIT>(p, A, B, lds, m0, n0, k0, k1, acc);
        if (k0 == 0 && k1 == nk) {
            gemm_epilogue<BM, BN, WM, WN>(p, p.C + zo * p.sC_o + zi * p.sC_i, lds, m0, n0, acc, zo);
            __syncthreads();                                  // the staging area is the next piece's A/B tile
        } else {
            // partial piece: raw accumulators to this workgroup's slab (0: the piece its range starts with, 1: a later one)
            float* __restrict__ slab = p.sk_ws + ((int64_t)2 * g + (u == u0 ? 0 : 1)) * (BM * BN);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) {
                        f32x4 v = {acc[i][j][r], acc[i][j][r + 1], acc[i][j][r + 2], acc[i][j][r + 3]};
                        *reinterpret_cast<f32x4*>(slab + ((((i * TN + j) * 4 + (r >> 2)) * 256 + tid) << 2)) = v;
                    }
        }
        u += k1 - k0;
    }
}

// one workgroup per interior range boundary g = 1 .. G-1: the workgroup of the FIRST boundary that falls inside a tile owns that tile
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_sk_fixup_kernel(GemmArgs p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    __shared__ __attribute__((aligned(16))) float lds[4 * 32 * GemmCfg<BM, BN, WM, WN>::SP];
    const int G = p.sk_G, nk = p.sk_nk;
    const int g = (int)blockIdx.x + 1;
    const int64_t U = (int64_t)p.sk_tiles * nk;
    const int64_t ub = sk_bound(g, U, G);
    const int t = (int)(ub / nk);
    const int64_t tb = (int64_t)t * nk, te = tb + nk;
    if (ub == tb) return;                                     // boundary on a tile edge: nothing is split here
    if (sk_bound(g - 1, U, G) > tb) return;                   // an earlier boundary already lies inside this tile: its workgroup owns it
    const int tid = threadIdx.x;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int j2 = g - 1; j2 < G; ++j2) {                       // pieces in ascending workgroup order: a fixed summation order
        const int64_t b0 = sk_bound(j2, U, G);
        if (b0 >= te) break;
        const int which = (b0 >= tb) ? 0 : 1;                 // the piece is the first of workgroup j2's range iff the range starts inside the tile
        const float* __restrict__ slab = p.sk_ws + ((int64_t)2 * j2 + which) * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(slab + ((((i * TN + j) * 4 + (r >> 2)) * 256 + tid) << 2));
                    acc[i][j][r] += v[0]; acc[i][j][r + 1] += v[1]; acc[i][j][r + 2] += v[2]; acc[i][j][r + 3] += v[3];
                }
    }
    const uint32_t tiles_n = (p.N + BN - 1) / BN;
    const int z = t / p.sk_tiles_mn, tile = t - z * p.sk_tiles_mn;
    const int zo = z / p.Zi, zi = z - zo * p.Zi;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    gemm_epilogue<BM, BN, WM, WN>(p, p.C + zo * p.sC_o + zi * p.sC_i, lds, m0, n0, acc, zo);
}

#ifndef IGEMM_PROBE_ONLY      // tools/probe/ks_probe.hip compiles single kernel instantiations of this file (register reports in seconds)
// ---- host side ---------------------------------------------------------------------------------------------------
// Stream-K plan of one launch: G persistent workgroups (0 = classic one-tile-per-workgroup launch).
// Classic when the grid already fills the chip many times (ceil() loss < ~6 %) or when there is too little K to split.
template <typename K>
static int resident_per_cu(K kernel) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(kernel), 256, 0) != hipSuccess || n < 1) {
        (void)hipGetLastError();
        n = 2;
    }
    return n > 4 ? 4 : n;
}
static int num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); cus = 256; }
        else cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return cus;
}
static int sk_plan(int64_t tiles_total, int nk, int slots) {
    if (tiles_total >= 8 * (int64_t)slots) return 0;          // >= 8 full waves: at most 1/8 of a wave is lost to the ceil()
    const int64_t waves = (tiles_total + slots - 1) / slots;
    if (tiles_total * 100 >= waves * slots * 92) return 0;    // the one-tile-per-workgroup grid already keeps >= 92 % of the slots busy
    const int64_t U = tiles_total * nk;
    int64_t G = slots;
    if (U < 8 * G) G = U / 8;                                 // at least 8 K tiles of work per workgroup
    if (G < 2 || (tiles_total <= slots && G <= tiles_total)) return 0;      // too little K to share out: splitting would not add parallelism
    return (int)G;
}

extern thread_local int g_plan_split;     // set by the plan / launch: did the chosen tile run the split-bf16 main loop (corrif_gemm_fwd_is_split); defined in igemm.hip
// One instantiation = the kernels of one (tile, loader, B layout) combination in all their main-loop modes.  Not static: the implicit-
// convolution half (GEMM = false) is instantiated in igemm_conv.hip, the plain-GEMM half in igemm.hip, so that the two halves compile in
// parallel (this family is 3/4 of the library's build time).
template <int BM, int BN, int WM, int WN, int VEC, bool GEMM, int BL>
int launch_variant(GemmArgs& a, int Z, hipStream_t s, bool plan_only, size_t* ws_bytes) {
    // stream-K exists for the two tiles that carry the encoder's shapes (128x128, 128x64, float4 loader); the split-bf16 main loop for
    // those and the 64x64 tile - every tile the N > 32 tile choice can land on, which depends on the launch's tile count and therefore on
    // Z: a grouped launch (Z = 3) and its three per-modality twins must run the same arithmetic to stay bit-identical
    constexpr bool BIG = VEC == 4 && BM == 128 && (BN == 128 || BN == 64);
    constexpr bool SPL = VEC == 4 && (BM == 128 || BM == 64) && (BN == 128 || BN == 64);
    const int64_t tiles_mn = (int64_t)((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const int nk = (a.K + BK - 1) / BK;
    const bool split = SPL && !a.f32_mfma;
    g_plan_split = split ? 1 : 0;
    int G = 0;
    if constexpr (BIG) {
        static int per_cu[2] = {0, 0};
        if (!per_cu[0]) per_cu[0] = resident_per_cu(gemm_sk_kernel<BM, BN, WM, WN, VEC, GEMM, BL, 0>);
        if (!per_cu[1]) per_cu[1] = resident_per_cu(gemm_sk_kernel<BM, BN, WM, WN, VEC, GEMM, BL, 1>);
        G = a.sk_allowed ? sk_plan(tiles_mn * Z, nk, num_cus() * per_cu[split ? 1 : 0]) : 0;
    }
    if (ws_bytes) *ws_bytes = G ? (size_t)2 * G * BM * BN * sizeof(float) : 0;
    if (plan_only) return CORRIF_OK;
    if (G && !a.sk_ws) return CORRIF_EINVAL;                  // the caller did not provide the workspace corrif_gemm_fwd_workspace asked for
    if (!G) {
        dim3 grid((uint32_t)tiles_mn, 1, Z);
        if constexpr (SPL) {
            if (split) {
                hipLaunchKernelGGL((gemm_fwd_kernel<BM, BN, WM, WN, VEC, GEMM, BL, 2>), grid, dim3(256), 0, s, a);
                CORRIF_CHECK_LAUNCH();
                return CORRIF_OK;
            }
        }
        hipLaunchKernelGGL((gemm_fwd_kernel<BM, BN, WM, WN, VEC, GEMM, BL, 0>), grid, dim3(256), 0, s, a);
        CORRIF_CHECK_LAUNCH();
        return CORRIF_OK;
    }
    if constexpr (BIG) {
        a.sk_G = G; a.sk_nk = nk; a.sk_tiles_mn = (int)tiles_mn; a.sk_tiles = (int)(tiles_mn * Z);
        if (split) hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, WM, WN, VEC, GEMM, BL, 1>), dim3(G), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gemm_sk_kernel<BM, BN, WM, WN, VEC, GEMM, BL, 0>), dim3(G), dim3(256), 0, s, a);
        CORRIF_CHECK_LAUNCH();
        hipLaunchKernelGGL((gemm_sk_fixup_kernel<BM, BN, WM, WN>), dim3(G - 1), dim3(256), 0, s, a);
        CORRIF_CHECK_LAUNCH();
    }
    return CORRIF_OK;
}

#endif  // IGEMM_PROBE_ONLY
