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

static thread_local int g_plan_split = 0;      // set by the plan / launch: did the chosen tile run the split-bf16 main loop (corrif_gemm_fwd_is_split)
template <int BM, int BN, int WM, int WN, int VEC, bool GEMM, int BL>
static int launch_variant(GemmArgs& a, int Z, hipStream_t s, bool plan_only, size_t* ws_bytes) {
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

template <int BM, int BN, int WM, int WN, int VEC = 4>
static int launch_fwd(GemmArgs& a, int Z, hipStream_t s, bool plan_only, size_t* ws_bytes) {
    if constexpr (VEC == 1) {
        return launch_variant<BM, BN, WM, WN, 1, false, 0>(a, Z, s, plan_only, ws_bytes);
    } else if (a.g.is_gemm) {
        if (a.b_layout == 0) return launch_variant<BM, BN, WM, WN, VEC, true, 0>(a, Z, s, plan_only, ws_bytes);
        return launch_variant<BM, BN, WM, WN, VEC, true, 1>(a, Z, s, plan_only, ws_bytes);
    } else {
        if (a.b_layout == 0) return launch_variant<BM, BN, WM, WN, VEC, false, 0>(a, Z, s, plan_only, ws_bytes);
        return launch_variant<BM, BN, WM, WN, VEC, false, 1>(a, Z, s, plan_only, ws_bytes);
    }
}

// validation + tile choice + launch (plan_only: only report the stream-K workspace the launch would need)
static int gemm_fwd_impl(const CorrifGemm* p, void* stream, bool plan_only, size_t* ws_bytes) {
    if (ws_bytes) *ws_bytes = 0;
    if (!p || !p->A || !p->B || !p->C) return CORRIF_EINVAL;
    if (p->M <= 0 || p->N <= 0 || p->K <= 0 || p->Z < 1 || p->Zi < 1) return CORRIF_EINVAL;
    const bool scalar = p->Cs == 1;
    if (p->Cs <= 0 || (p->K & 3) || (p->ldb & 3)) return CORRIF_EUNSUPPORTED;
    if (!scalar && ((p->Cs & 3) || (p->lda & 3) || ((uintptr_t)p->A & 15) || (p->sA_o & 3) || (p->sA_i & 3) ||
                    (p->g.src_batch_pitch & 3)))
        return CORRIF_EUNSUPPORTED;
    if (scalar && (p->g.is_gemm || p->b_layout != 0)) return CORRIF_EUNSUPPORTED;
    if (((uintptr_t)p->B & 15)) return CORRIF_EUNSUPPORTED;
    if ((p->sB_o & 3) || (p->sB_i & 3)) return CORRIF_EUNSUPPORTED;
    if (p->b_layout == 1 && (p->N & 3)) return CORRIF_EUNSUPPORTED;
    if (p->b_layout != 0 && p->b_layout != 1) return CORRIF_EINVAL;
    if (!geom_ok(p->g)) return CORRIF_EINVAL;
    if (!p->g.is_gemm && (int64_t)p->g.Sd * p->g.Sh * p->g.Sw * p->lda >= (int64_t)1 << 31) return CORRIF_EUNSUPPORTED;   // 32-bit in-sample offsets
    if (p->ntap_sel < 0 || p->ntap_sel > 28 || (p->ntap_sel && (p->g.is_gemm || scalar))) return CORRIF_EINVAL;
    if (p->out_map && (p->g.is_gemm || p->addend || p->addend2 || p->N > 4096 * 1024)) return CORRIF_EINVAL;
    if (p->addend2 && !p->addend) return CORRIF_EINVAL;
    if (!p->g.is_gemm && !scalar && p->K != (p->ntap_sel ? p->ntap_sel : p->g.kd * p->g.kh * p->g.kw) * p->Cs) return CORRIF_EINVAL;
    if (scalar && (p->g.ntaps <= 0 || p->g.ntaps > p->g.kd * p->g.kh * p->g.kw || p->K < p->g.ntaps)) return CORRIF_EINVAL;
    if (p->Z > 65535) return CORRIF_EUNSUPPORTED;
    if (p->ws && ((uintptr_t)p->ws & 15)) return CORRIF_EUNSUPPORTED;
    GemmArgs a;
    a.A = p->A; a.B = p->B; a.C = p->C; a.bias = p->bias; a.addend = p->addend; a.addend2 = p->addend2; a.ld_add2 = p->ld_add2;
    a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc; a.ld_add = p->ld_add;
    a.M = p->M; a.N = p->N; a.K = p->K; a.Cs4 = p->Cs / 4; a.act = p->act; a.b_layout = p->b_layout; a.Zi = p->Zi;
    a.sA_o = p->sA_o; a.sA_i = p->sA_i; a.sB_o = p->sB_o; a.sB_i = p->sB_i; a.sC_o = p->sC_o; a.sC_i = p->sC_i;
    a.g = make_devgeom(p->g, p->lda);
    a.ntap_sel = p->ntap_sel;
    for (int i = 0; i < 28; ++i) a.tap_sel[i] = p->tap_sel[i];
    a.stats_part = p->stats_part; a.stats_relu = p->stats_relu; a.stats_rpg = 1; a.stats_chunks = 1;
    a.bs_x = p->bstats_x; a.bs_ldx = p->bstats_ldx; a.bs_y = p->bstats_y; a.bs_ldy = p->bstats_ldy; a.bs_mean = p->bstats_mean; a.bs_rstd = p->bstats_rstd;
    if (p->bstats_x) {
        if (!p->stats_part || !p->bstats_mean || !p->bstats_rstd || p->bstats_ldx < p->N || (p->bstats_y && p->bstats_ldy < p->N)) return CORRIF_EINVAL;
        if ((p->bstats_ldx & 3) || (p->bstats_ldy & 3) || ((uintptr_t)p->bstats_x & 15) || ((uintptr_t)p->bstats_y & 15)) return CORRIF_EUNSUPPORTED;
    }
    a.zs_bias = p->zs_bias; a.zs_add = p->zs_add; a.zs_add2 = p->zs_add2; a.zs_stats = p->zs_stats; a.zs_bsx = p->zs_bsx;
    a.zs_bsy = p->zs_bsy; a.zs_bsstat = p->zs_bsstat;
    if ((p->zs_add & 3) || (p->zs_add2 & 3) || (p->zs_bsx & 3) || (p->zs_bsy & 3)) return CORRIF_EUNSUPPORTED;
    if (p->stats_part) {
        if ((p->Z != 1 && (p->Zi != 1 || p->stats_rows_per_group != p->M)) || p->stats_rows_per_group <= 0 || p->stats_rows_per_group > p->M || (p->M % p->stats_rows_per_group)) return CORRIF_EINVAL;
        if (p->stats_rows_per_group != p->M && (p->stats_rows_per_group & 63)) return CORRIF_EUNSUPPORTED;
        if (scalar || p->out_map || p->N <= 16) return CORRIF_EUNSUPPORTED;     // only the TM = 2 tiles of gemm_fwd_kernel produce them
        a.stats_rpg = (int)p->stats_rows_per_group;
        a.stats_chunks = (int)((p->stats_rows_per_group + 63) / 64);
    }
    a.out_map = p->out_map; a.OD = p->OD; a.OH = p->OH; a.OW = p->OW;
    a.om_d = p->om_d; a.om_h = p->om_h; a.om_w = p->om_w; a.oo_d = p->oo_d; a.oo_h = p->oo_h; a.oo_w = p->oo_w;
    a.sk_ws = p->ws; a.sk_G = 0; a.sk_nk = 0; a.sk_tiles = 0; a.sk_tiles_mn = 0;
    a.sk_allowed = p->no_split ? 0 : 1;
    a.f32_mfma = p->f32_mfma ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const int Z = p->Z;
    if (scalar) return launch_fwd<128, 64, 2, 2, 1>(a, Z, s, plan_only, ws_bytes);
    if (p->ntap_sel || p->out_map) {       // parity-class data gradient: only the 32x32-MFMA tiles implement these options
        if (p->N <= 32) return launch_fwd<256, 32, 4, 1>(a, Z, s, plan_only, ws_bytes);
        if (p->N <= 64) return launch_fwd<128, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
        return launch_fwd<128, 128, 2, 2>(a, Z, s, plan_only, ws_bytes);
    }
    if (p->N <= 16 && !p->addend2 && !(p->N & 3) && !(p->ldc & 3) && !((uintptr_t)p->C & 15) && !(p->sC_o & 3) && !(p->sC_i & 3) &&
        (!p->addend || (!(p->ld_add & 3) && !((uintptr_t)p->addend & 15)))) {
        if (plan_only) return CORRIF_OK;
        return launch_smalln_fwd(a, Z, s);
    }
    if (p->N <= 32) return launch_fwd<256, 32, 4, 1>(a, Z, s, plan_only, ws_bytes);
    // 128-wide tiles measured faster than 64x64 even at ~1.5 workgroups per CU (e4: M=25088,N=256,K=2304: 70 vs 56 TF/s);
    // shrink only when the grid could not even cover the 256 CUs once.  (With the stream-K split the tile count no longer has to
    // cover the chip: the K loop is what is shared out.)
    auto tiles = [&](int bm, int bn) { return (int64_t)((p->M + bm - 1) / bm) * ((p->N + bn - 1) / bn) * p->Z; };
    const int nk = (p->K + BK - 1) / BK;
    const bool st = p->stats_part != nullptr;       // fused statistics need 64-row wave blocks (TM = 2)
    const bool splittable = a.sk_allowed != 0;
    if (p->N <= 64) {
        if (st || tiles(128, 64) >= 192 || (splittable && tiles(128, 64) * nk >= 2048)) return launch_fwd<128, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
        return launch_fwd<64, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
    }
    if (tiles(128, 128) >= 192 || (splittable && tiles(128, 128) * nk >= 2048)) return launch_fwd<128, 128, 2, 2>(a, Z, s, plan_only, ws_bytes);
    if (st || tiles(128, 64) >= 192) return launch_fwd<128, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
    return launch_fwd<64, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
}

extern "C" int corrif_gemm_fwd(const CorrifGemm* p, void* stream) { return gemm_fwd_impl(p, stream, false, nullptr); }

extern "C" int corrif_gemm_fwd_is_split(const CorrifGemm* p) {
    size_t n = 0;
    g_plan_split = 0;
    if (gemm_fwd_impl(p, nullptr, true, &n) != CORRIF_OK) return 0;
    return g_plan_split;
}

extern "C" size_t corrif_gemm_fwd_workspace(const CorrifGemm* p) {
    size_t n = 0;
    if (gemm_fwd_impl(p, nullptr, true, &n) != CORRIF_OK) return 0;
    return n;
}

#endif  // IGEMM_PROBE_ONLY

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
