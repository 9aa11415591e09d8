// Patch-staged direct 3x3x3 convolution for narrow layers (Cout <= 32): the decoder's 16^3..128^3 stages.
//
// Why: as an implicit GEMM these layers re-fetch every input voxel 27 times through L2/L1 (one K tile per tap) and were
// bound by that traffic (6 TB/s of L2->LDS at 20-25 TF/s), not by the matrix cores.  Here a workgroup stages the halo'd
// input patch of its 256-voxel output tile (e.g. 4x4x16 -> 6x6x18 voxels) in LDS ONCE per channel chunk and every tap is
// an LDS read at a compile-time offset: global traffic drops from 27x to ~2.5x the input, the per-tap gather arithmetic
// disappears, replicate / zero padding is resolved while staging.
//
// Matrix core use: v_mfma_f32_4x4x1_16b_f32 with the A-operand broadcast (CBSZ=4, ABID=b): one VGPR holds the weights of
// 4 output channels for 16 different k (lane 4b+i = W[co i][k b]); MFMA #b broadcasts block b's weights to all 16 blocks,
// while every lane supplies its own voxel's activation -> 64 voxels x 4 channels x 1 k per instruction, no padding waste
// at Cout = 8, and one ds_read_b128 of weights feeds 64 MFMAs.  Lane l ends with out[voxel l][4g..4g+3] in one VGPR quad.
//
// The same kernel is the data gradient (weights flipped/transposed by the re-layout, pad = 2 on the (n+2)^3 grid for
// replicate padding, pad = 1 for zero padding).
#include <utility>
#include "common.h"

struct PatchArgs {
    const float* X; int64_t ldx;
    const float* Wp;                 // [nchunk][Co][27*CC]
    float* Y; int64_t ldy;
    const float* bias;
    int B, Sd, Sh, Sw, Od, Oh, Ow, Ci, Co, pad, clamp, ntiles, wres, fold, patch_floats;
    int ltd, lth, ltw, ntd, nth, ntw;
    FastDiv dHW, dW;                 // patch voxel index -> (pd, ph, pw)
    FastDiv dT0, dT1, dT2;           // tile index -> (b, td, th, tw)
    // fused epilogue extras (forward of a general_conv3d_prenorm layer, mmvit4.py:41-45):
    //   add_src: y += add_src[b, cls(d), h, w, :]  - the compact skip branch's share of d*_c2 (depth classes, see common.h) - the separate
    //            depth_bcast_add pass over the full-depth tensor disappears;
    //   stats:   per (sample, channel) sum / sum of squares of max(y, 0) for the InstanceNorm that follows, partial per (workgroup, wave)
    //            slot: stats[((b*Co + c)*stats_chunks + slot*4 + wave)*2 + {0,1}] (pre-zeroed; corrif_norm_stats_finalize reads it) -
    //            the separate statistics pass over the conv output disappears.
    const float* add_src; int64_t ld_add; DAxis add_ax;
    double* stats; int stats_chunks, stats_relu, tiles_per_sample;
};

template <int NG, int CC>
struct PatchCfg {
    static constexpr int KC = 27 * CC;
    static constexpr int KCP = (KC + 63) / 64 * 64;
    static constexpr int WP = KCP + 4;         // weight row pitch (floats)
    static constexpr int CP = CC + 4;          // patch voxel pitch (floats)
    static constexpr int NKG = KCP / 64;
};

template <int NG>
struct PatchAcc { static constexpr int NA = NG; };

// One step = 4 consecutive k (one float4 of this lane's voxel at a compile-time tap/channel offset) x NG output-channel groups.
// The patch reads run TWO steps ahead of the MFMAs that consume them and the weight registers of the next 64-k group are
// requested half a group early: a read -> wait -> 4 NG MFMAs chain exposes the LDS latency every 32 NG cycles.
template <int CC, int T>
__device__ __forceinline__ f32x4 patch_x4(const float* __restrict__ patch, int pvoff, int PH, int PW) {
    constexpr int kidx = T * 4, tap = kidx / CC, c = kidx % CC;
    constexpr int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
    return *reinterpret_cast<const f32x4*>(patch + (pvoff + (td * PH + th) * PW + tw) * (CC + 4) + c);
}
template <int NG, int CC, int KG>
__device__ __forceinline__ void patch_wload(f32x4 (&wreg)[NG], const float* __restrict__ wl, int lane) {
    using Cfg = PatchCfg<NG, CC>;
#pragma unroll
    for (int g = 0; g < NG; ++g) wreg[g] = *reinterpret_cast<const f32x4*>(wl + (g * 4 + (lane & 3)) * Cfg::WP + KG * 64 + (lane >> 2) * 4);
}
template <int NG, int CC, int T>
__device__ __forceinline__ void patch_step(f32x4 (&acc)[NG], f32x4 (&wreg)[2][NG], f32x4 (&xb)[3], const float* __restrict__ wl,
                                           const float* __restrict__ patch, int pvoff, int PH, int PW, int lane) {
    constexpr int NSTEP = 27 * CC / 4, KG = T / 16, Bk = T % 16;
    if constexpr (T + 2 < NSTEP) xb[(T + 2) % 3] = patch_x4<CC, T + 2>(patch, pvoff, PH, PW);
    if constexpr (Bk == 8 && (KG + 1) * 16 < NSTEP) patch_wload<NG, CC, KG + 1>(wreg[(KG + 1) & 1], wl, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[KG & 1][g][e], xb[T % 3][e], acc[g], 4, Bk, 0);
}
template <int NG, int CC, int... Ts>
__device__ __forceinline__ void patch_chunk_seq(f32x4 (&acc)[NG], const float* __restrict__ wl, const float* __restrict__ patch, int pvoff,
                                                int PH, int PW, int lane, std::integer_sequence<int, Ts...>) {
    f32x4 wreg[2][NG], xb[3];
    patch_wload<NG, CC, 0>(wreg[0], wl, lane);
    xb[0] = patch_x4<CC, 0>(patch, pvoff, PH, PW);
    xb[1] = patch_x4<CC, 1>(patch, pvoff, PH, PW);
    (patch_step<NG, CC, Ts>(acc, wreg, xb, wl, patch, pvoff, PH, PW, lane), ...);
}
template <int NG, int CC>
__device__ __forceinline__ void patch_chunk(f32x4 (&acc)[NG], const float* __restrict__ wl, const float* __restrict__ patch, int pvoff,
                                            int PH, int PW, int lane) {
    patch_chunk_seq<NG, CC>(acc, wl, patch, pvoff, PH, PW, lane, std::make_integer_sequence<int, 27 * CC / 4>{});
}

// Persistent workgroups walk a contiguous range of tiles (neighbouring tiles share their halo through L2).  The work list
// is the sequence of (tile, channel chunk) items; while the MFMAs of item i run, the global loads of item i+1 (halo'd patch
// and, when there is more than one chunk, its weights) are already in flight into registers, so staging costs only the LDS
// stores and two barriers instead of exposed HBM/L2 round trips.  Loads are unconditional from the clamped voxel (always
// mapped) and zero padding is applied at the LDS store: a load under a branch makes the compiler drain vmcnt at the join.
// NPF = halo float4 per thread: ceil(648 * CC/4 / 256) for the usual 4 x 4 x 16 tile, ceil(972 * ...) for the shallow-grid tiles.
// ST: the variant with the fused InstanceNorm statistics (forward of the 8- / 16-channel layers at 128^3 / 64^3, where the separate
// statistics pass costs most); its accumulators must not cost the data-gradient launches of the same kernel their registers.
template <int NG, int CC, int NPF, bool ST = false>
__global__ __launch_bounds__(256, ST ? 2 : 1) void conv3_patch_kernel(PatchArgs p) {
    using Cfg = PatchCfg<NG, CC>;
    constexpr int C4 = CC / 4;
    constexpr int K4 = Cfg::KC / 4;
    constexpr int NWF = (4 * NG * K4 + 255) / 256;        // weight float4 per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TD = 1 << p.ltd, TH = 1 << p.lth, TW = 1 << p.ltw;
    const int PD = TD + 2, PH = TH + 2, PW = TW + 2;
    const int NPV = PD * PH * PW;
    float* wl = smem;                                   // [4*NG][WP]
    float* patch = smem + (p.wres ? p.Ci / CC : 1) * 4 * NG * Cfg::WP;      // [NPV][CP]
    float* biasl = patch + p.patch_floats;              // [4*NG]; patch_floats = max(NPV * CP, fold staging 256 * (4 NG + 4))

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 4 * NG) biasl[tid] = (p.bias && tid < p.Co) ? p.bias[tid] : 0.f;
    // this lane's voxel inside the tile
    const int lv = wave * 64 + lane;
    const int vw = lv & (TW - 1), vh = (lv >> p.ltw) & (TH - 1), vd = lv >> (p.ltw + p.lth);
    const int pvoff = (vd * PH + vh) * PW + vw;

    // tile-independent staging roles of this thread: patch item u -> halo voxel (pd, ph, pw), channel quad, LDS offset
    const int total = NPV * C4;
    uint32_t pcoord[NPF];
    int pdst[NPF];
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int i = tid + u * 256;
        const int ic = min(i, total - 1);
        const int pv = ic / C4, c4 = ic - pv * C4;
        const uint32_t pd = fdiv((uint32_t)pv, p.dHW);
        const uint32_t rem = pv - pd * p.dHW.d;
        const uint32_t ph = fdiv(rem, p.dW);
        const uint32_t pw = rem - ph * p.dW.d;
        pcoord[u] = (pd << 24) | (ph << 16) | (pw << 8) | (uint32_t)c4;
        pdst[u] = i < total ? pv * Cfg::CP + c4 * 4 : -1;
    }
    const int nw = p.Co * K4;                            // weight float4 of one chunk
    const int nchunk = p.Ci / CC;
    const int nwl = p.wres ? nchunk : 1;                 // weight chunks resident in LDS (all of them when they fit beside the patch)

    // zero the weight rows/tails that are never overwritten
    for (int i = tid; i < nwl * 4 * NG * Cfg::WP; i += 256) wl[i] = 0.f;
    if (p.wres) {
        __syncthreads();
        for (int c = 0; c < nchunk; ++c)
            for (int i = tid; i < nw; i += 256) {
                const int co = i / K4, kk = i - co * K4;
                *reinterpret_cast<f32x4*>(wl + (c * 4 * NG + co) * Cfg::WP + kk * 4) =
                    *reinterpret_cast<const f32x4*>(p.Wp + ((int64_t)c * p.Co * K4 + i) * 4);
            }
    }

    const int per = (p.ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)blockIdx.x * per, t_end = min(p.ntiles, t_begin + per);
    const int nitems = max(t_end - t_begin, 0) * nchunk;

    f32x4 pre[NPF], wpre[NWF];
    uint32_t zmask = 0;
    auto issue = [&](int tile, int ch, bool with_w) {
        uint32_t t = (uint32_t)tile;
        const uint32_t b = fdiv(t, p.dT0);
        t -= b * p.dT0.d;
        const uint32_t itd = fdiv(t, p.dT1);
        t -= itd * p.dT1.d;
        const uint32_t ith = fdiv(t, p.dT2);
        const uint32_t itw = t - ith * p.dT2.d;
        const int od0 = (int)(itd << p.ltd) - p.pad, oh0 = (int)(ith << p.lth) - p.pad, ow0 = (int)(itw << p.ltw) - p.pad;
        const float* __restrict__ Xb = p.X + (int64_t)b * p.Sd * p.Sh * p.Sw * p.ldx + ch * CC;
        zmask = 0;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int sd = od0 + (int)(pcoord[u] >> 24), sh = oh0 + (int)((pcoord[u] >> 16) & 255), sw = ow0 + (int)((pcoord[u] >> 8) & 255);
            const int cd = min(max(sd, 0), p.Sd - 1), chh = min(max(sh, 0), p.Sh - 1), cw = min(max(sw, 0), p.Sw - 1);
            zmask |= (uint32_t)(!p.clamp && (cd != sd || chh != sh || cw != sw)) << u;
            pre[u] = *reinterpret_cast<const f32x4*>(Xb + ((int64_t)(cd * p.Sh + chh) * p.Sw + cw) * p.ldx + (pcoord[u] & 255) * 4);
        }
        if (with_w) {
            const float* __restrict__ wsrc = p.Wp + (int64_t)ch * p.Co * Cfg::KC;
#pragma unroll
            for (int u = 0; u < NWF; ++u) wpre[u] = *reinterpret_cast<const f32x4*>(wsrc + (int64_t)min(tid + u * 256, nw - 1) * 4);
        }
    };

    constexpr int NA = PatchAcc<NG>::NA;
    f32x4 acc[NA];
#pragma unroll
    for (int g = 0; g < NA; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fused InstanceNorm statistics: fp32 per-lane running sums over this workgroup's tiles of ONE sample (<= a few hundred values per
    // lane; the lane sums' rounding errors are independent and average out over the 256 lanes x workgroups that are then combined in
    // double), flushed whenever the sample changes and at the end
    constexpr int NS = ST ? NG : 1;
    f32x4 s1[NS], s2[NS];
#pragma unroll
    for (int g = 0; g < NS; ++g) { s1[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; s2[g] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    int stat_b = -1;
    auto flush_stats = [&]() {
        if (!ST || stat_b < 0) return;
        const int slot = (int)blockIdx.x - (int)(((int64_t)stat_b * p.tiles_per_sample) / per);
#pragma unroll
        for (int g = 0; g < NS; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double a = (double)s1[g][e], q = (double)s2[g][e];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
                if (lane == 0 && g * 4 + e < p.Co) {
                    double* o2 = p.stats + ((((int64_t)stat_b * p.Co + g * 4 + e) * p.stats_chunks) + slot * 4 + wave) * 2;
                    o2[0] = a;
                    o2[1] = q;
                }
            }
            s1[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            s2[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };

    if (nitems > 0) issue(t_begin, 0, !p.wres);
    int tile = t_begin, ch = 0;
    for (int it = 0; it < nitems; ++it) {
        __syncthreads();                                 // the previous item's readers are done (and the zero fill is visible)
#pragma unroll
        for (int u = 0; u < NPF; ++u)
            if (pdst[u] >= 0) *reinterpret_cast<f32x4*>(patch + pdst[u]) = (zmask >> u) & 1 ? (f32x4){0.f, 0.f, 0.f, 0.f} : pre[u];
        if (!p.wres && (it == 0 || nchunk > 1)) {        // a single chunk's weights stay resident for all tiles
#pragma unroll
            for (int u = 0; u < NWF; ++u) {
                const int i = tid + u * 256;
                if (i < nw) {
                    const int co = i / K4, kk = i - co * K4;
                    *reinterpret_cast<f32x4*>(wl + co * Cfg::WP + kk * 4) = wpre[u];
                }
            }
        }
        __syncthreads();
        int ntile = tile, nch = ch + 1;
        if (nch == nchunk) { nch = 0; ++ntile; }
        if (it + 1 < nitems) issue(ntile, nch, !p.wres && nchunk > 1);      // in flight during the MFMAs below
        patch_chunk<NG, CC>(acc, wl + (p.wres ? ch : 0) * 4 * NG * Cfg::WP, patch, pvoff, PH, PW, lane);
        if (ch == nchunk - 1) {
            uint32_t t = (uint32_t)tile;
            const uint32_t b = fdiv(t, p.dT0);
            t -= b * p.dT0.d;
            const uint32_t itd = fdiv(t, p.dT1);
            t -= itd * p.dT1.d;
            const uint32_t ith = fdiv(t, p.dT2);
            const uint32_t itw = t - ith * p.dT2.d;
            const int od = (int)(itd << p.ltd) + vd, oh = (int)(ith << p.lth) + vh, ow = (int)(itw << p.ltw) + vw;
            if (!p.fold) {
                if constexpr (ST) {
                    if ((int)b != stat_b) {                   // uniform per workgroup: the tile's sample index
                        flush_stats();
                        stat_b = (int)b;
                    }
                }
                if (od < p.Od && oh < p.Oh && ow < p.Ow) {
                    const int64_t row = (((int64_t)b * p.Od + od) * p.Oh + oh) * p.Ow + ow;
                    const float* __restrict__ addrow = nullptr;
                    if (p.add_src)
                        addrow = p.add_src + ((((int64_t)b * 3 * p.add_ax.in + depth_class(od, p.add_ax)) * p.Oh + oh) * p.Ow + ow) * p.ld_add;
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (g * 4 >= p.Co) continue;
                        f32x4 v = acc[g];
                        v += *reinterpret_cast<const f32x4*>(biasl + g * 4);       // LDS copy: a global load here would wait for the prefetch in flight
                        if (addrow) v += *reinterpret_cast<const f32x4*>(addrow + g * 4);
                        *reinterpret_cast<f32x4*>(p.Y + row * p.ldy + g * 4) = v;
                        if constexpr (ST) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float t = p.stats_relu ? fmaxf(v[e], 0.f) : v[e];
                                s1[g][e] += t;
                                s2[g][e] = fmaf(t, t, s2[g][e]);
                            }
                        }
                    }
                }
            } else {
                // Adjoint of replicate padding folded into the epilogue: this tile lives on the (n+2)^3 padded grid; padded
                // coordinate 0 belongs to voxel 1 and n+1 to voxel n.  n is a multiple of the tile extent on every axis (host
                // check), so both members of a pair sit in the SAME tile: the tile's results go through LDS once and every real
                // voxel sums its up to 2x2x2 pre-images in a fixed order (deterministic, no (n+2)^3 buffer, no pad_fold pass).
                constexpr int RP = 4 * NG + 4;
                float* res = patch;                                  // the patch is dead until the next item's staging
                __syncthreads();
#pragma unroll
                for (int g = 0; g < NG; ++g) *reinterpret_cast<f32x4*>(res + lv * RP + g * 4) = acc[g];
                __syncthreads();
                const int nd = p.Od - 2, nh = p.Oh - 2, nw = p.Ow - 2;
                if (od >= 1 && od <= nd && oh >= 1 && oh <= nh && ow >= 1 && ow <= nw) {
                    const int d_lo = od == 1 ? -1 : 0, d_hi = od == nd ? 1 : 0;
                    const int h_lo = oh == 1 ? -1 : 0, h_hi = oh == nh ? 1 : 0;
                    const int w_lo = ow == 1 ? -1 : 0, w_hi = ow == nw ? 1 : 0;
                    const int64_t row = (((int64_t)b * nd + (od - 1)) * nh + (oh - 1)) * nw + (ow - 1);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (g * 4 >= p.Co) continue;
                        f32x4 v = {0.f, 0.f, 0.f, 0.f};
                        for (int dd = d_lo; dd <= d_hi; ++dd)
                            for (int hh = h_lo; hh <= h_hi; ++hh)
                                for (int ww = w_lo; ww <= w_hi; ++ww)
                                    v += *reinterpret_cast<const f32x4*>(res + (lv + ((dd << p.lth) + hh) * TW + ww) * RP + g * 4);
                        *reinterpret_cast<f32x4*>(p.Y + row * p.ldy + g * 4) = v;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < NA; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        tile = ntile;
        ch = nch;
    }
    if constexpr (ST) flush_stats();
}

extern "C" int corrif_conv3_patch_cc(int32_t Ci, int32_t Co);
template <int NG, int CC>
static constexpr bool patch_stats_variant() { return (NG == 2 && (CC == 16 || CC == 8)) || (NG == 4 && CC == 8); }
extern "C" int corrif_conv3_patch_stats_supported(int32_t Ci, int32_t Co) {
    const int cc = corrif_conv3_patch_cc(Ci, Co);
    return cc != 0 && (Co == 8 || (Co == 16 && cc == 8));
}

template <int NG, int CC, int NPF, bool ST>
static int launch_patch_npf(const PatchArgs& a, unsigned tiles, size_t lds, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_patch_kernel<NG, CC, NPF, ST>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return CORRIF_ELAUNCH;
        }
        attr_done = true;
    }
    const unsigned per_cu = lds > 80 * 1024 ? 1 : 2;                     // co-resident workgroups per CU (LDS bound)
    const unsigned grid = tiles < 256 * per_cu ? tiles : 256 * per_cu;
    if (a.stats) {      // the partial-slot layout the caller sized with corrif_conv3_patch_stats_chunks assumes at most 512 workgroups
        const int64_t per = ((int64_t)tiles + grid - 1) / grid;
        const int64_t nslots = (a.tiles_per_sample + per - 1) / per + 1;
        if (4 * nslots > a.stats_chunks) return CORRIF_EINVAL;
    }
    hipLaunchKernelGGL((conv3_patch_kernel<NG, CC, NPF, ST>), dim3(grid), dim3(256), lds, s, a);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
template <int NG, int CC>
static int launch_patch(const PatchArgs& a, unsigned tiles, int npv, hipStream_t s) {
    using Cfg = PatchCfg<NG, CC>;
    PatchArgs b = a;
    const int nchunk = a.Ci / CC;
    b.patch_floats = npv * Cfg::CP;
    if (a.fold && b.patch_floats < 256 * (4 * NG + 4)) b.patch_floats = 256 * (4 * NG + 4);
    const size_t fixed = (size_t)(b.patch_floats + 4 * NG) * sizeof(float), wchunk = (size_t)4 * NG * Cfg::WP * sizeof(float);
    b.wres = nchunk > 1 && fixed + nchunk * wchunk <= 80 * 1024;         // all weight chunks resident while two workgroups still share a CU
    const size_t lds = fixed + (b.wres ? nchunk : 1) * wchunk;
    if (lds > 160 * 1024) return CORRIF_EUNSUPPORTED;
    constexpr int C4 = CC / 4, NPF_STD = (648 * C4 + 255) / 256, NPF_MAX = (972 * C4 + 255) / 256;
    if (a.stats) {
        if constexpr (patch_stats_variant<NG, CC>()) {
            if (npv * C4 <= NPF_STD * 256) return launch_patch_npf<NG, CC, NPF_STD, true>(b, tiles, lds, s);
            if (npv * C4 <= NPF_MAX * 256) return launch_patch_npf<NG, CC, NPF_MAX, true>(b, tiles, lds, s);
        }
        return CORRIF_EUNSUPPORTED;
    }
    if (npv * C4 <= NPF_STD * 256) return launch_patch_npf<NG, CC, NPF_STD, false>(b, tiles, lds, s);
    if (npv * C4 <= NPF_MAX * 256) return launch_patch_npf<NG, CC, NPF_MAX, false>(b, tiles, lds, s);
    return CORRIF_EUNSUPPORTED;
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

extern "C" int corrif_conv3_patch_cc(int32_t Ci, int32_t Co) {
    if (Co <= 0 || Co > 32 || (Co & 3) || Ci <= 0 || (Ci & 7)) return 0;
    if (Ci > 64) return 0;      // many channel chunks re-stage the halo too often: the implicit GEMM is faster there (d3_c2: 50 vs 38 TF/s)
    return (Co <= 8 && !(Ci & 15)) ? 16 : 8;     // measured: 8-channel chunks (more co-resident workgroups) are not faster for Cout <= 8
}

// number of statistics partial slots per (sample, channel) a forward launch on this output grid may use (CorrifConv3Patch.stats_chunks):
// 4 waves x (workgroups that can touch one sample); the launch never uses more than 512 workgroups
extern "C" int corrif_conv3_patch_stats_chunks(int32_t B, int32_t Od, int32_t Oh, int32_t Ow) {
    if (B <= 0 || Od <= 0 || Oh <= 0 || Ow <= 0) return 0;
    int TD = 4, TH = 4, TW = 16;
    if (Od < 4) { TD = Od >= 2 ? 2 : 1; TH = 256 / (TD * TW); }
    const int64_t tps = (int64_t)((Od + TD - 1) / TD) * ((Oh + TH - 1) / TH) * ((Ow + TW - 1) / TW);
    const int64_t tiles = tps * B;
    int64_t worst = 0;
    for (int64_t grid = 256; grid <= 512; grid += 256) {       // 1 or 2 workgroups per CU, or one per tile when there are fewer tiles
        const int64_t g = tiles < grid ? tiles : grid;
        const int64_t per = (tiles + g - 1) / g;
        const int64_t n = (tps + per - 1) / per + 1;
        if (n > worst) worst = n;
    }
    return (int)(4 * worst);
}

extern "C" int corrif_conv3_patch(const CorrifConv3Patch* q, void* stream) {
    if (!q || !q->X || !q->Wp || !q->Y) return CORRIF_EINVAL;
    if (q->B <= 0 || q->Sd <= 0 || q->Sh <= 0 || q->Sw <= 0 || q->Od <= 0 || q->Oh <= 0 || q->Ow <= 0) return CORRIF_EINVAL;
    if (q->pad != 1 && q->pad != 2) return CORRIF_EINVAL;
    const int CC = corrif_conv3_patch_cc(q->Ci, q->Co);
    if (!CC || q->cc != CC) return CORRIF_EUNSUPPORTED;
    if ((q->ldx & 3) || (q->ldy & 3) || ((uintptr_t)q->X & 15) || ((uintptr_t)q->Y & 15) || ((uintptr_t)q->Wp & 15) ||
        ((uintptr_t)q->bias & 15))
        return CORRIF_EUNSUPPORTED;
    PatchArgs a;
    a.X = q->X; a.ldx = q->ldx; a.Wp = q->Wp; a.Y = q->Y; a.ldy = q->ldy; a.bias = q->bias;
    a.B = q->B; a.Sd = q->Sd; a.Sh = q->Sh; a.Sw = q->Sw; a.Od = q->Od; a.Oh = q->Oh; a.Ow = q->Ow;
    a.Ci = q->Ci; a.Co = q->Co; a.pad = q->pad; a.clamp = q->clamp; a.fold = q->fold;
    a.add_src = q->add_src; a.ld_add = q->ld_add; a.add_ax = make_daxis(q->add_Ds > 0 ? q->add_Ds : 1, q->Od);
    a.stats = q->stats_part; a.stats_chunks = q->stats_chunks; a.stats_relu = q->stats_relu; a.tiles_per_sample = 0;
    if (q->add_src && (q->fold || q->add_Ds < 1 || q->Od < 2 * q->add_Ds || (q->ld_add & 3) || q->ld_add < q->Co || ((uintptr_t)q->add_src & 15)))
        return CORRIF_EINVAL;
    if (q->stats_part && (q->fold || q->stats_chunks < 4)) return CORRIF_EINVAL;
    if (q->fold) {      // O = (n+2)^3 padded grid, Y = the n^3 gradient; pairs (0,1) / (n,n+1) must share a 4 x 4 x 16 tile
        if (q->pad != 2 || q->clamp || q->bias || q->Od < 6 || ((q->Od - 2) & 3) || ((q->Oh - 2) & 3) || ((q->Ow - 2) & 15)) return CORRIF_EUNSUPPORTED;
    }
    // 256-voxel tile: prefer 4 x 4 x 16; squeeze the depth for shallow grids
    int TD = 4, TH = 4, TW = 16;
    if (q->Od < 4) { TD = q->Od >= 2 ? 2 : 1; TH = 256 / (TD * TW); }
    a.ltd = ilog2(TD); a.lth = ilog2(TH); a.ltw = ilog2(TW);
    a.ntd = (q->Od + TD - 1) / TD; a.nth = (q->Oh + TH - 1) / TH; a.ntw = (q->Ow + TW - 1) / TW;
    const int PH = TH + 2, PW = TW + 2, PD = TD + 2;
    a.dHW = make_fastdiv((uint32_t)(PH * PW));
    a.dW = make_fastdiv((uint32_t)PW);
    a.dT0 = make_fastdiv((uint32_t)(a.ntd * a.nth * a.ntw));
    a.dT1 = make_fastdiv((uint32_t)(a.nth * a.ntw));
    a.dT2 = make_fastdiv((uint32_t)a.ntw);
    const int64_t tiles = (int64_t)q->B * a.ntd * a.nth * a.ntw;
    if (tiles <= 0 || tiles >= ((int64_t)1 << 31)) return CORRIF_EUNSUPPORTED;
    a.ntiles = (int)tiles;
    a.tiles_per_sample = a.ntd * a.nth * a.ntw;
    const int NG = (q->Co + 3) / 4;
    const int NPV = PD * PH * PW;
    hipStream_t s = (hipStream_t)stream;
#define PATCH_CASE(ng, cc) \
    if (NG == ng && CC == cc) return launch_patch<ng, cc>(a, (unsigned)tiles, NPV, s);
    PATCH_CASE(1, 16) PATCH_CASE(2, 16) PATCH_CASE(1, 8) PATCH_CASE(2, 8) PATCH_CASE(3, 8) PATCH_CASE(4, 8) PATCH_CASE(5, 8)
    PATCH_CASE(6, 8) PATCH_CASE(7, 8) PATCH_CASE(8, 8)
#undef PATCH_CASE
    return CORRIF_EUNSUPPORTED;
}

// ================================================================================================================
// Weight gradient of the same layers: dW[co][tap][ci] = sum_vox dY[vox][co] * Xpad[vox + tap][ci].
//
// Persistent workgroups (grid.x) walk the 256-voxel tiles; grid.z slices the input channels so that all outputs a
// workgroup owns (27 taps x 16*NCH channels x Co) stay in accumulator VGPRs for the whole kernel.  Per tile and 16-channel
// chunk the halo'd X patch and the dY tile are staged in LDS.  MFMA 4x4x1 with A-broadcast: lane l of a wave IS output
// column (tap, ci) = 64*cg + l and reads its own X value for the current voxel (one conflict-free ds_read_b32 at an
// immediate offset); the A register holds dY[16 voxels][4 co] and ABID selects the voxel, so every instruction adds
// one voxel's outer product dY[v][4 co] x X[v+tap][64 columns] into the SAME accumulators - no padding waste, no
// block partials.  Each wave takes 64 of the tile's voxels; waves write separate slabs, summed in a fixed order
// by slab_reduce (deterministic, no atomics).
// ================================================================================================================
struct PatchWgArgs {
    const float* X; int64_t ldx;
    const float* DY; int64_t lddy;
    float* ws;                       // [nslabs][Co][27][Ci]
    int B, Sd, Sh, Sw, Od, Oh, Ow, Ci, Co, pad, clamp, ntiles;
    int ltd, lth, ntd, nth, ntw;     // TW = 16 fixed
    FastDiv dHW, dW, dT0, dT1, dT2;
};

// half a w-row (8 voxels, ABID = HB*8 .. HB*8+7) of one column group
template <int NG, int HB, int... Vs>
__device__ __forceinline__ void wg_half(f32x4 (&a)[NG], const float (&areg)[NG], const float (&x)[8], std::integer_sequence<int, Vs...>) {
    ([&] {
#pragma unroll
        for (int g = 0; g < NG; ++g) a[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(areg[g], x[Vs], a[g], 4, HB * 8 + Vs, 0);
    }(), ...);
}
// 8 voxels of one w-row for this lane's column: xaddr includes the column offset, the w-row base and the half; voxel v sits CP floats on
template <int CP>
__device__ __forceinline__ void wg_load8(float (&x)[8], const float* __restrict__ xaddr) {
#pragma unroll
    for (int v = 0; v < 8; ++v) x[v] = xaddr[v * CP];
}

// CC = channels per chunk: 16 (7 column groups of 64 lanes for the 432 (tap, ci) columns) or 8 (4 groups for 216 columns: layers whose
// input channel count is not a multiple of 16 - the 8-channel d1_c1 share of d1_c2, the 24 skip channels on the compact grid).
template <int NG, int NCH, int NPF, int CC>
__global__ __launch_bounds__(256, 2) void conv3_patch_wgrad_kernel(PatchWgArgs p) {
    constexpr int CP = CC, TW = 16, PW = TW + 2, DP = 4 * NG + 4, KC = 27 * CC;   // CP = CC: taps 1 apart in w sit CC banks apart
    constexpr int C4 = CC / 4, NCG = (KC + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TD = 1 << p.ltd, TH = 1 << p.lth, PH = TH + 2;
    const int NPV = (TD + 2) * PH * PW;
    float* dys = smem;                      // [256][DP]
    float* patch = smem + 256 * DP;         // [NPV][CP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c_lo = blockIdx.z * (CC * NCH);

    // lane -> column (tap, ci) of column group cg: patch offset of the tap + channel
    int colbase[NCG];
    bool colok[NCG];
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
        const int col = cg * 64 + lane;
        colok[cg] = col < KC;
        const int tap = colok[cg] ? col / CC : 0, ci = colok[cg] ? col % CC : 0;
        const int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
        colbase[cg] = ((td * PH + th) * PW + tw) * CP + ci;
    }
    f32x4 acc[NCH][NCG][NG];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[c][cg][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // halo item i = tid + 256 u -> (halo voxel pv, channel quad c4); decoded on the fly (the accumulators leave no registers
    // to keep the decoded coordinates of every item)
    const int total = NPV * C4;
    // The loads of the next (tile, chunk) item fly while the MFMAs of the current one run; staging costs only the LDS stores.
    f32x4 pre[NPF], dpre[NG];
    uint32_t zmask = 0, dmask = 0;
    auto tile_origin = [&](int tile, uint32_t& b, int& od0, int& oh0, int& ow0) {
        uint32_t t = (uint32_t)tile;
        b = fdiv(t, p.dT0);
        t -= b * p.dT0.d;
        const uint32_t itd = fdiv(t, p.dT1);
        t -= itd * p.dT1.d;
        const uint32_t ith = fdiv(t, p.dT2);
        const uint32_t itw = t - ith * p.dT2.d;
        od0 = (int)(itd << p.ltd); oh0 = (int)(ith << p.lth); ow0 = (int)itw * TW;
    };
    auto issue_patch = [&](int tile, int ch) {
        uint32_t b;
        int od0, oh0, ow0;
        tile_origin(tile, b, od0, oh0, ow0);
        const float* __restrict__ xb = p.X + (int64_t)b * p.Sd * p.Sh * p.Sw * p.ldx + c_lo + ch * CC;
        zmask = 0;
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int ic = min(tid + u * 256, total - 1);
            const int pv = ic / C4, c4 = ic - pv * C4;
            const uint32_t pd = fdiv((uint32_t)pv, p.dHW);
            const uint32_t rem = pv - pd * p.dHW.d;
            const uint32_t ph = fdiv(rem, p.dW);
            const uint32_t pw = rem - ph * p.dW.d;
            const int sd = od0 - p.pad + (int)pd, sh = oh0 - p.pad + (int)ph, sw = ow0 - p.pad + (int)pw;
            const int cd = min(max(sd, 0), p.Sd - 1), chh = min(max(sh, 0), p.Sh - 1), cw = min(max(sw, 0), p.Sw - 1);
            zmask |= (uint32_t)(!p.clamp && (cd != sd || chh != sh || cw != sw)) << u;
            pre[u] = *reinterpret_cast<const f32x4*>(xb + ((int64_t)(cd * p.Sh + chh) * p.Sw + cw) * p.ldx + c4 * 4);
        }
    };
    auto stage_direct = [&](int tile, int ch) {      // no prefetch: batches of 4 loads -> LDS stores
        uint32_t b;
        int od0, oh0, ow0;
        tile_origin(tile, b, od0, oh0, ow0);
        const float* __restrict__ xb = p.X + (int64_t)b * p.Sd * p.Sh * p.Sw * p.ldx + c_lo + ch * CC;
        for (int i0 = tid; i0 < total; i0 += 256 * 4) {
            f32x4 v[4];
            bool zf[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ic = min(i0 + u * 256, total - 1);
                const int pv = ic / C4, c4 = ic - pv * C4;
                const uint32_t pd = fdiv((uint32_t)pv, p.dHW);
                const uint32_t rem = pv - pd * p.dHW.d;
                const uint32_t ph = fdiv(rem, p.dW);
                const uint32_t pw = rem - ph * p.dW.d;
                const int sd = od0 - p.pad + (int)pd, sh = oh0 - p.pad + (int)ph, sw = ow0 - p.pad + (int)pw;
                const int cd = min(max(sd, 0), p.Sd - 1), chh = min(max(sh, 0), p.Sh - 1), cw = min(max(sw, 0), p.Sw - 1);
                zf[u] = !p.clamp && (cd != sd || chh != sh || cw != sw);
                v[u] = *reinterpret_cast<const f32x4*>(xb + ((int64_t)(cd * p.Sh + chh) * p.Sw + cw) * p.ldx + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * 256 < total) *reinterpret_cast<f32x4*>(patch + (i0 + u * 256) * 4) = zf[u] ? (f32x4){0.f, 0.f, 0.f, 0.f} : v[u];
        }
    };
    auto issue_dy = [&](int tile) {      // dY tile: voxel v = (vd*TH + vh)*16 + vw, item i = v*NG + g
        uint32_t b;
        int od0, oh0, ow0;
        tile_origin(tile, b, od0, oh0, ow0);
        const float* __restrict__ gb = p.DY + (int64_t)b * p.Od * p.Oh * p.Ow * p.lddy;
        dmask = 0;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int i = tid + 256 * j;
            const int v = i / NG, g = i - v * NG;
            const int vw = v & 15, vh = (v >> 4) & (TH - 1), vd = v >> (4 + p.lth);
            const int od = od0 + vd, oh = oh0 + vh, ow = ow0 + vw;
            const bool ok = od < p.Od && oh < p.Oh && ow < p.Ow && g * 4 < p.Co;
            dpre[j] = *reinterpret_cast<const f32x4*>(gb + (ok ? ((int64_t)(od * p.Oh + oh) * p.Ow + ow) * p.lddy + g * 4 : 0));
            dmask |= (uint32_t)ok << j;
        }
    };

    // NG >= 3: the accumulators (28 NG registers) leave no room to hold the next item across the MFMA section without spilling,
    // so those variants load right before the LDS stores (two workgroups per CU still overlap each other's staging).
    constexpr bool PF = NG <= 2 && NPF <= 11;     // (the shallow-grid NPF = 16 variants would spill as well)
    int tile = blockIdx.x;
    if (PF && tile < p.ntiles) { issue_patch(tile, 0); issue_dy(tile); }
    for (; tile < p.ntiles; tile += gridDim.x) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            __syncthreads();                                    // the previous item's readers are done
            if constexpr (!PF) {
                stage_direct(tile, ch);
                if (ch == 0) issue_dy(tile);
            }
#pragma unroll
            for (int u = 0; u < (PF ? NPF : 0); ++u) {      // CP == CC: the patch is item-linear
                const int i = tid + u * 256;
                if (i < total) *reinterpret_cast<f32x4*>(patch + i * 4) = (zmask >> u) & 1 ? (f32x4){0.f, 0.f, 0.f, 0.f} : pre[u];
            }
            if (ch == 0) {
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    const int i = tid + 256 * j;
                    const int v = i / NG, g = i - v * NG;
                    *reinterpret_cast<f32x4*>(dys + v * DP + g * 4) = (dmask >> j) & 1 ? dpre[j] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
            __syncthreads();
            if constexpr (PF) {
                if (ch + 1 < NCH) {
                    issue_patch(tile, ch + 1);
                } else if (tile + (int)gridDim.x < p.ntiles) {
                    issue_patch(tile + (int)gridDim.x, 0);
                    issue_dy(tile + (int)gridDim.x);
                }
            }
            // ---- this wave's 64 voxels = 4 w-rows of 16.  The X values of column group cg+1 (and of the next w-row's first
            //      group) are requested before the 16 NG MFMAs of group cg issue: a ds_read -> wait -> 4 MFMAs chain exposed the
            //      LDS latency every 32 cycles.
            auto rowoff_of = [&](int r) { return (((r >> p.lth) * PH + (r & (TH - 1))) * PW) * CP; };
            float xb[2][8];
            float areg[2][NG];
            wg_load8<CP>(xb[0], patch + rowoff_of(wave * 4) + colbase[0]);
#pragma unroll
            for (int g = 0; g < NG; ++g) areg[0][g] = dys[(wave * 4 * 16 + (lane >> 2)) * DP + g * 4 + (lane & 3)];
#pragma unroll
            for (int vq = 0; vq < 4; ++vq) {
                const int r = wave * 4 + vq;                    // w-row inside the tile
                int ro = rowoff_of(r), ro_next = rowoff_of(r + 1);           // r + 1 of the last row stays inside the halo'd patch (unused)
                // opaque to the optimiser: otherwise all 28 (row, column group) LDS addresses are hoisted out of the persistent tile
                // loop and held in registers for the whole kernel (spills); one v_add per 8 reads is free
                asm volatile("" : "+v"(ro), "+v"(ro_next));
                if (vq + 1 < 4) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) areg[(vq + 1) & 1][g] = dys[((r + 1) * 16 + (lane >> 2)) * DP + g * 4 + (lane & 3)];
                }
#pragma unroll
                for (int cg = 0; cg < NCG; ++cg) {
                    // first half: voxels 0..7 are in xb[0]; request voxels 8..15, then the next group's (or next row's) first half
                    wg_load8<CP>(xb[1], patch + ro + colbase[cg] + 8 * CP);
                    __builtin_amdgcn_sched_barrier(0);
                    wg_half<NG, 0>(acc[ch][cg], areg[vq & 1], xb[0], std::make_integer_sequence<int, 8>{});
                    if (cg + 1 < NCG) wg_load8<CP>(xb[0], patch + ro + colbase[cg + 1]);
                    else if (vq + 1 < 4) wg_load8<CP>(xb[0], patch + ro_next + colbase[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    wg_half<NG, 1>(acc[ch][cg], areg[vq & 1], xb[1], std::make_integer_sequence<int, 8>{});
                }
            }
        }
    }
    // ---- every wave writes its own slab
    float* __restrict__ out = p.ws + ((int64_t)blockIdx.x * 4 + wave) * p.Co * 27 * p.Ci;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            if (!colok[cg]) continue;
            const int col = cg * 64 + lane;
            const int tap = col / CC, ci = c_lo + ch * CC + col % CC;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = g * 4 + i;
                    if (co < p.Co) out[((int64_t)co * 27 + tap) * p.Ci + ci] = acc[ch][cg][g][i];
                }
        }
}

static int wg_cc(int Ci) { return (Ci & 15) ? 8 : 16; }
static void wg_cfg(int Ci, int Co, int& NG, int& NCH, int& nz, int& wgs) {
    NG = (Co + 3) / 4;
    NCH = 1;        // one channel chunk per workgroup: with the next item prefetched in registers there is no room for more accumulators
    nz = Ci / (wg_cc(Ci) * NCH);
    wgs = 512 / nz;
    if (wgs < 1) wgs = 1;
}
extern "C" int corrif_conv3_patch_wgrad_slots(int32_t Ci, int32_t Co) {
    if (Co <= 0 || Co > 16 || (Co & 3) || Ci <= 0 || (Ci & 7)) return 0;
    int NG, NCH, nz, wgs;
    wg_cfg(Ci, Co, NG, NCH, nz, wgs);
    return wgs * 4;
}
extern "C" size_t corrif_conv3_patch_wgrad_workspace(int32_t Ci, int32_t Co) {
    return (size_t)corrif_conv3_patch_wgrad_slots(Ci, Co) * Co * 27 * Ci * sizeof(float);
}

template <int NG, int NCH, int NPF, int CC>
static int launch_wg_npf(const PatchWgArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_patch_wgrad_kernel<NG, NCH, NPF, CC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return CORRIF_ELAUNCH;
        }
        done = true;
    }
    hipLaunchKernelGGL((conv3_patch_wgrad_kernel<NG, NCH, NPF, CC>), grid, dim3(256), lds, s, a);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
template <int NG, int NCH, int CC>
static int launch_wg(const PatchWgArgs& a, dim3 grid, size_t lds, int npv, hipStream_t s) {
    constexpr int C4 = CC / 4, NPF_STD = (648 * C4 + 255) / 256, NPF_MAX = (972 * C4 + 255) / 256;     // halo float4 per thread
    if (npv * C4 <= NPF_STD * 256) return launch_wg_npf<NG, NCH, NPF_STD, CC>(a, grid, lds, s);
    if (npv * C4 <= NPF_MAX * 256) return launch_wg_npf<NG, NCH, NPF_MAX, CC>(a, grid, lds, s);
    return CORRIF_EUNSUPPORTED;
}

extern "C" int corrif_conv3_patch_wgrad(const CorrifConv3PatchWgrad* q, void* stream) {
    if (!q || !q->X || !q->DY || !q->dW || !q->ws) return CORRIF_EINVAL;
    if (q->B <= 0 || q->Sd <= 0 || q->Sh <= 0 || q->Sw <= 0 || q->Od <= 0 || q->Oh <= 0 || q->Ow <= 0 || q->pad != 1) return CORRIF_EINVAL;
    if (!corrif_conv3_patch_wgrad_slots(q->Ci, q->Co)) return CORRIF_EUNSUPPORTED;
    if ((q->ldx & 3) || (q->lddy & 3) || ((uintptr_t)q->X & 15) || ((uintptr_t)q->DY & 15)) return CORRIF_EUNSUPPORTED;
    int NG, NCH, nz, wgs;
    wg_cfg(q->Ci, q->Co, NG, NCH, nz, wgs);
    PatchWgArgs a;
    a.X = q->X; a.ldx = q->ldx; a.DY = q->DY; a.lddy = q->lddy; a.ws = q->ws;
    a.B = q->B; a.Sd = q->Sd; a.Sh = q->Sh; a.Sw = q->Sw; a.Od = q->Od; a.Oh = q->Oh; a.Ow = q->Ow;
    a.Ci = q->Ci; a.Co = q->Co; a.pad = q->pad; a.clamp = q->clamp;
    int TD = 4, TH = 4;
    if (q->Od < 4) { TD = q->Od >= 2 ? 2 : 1; TH = 16 / TD; }
    a.ltd = ilog2(TD); a.lth = ilog2(TH);
    a.ntd = (q->Od + TD - 1) / TD; a.nth = (q->Oh + TH - 1) / TH; a.ntw = (q->Ow + 15) / 16;
    const int64_t ntiles = (int64_t)q->B * a.ntd * a.nth * a.ntw;
    if (ntiles >= ((int64_t)1 << 31)) return CORRIF_EUNSUPPORTED;
    a.ntiles = (int)ntiles;
    const int PH = TH + 2;
    a.dHW = make_fastdiv((uint32_t)(PH * 18));
    a.dW = make_fastdiv(18);
    a.dT0 = make_fastdiv((uint32_t)(a.ntd * a.nth * a.ntw));
    a.dT1 = make_fastdiv((uint32_t)(a.nth * a.ntw));
    a.dT2 = make_fastdiv((uint32_t)a.ntw);
    const int CC = wg_cc(q->Ci);
    const size_t lds = (size_t)(256 * (4 * NG + 4) + (TD + 2) * PH * 18 * CC) * sizeof(float);
    if (lds > 160 * 1024) return CORRIF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(wgs, 1, nz);          // slabs of workgroups that own no tile stay zero: the kernel still writes them
    int rc = CORRIF_EUNSUPPORTED;
    const int npv = (TD + 2) * PH * 18;
    if (CC == 16) {
        if (NG == 1 && NCH == 1) rc = launch_wg<1, 1, 16>(a, grid, lds, npv, s);
        else if (NG == 2 && NCH == 1) rc = launch_wg<2, 1, 16>(a, grid, lds, npv, s);
        else if (NG == 3) rc = launch_wg<3, 1, 16>(a, grid, lds, npv, s);
        else if (NG == 4) rc = launch_wg<4, 1, 16>(a, grid, lds, npv, s);
    } else {
        if (NG == 1 && NCH == 1) rc = launch_wg<1, 1, 8>(a, grid, lds, npv, s);
        else if (NG == 2 && NCH == 1) rc = launch_wg<2, 1, 8>(a, grid, lds, npv, s);
        else if (NG == 3) rc = launch_wg<3, 1, 8>(a, grid, lds, npv, s);
        else if (NG == 4) rc = launch_wg<4, 1, 8>(a, grid, lds, npv, s);
    }
    if (rc != CORRIF_OK) return rc;
    return corrif_slab_reduce(q->ws, q->dW, (int64_t)q->Co * 27 * q->Ci, wgs * 4, stream);
}
