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
    int B, Sd, Sh, Sw, Od, Oh, Ow, Ci, Co, pad, clamp;
    int ltd, lth, ltw, ntd, nth, ntw;
    FastDiv dHW, dW;                 // patch voxel index -> (pd, ph, pw)
    FastDiv dT0, dT1, dT2;           // tile index -> (b, td, th, tw)
};

template <int NG, int CC>
struct PatchCfg {
    static constexpr int KC = 27 * CC;
    static constexpr int KCP = (KC + 63) / 64 * 64;
    static constexpr int WP = KCP + 4;         // weight row pitch (floats)
    static constexpr int CP = CC + 4;          // patch voxel pitch (floats)
    static constexpr int NKG = KCP / 64;
};

template <int NG, int CC, int KG, int Bk>
__device__ __forceinline__ void patch_mfma_step(f32x4 (&acc)[NG], const f32x4 (&wreg)[NG], const float* __restrict__ patch, int pvoff,
                                                int PH, int PW) {
    constexpr int kidx = KG * 64 + Bk * 4;
    if constexpr (kidx < 27 * CC) {
        constexpr int tap = kidx / CC, c = kidx % CC;
        constexpr int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
        const int off = (pvoff + (td * PH + th) * PW + tw) * (CC + 4) + c;
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(patch + off);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[g][e], x4[e], acc[g], 4, Bk, 0);
    }
}
template <int NG, int CC, int KG, int... Bs>
__device__ __forceinline__ void patch_kgroup(f32x4 (&acc)[NG], const float* __restrict__ wl, const float* __restrict__ patch, int pvoff,
                                             int PH, int PW, int lane, std::integer_sequence<int, Bs...>) {
    using Cfg = PatchCfg<NG, CC>;
    f32x4 wreg[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) wreg[g] = *reinterpret_cast<const f32x4*>(wl + (g * 4 + (lane & 3)) * Cfg::WP + KG * 64 + (lane >> 2) * 4);
    (patch_mfma_step<NG, CC, KG, Bs>(acc, wreg, patch, pvoff, PH, PW), ...);
}
template <int NG, int CC, int... KGs>
__device__ __forceinline__ void patch_chunk(f32x4 (&acc)[NG], const float* __restrict__ wl, const float* __restrict__ patch, int pvoff,
                                            int PH, int PW, int lane, std::integer_sequence<int, KGs...>) {
    (patch_kgroup<NG, CC, KGs>(acc, wl, patch, pvoff, PH, PW, lane, std::make_integer_sequence<int, 16>{}), ...);
}

template <int NG, int CC>
__global__ __launch_bounds__(256) void conv3_patch_kernel(PatchArgs p) {
    using Cfg = PatchCfg<NG, CC>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TD = 1 << p.ltd, TH = 1 << p.lth, TW = 1 << p.ltw;
    const int PD = TD + 2, PH = TH + 2, PW = TW + 2;
    const int NPV = PD * PH * PW;
    float* wl = smem;                                   // [4*NG][WP]
    float* patch = smem + 4 * NG * Cfg::WP;             // [NPV][CP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tile -> (b, td, th, tw)
    uint32_t t = blockIdx.x;
    const uint32_t b = fdiv(t, p.dT0);
    t -= b * p.dT0.d;
    const uint32_t itd = fdiv(t, p.dT1);
    t -= itd * p.dT1.d;
    const uint32_t ith = fdiv(t, p.dT2);
    const uint32_t itw = t - ith * p.dT2.d;
    const int od0 = itd << p.ltd, oh0 = ith << p.lth, ow0 = itw << p.ltw;

    // this lane's voxel inside the tile
    const int lv = wave * 64 + lane;
    const int vw = lv & (TW - 1), vh = (lv >> p.ltw) & (TH - 1), vd = lv >> (p.ltw + p.lth);
    const int pvoff = (vd * PH + vh) * PW + vw;

    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // zero the weight rows/tails that are never overwritten
    for (int i = tid; i < 4 * NG * Cfg::WP; i += 256) wl[i] = 0.f;

    const float* __restrict__ Xb = p.X + (int64_t)b * p.Sd * p.Sh * p.Sw * p.ldx;
    const int nchunk = p.Ci / CC;
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();                                 // previous chunk's readers are done (and the zero fill is visible)
        // ---- stage weights of this chunk: Co rows x KC floats
        {
            constexpr int K4 = Cfg::KC / 4;
            const float* __restrict__ wsrc = p.Wp + (int64_t)ch * p.Co * Cfg::KC;
            for (int i = tid; i < p.Co * K4; i += 256) {
                const int co = i / K4, kk = i - co * K4;
                *reinterpret_cast<f32x4*>(wl + co * Cfg::WP + kk * 4) = *reinterpret_cast<const f32x4*>(wsrc + (int64_t)co * Cfg::KC + kk * 4);
            }
        }
        // ---- stage the halo'd input patch of this channel chunk; loads are issued in batches of 8 per thread so that
        //      their latencies overlap (a load -> wait -> ds_write loop is bound by 14 serial HBM round trips).  The loads are
        //      unconditional from the clamped voxel (always mapped) and zero padding is applied at the LDS store: a load under a
        //      branch makes the compiler drain vmcnt at the join.
        {
            constexpr int C4 = CC / 4, UB = 8;
            const int total = NPV * C4;
            for (int i0 = tid; i0 < total; i0 += 256 * UB) {
                f32x4 v[UB];
                int dst[UB];
                bool zf[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int i = i0 + u * 256;
                    const int ic = min(i, total - 1);
                    const int pv = ic / C4, c4 = ic - pv * C4;
                    const uint32_t pd = fdiv((uint32_t)pv, p.dHW);
                    const uint32_t rem = pv - pd * p.dHW.d;
                    const uint32_t ph = fdiv(rem, p.dW);
                    const uint32_t pw = rem - ph * p.dW.d;
                    const int sd = od0 + (int)pd - p.pad, sh = oh0 + (int)ph - p.pad, sw = ow0 + (int)pw - p.pad;
                    const int cd = min(max(sd, 0), p.Sd - 1), chh = min(max(sh, 0), p.Sh - 1), cw = min(max(sw, 0), p.Sw - 1);
                    zf[u] = !p.clamp && (cd != sd || chh != sh || cw != sw);
                    dst[u] = i < total ? pv * Cfg::CP + c4 * 4 : -1;
                    v[u] = *reinterpret_cast<const f32x4*>(Xb + ((int64_t)(cd * p.Sh + chh) * p.Sw + cw) * p.ldx + ch * CC + c4 * 4);
                }
#pragma unroll
                for (int u = 0; u < UB; ++u)
                    if (dst[u] >= 0) *reinterpret_cast<f32x4*>(patch + dst[u]) = zf[u] ? (f32x4){0.f, 0.f, 0.f, 0.f} : v[u];
            }
        }
        __syncthreads();
        patch_chunk<NG, CC>(acc, wl, patch, pvoff, PH, PW, lane, std::make_integer_sequence<int, Cfg::NKG>{});
    }

    const int od = od0 + vd, oh = oh0 + vh, ow = ow0 + vw;
    if (od < p.Od && oh < p.Oh && ow < p.Ow) {
        const int64_t row = (((int64_t)b * p.Od + od) * p.Oh + oh) * p.Ow + ow;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g * 4 >= p.Co) continue;
            f32x4 v = acc[g];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + g * 4);
            *reinterpret_cast<f32x4*>(p.Y + row * p.ldy + g * 4) = v;
        }
    }
}

template <int NG, int CC>
static int launch_patch(const PatchArgs& a, unsigned tiles, size_t lds, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_patch_kernel<NG, CC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return CORRIF_ELAUNCH;
        }
        attr_done = true;
    }
    hipLaunchKernelGGL((conv3_patch_kernel<NG, CC>), dim3(tiles), dim3(256), lds, s, a);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

extern "C" int corrif_conv3_patch_cc(int32_t Ci, int32_t Co) {
    if (Co <= 0 || Co > 32 || (Co & 3) || Ci <= 0 || (Ci & 7)) return 0;
    if (Ci > 64) return 0;      // many channel chunks re-stage the halo too often: the implicit GEMM is faster there (d3_c2: 50 vs 38 TF/s)
    return (Co <= 8 && !(Ci & 15)) ? 16 : 8;
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
    a.Ci = q->Ci; a.Co = q->Co; a.pad = q->pad; a.clamp = q->clamp;
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
    const int NG = (q->Co + 3) / 4;
    const int NPV = PD * PH * PW;
    hipStream_t s = (hipStream_t)stream;
#define PATCH_CASE(ng, cc)                                                                                         \
    if (NG == ng && CC == cc) {                                                                                    \
        size_t lds = (size_t)(4 * ng * PatchCfg<ng, cc>::WP + NPV * PatchCfg<ng, cc>::CP) * sizeof(float);         \
        if (lds > 160 * 1024) return CORRIF_EUNSUPPORTED;                                                          \
        return launch_patch<ng, cc>(a, (unsigned)tiles, lds, s);                                                   \
    }
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

template <int NG, int VB>
__device__ __forceinline__ void wg_step(f32x4 (&a)[NG], const float (&areg)[NG], const float* __restrict__ xaddr) {
    // xaddr already includes the lane's column offset and the w-row base; VB*16 floats = this voxel inside the w-row
    const float x = xaddr[VB * 16];
#pragma unroll
    for (int g = 0; g < NG; ++g) a[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(areg[g], x, a[g], 4, VB, 0);
}
template <int NG, int... VBs>
__device__ __forceinline__ void wg_cgroup(f32x4 (&a)[NG], const float (&areg)[NG], const float* __restrict__ xaddr,
                                          std::integer_sequence<int, VBs...>) {
    (wg_step<NG, VBs>(a, areg, xaddr), ...);
}

template <int NG, int NCH>
__global__ __launch_bounds__(256, 2) void conv3_patch_wgrad_kernel(PatchWgArgs p) {
    constexpr int CC = 16, CP = 16, TW = 16, PW = TW + 2, DP = 4 * NG + 4, KC = 27 * CC;   // CP = 16: taps 1 apart in w sit 16 banks apart
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TD = 1 << p.ltd, TH = 1 << p.lth, PH = TH + 2;
    const int NPV = (TD + 2) * PH * PW;
    float* dys = smem;                      // [256][DP]
    float* patch = smem + 256 * DP;         // [NPV][CP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c_lo = blockIdx.z * (CC * NCH);

    // lane -> column (tap, ci) of column group cg: patch offset of the tap + channel
    int colbase[7];
    bool colok[7];
#pragma unroll
    for (int cg = 0; cg < 7; ++cg) {
        const int col = cg * 64 + lane;
        colok[cg] = col < KC;
        const int tap = colok[cg] ? col / CC : 0, ci = colok[cg] ? col % CC : 0;
        const int td = tap / 9, th = (tap / 3) % 3, tw = tap % 3;
        colbase[cg] = ((td * PH + th) * PW + tw) * CP + ci;
    }
    f32x4 acc[NCH][7][NG];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int cg = 0; cg < 7; ++cg)
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[c][cg][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        uint32_t t = tile;
        const uint32_t b = fdiv(t, p.dT0);
        t -= b * p.dT0.d;
        const uint32_t itd = fdiv(t, p.dT1);
        t -= itd * p.dT1.d;
        const uint32_t ith = fdiv(t, p.dT2);
        const uint32_t itw = t - ith * p.dT2.d;
        const int od0 = itd << p.ltd, oh0 = ith << p.lth, ow0 = itw * TW;
        __syncthreads();                                        // previous tile's readers are done
        {   // ---- dY tile: voxel v = (vd*TH + vh)*16 + vw
            const float* __restrict__ gb = p.DY + (int64_t)b * p.Od * p.Oh * p.Ow * p.lddy;
            for (int i = tid; i < 256 * NG; i += 256) {
                const int v = i / NG, g = i - v * NG;
                const int vw = v & 15, vh = (v >> 4) & (TH - 1), vd = v >> (4 + p.lth);
                const int od = od0 + vd, oh = oh0 + vh, ow = ow0 + vw;
                const bool ok = od < p.Od && oh < p.Oh && ow < p.Ow && g * 4 < p.Co;
                const f32x4 val = *reinterpret_cast<const f32x4*>(gb + (ok ? ((int64_t)(od * p.Oh + oh) * p.Ow + ow) * p.lddy + g * 4 : 0));
                *reinterpret_cast<f32x4*>(dys + v * DP + g * 4) = ok ? val : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        const float* __restrict__ xb = p.X + (int64_t)b * p.Sd * p.Sh * p.Sw * p.ldx;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch > 0) __syncthreads();                        // the previous chunk's patch is no longer read
            {   // ---- halo'd X patch of this 16-channel chunk (batched loads)
                constexpr int C4 = CC / 4, UB = 4;
                const int total = NPV * C4;
                for (int i0 = tid; i0 < total; i0 += 256 * UB) {
                    f32x4 v[UB];
                    int dst[UB];
                    bool zf[UB];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        const int i = i0 + u * 256;
                        const int ic = min(i, total - 1);          // unconditional clamped loads, zero fill at the LDS store (see the forward kernel)
                        const int pv = ic / C4, c4 = ic - pv * C4;
                        const uint32_t pd = fdiv((uint32_t)pv, p.dHW);
                        const uint32_t rem = pv - pd * p.dHW.d;
                        const uint32_t ph = fdiv(rem, p.dW);
                        const uint32_t pw = rem - ph * p.dW.d;
                        const int sd = od0 + (int)pd - p.pad, sh = oh0 + (int)ph - p.pad, sw = ow0 + (int)pw - p.pad;
                        const int cd = min(max(sd, 0), p.Sd - 1), chh = min(max(sh, 0), p.Sh - 1), cw = min(max(sw, 0), p.Sw - 1);
                        zf[u] = !p.clamp && (cd != sd || chh != sh || cw != sw);
                        dst[u] = i < total ? pv * CP + c4 * 4 : -1;
                        v[u] = *reinterpret_cast<const f32x4*>(xb + ((int64_t)(cd * p.Sh + chh) * p.Sw + cw) * p.ldx + c_lo + ch * CC + c4 * 4);
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u)
                        if (dst[u] >= 0) *reinterpret_cast<f32x4*>(patch + dst[u]) = zf[u] ? (f32x4){0.f, 0.f, 0.f, 0.f} : v[u];
                }
            }
            __syncthreads();
            // ---- this wave's 64 voxels = 4 w-rows of 16
            for (int vq = 0; vq < 4; ++vq) {
                const int r = wave * 4 + vq;                    // w-row inside the tile
                const int vd = r >> p.lth, vh = r & (TH - 1);
                const int rowoff = ((vd * PH + vh) * PW) * CP;
                float areg[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) areg[g] = dys[(r * 16 + (lane >> 2)) * DP + g * 4 + (lane & 3)];
#pragma unroll
                for (int cg = 0; cg < 7; ++cg)
                    wg_cgroup<NG>(acc[ch][cg], areg, patch + rowoff + colbase[cg], std::make_integer_sequence<int, 16>{});
            }
        }
    }
    // ---- every wave writes its own slab
    float* __restrict__ out = p.ws + ((int64_t)blockIdx.x * 4 + wave) * p.Co * 27 * p.Ci;
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int cg = 0; cg < 7; ++cg) {
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

static void wg_cfg(int Ci, int Co, int& NG, int& NCH, int& nz, int& wgs) {
    NG = (Co + 3) / 4;
    NCH = NG == 1 ? 4 : (NG == 2 ? 2 : 1);
    while (NCH > 1 && (Ci % (16 * NCH))) NCH >>= 1;
    nz = Ci / (16 * NCH);
    wgs = 512 / nz;
    if (wgs < 1) wgs = 1;
}
extern "C" int corrif_conv3_patch_wgrad_slots(int32_t Ci, int32_t Co) {
    if (Co <= 0 || Co > 16 || (Co & 3) || Ci <= 0 || (Ci & 15)) return 0;
    int NG, NCH, nz, wgs;
    wg_cfg(Ci, Co, NG, NCH, nz, wgs);
    return wgs * 4;
}
extern "C" size_t corrif_conv3_patch_wgrad_workspace(int32_t Ci, int32_t Co) {
    return (size_t)corrif_conv3_patch_wgrad_slots(Ci, Co) * Co * 27 * Ci * sizeof(float);
}

template <int NG, int NCH>
static int launch_wg(const PatchWgArgs& a, dim3 grid, size_t lds, hipStream_t s) {
    static bool done = false;
    if (!done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_patch_wgrad_kernel<NG, NCH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return CORRIF_ELAUNCH;
        }
        done = true;
    }
    hipLaunchKernelGGL((conv3_patch_wgrad_kernel<NG, NCH>), grid, dim3(256), lds, s, a);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
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
    const size_t lds = (size_t)(256 * (4 * NG + 4) + (TD + 2) * PH * 18 * 16) * sizeof(float);
    if (lds > 160 * 1024) return CORRIF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(wgs, 1, nz);          // slabs of workgroups that own no tile stay zero: the kernel still writes them
    int rc = CORRIF_EUNSUPPORTED;
    if (NG == 1 && NCH == 4) rc = launch_wg<1, 4>(a, grid, lds, s);
    else if (NG == 1 && NCH == 2) rc = launch_wg<1, 2>(a, grid, lds, s);
    else if (NG == 1 && NCH == 1) rc = launch_wg<1, 1>(a, grid, lds, s);
    else if (NG == 2 && NCH == 2) rc = launch_wg<2, 2>(a, grid, lds, s);
    else if (NG == 2 && NCH == 1) rc = launch_wg<2, 1>(a, grid, lds, s);
    else if (NG == 3) rc = launch_wg<3, 1>(a, grid, lds, s);
    else if (NG == 4) rc = launch_wg<4, 1>(a, grid, lds, s);
    if (rc != CORRIF_OK) return rc;
    return corrif_slab_reduce(q->ws, q->dW, (int64_t)q->Co * 27 * q->Ci, wgs * 4, stream);
}
