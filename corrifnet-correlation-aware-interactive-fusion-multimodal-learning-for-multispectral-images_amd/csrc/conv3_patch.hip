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
        // ---- stage the halo'd input patch of this channel chunk
        {
            constexpr int C4 = CC / 4;
            const int total = NPV * C4;
            for (int i = tid; i < total; i += 256) {
                const int pv = i / C4, c4 = i - pv * C4;
                const uint32_t pd = fdiv((uint32_t)pv, p.dHW);
                const uint32_t rem = pv - pd * p.dHW.d;
                const uint32_t ph = fdiv(rem, p.dW);
                const uint32_t pw = rem - ph * p.dW.d;
                int sd = od0 + (int)pd - p.pad, sh = oh0 + (int)ph - p.pad, sw = ow0 + (int)pw - p.pad;
                bool ok = true;
                if (p.clamp) {
                    sd = min(max(sd, 0), p.Sd - 1);
                    sh = min(max(sh, 0), p.Sh - 1);
                    sw = min(max(sw, 0), p.Sw - 1);
                } else {
                    ok = sd >= 0 && sd < p.Sd && sh >= 0 && sh < p.Sh && sw >= 0 && sw < p.Sw;
                }
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(Xb + ((int64_t)(sd * p.Sh + sh) * p.Sw + sw) * p.ldx + ch * CC + c4 * 4);
                *reinterpret_cast<f32x4*>(patch + pv * Cfg::CP + c4 * 4) = v;
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
