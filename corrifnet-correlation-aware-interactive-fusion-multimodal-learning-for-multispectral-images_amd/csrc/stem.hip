// The encoder stem: Conv3d(1 -> 64, kernel (3,7,7), stride (1,2,2), padding (1,3,3), no bias) on one modality plane of the NCDHW
// input (mmvit4.py:120,172; inflate_conv of the ResNet-50 conv1).  Forward and weight gradient (the input needs no gradient).
//
// Why its own kernels: with Cin = 1 the K index of the implicit GEMM is the TAP, so the generic loader gathers scalars with one
// (td, th, tw) decode and one bounds test per element (27-40 TFLOP/s, matrix pipe busy 0.26-0.33).  Here a workgroup stages the
// halo'd input patch of a 16 x 16 (forward) / 8 x 16 (weight gradient) output tile ONCE in LDS - zero padding resolved while staging,
// coalesced along w - and every MFMA operand is an LDS read at (tap offset + voxel offset), the tap offsets coming from a 148-entry
// table.  v_mfma_f32_32x32x2_f32; K = 147 taps padded to 148 with a zero weight row.
#include "common.h"

namespace corrif_stem {

constexpr int KT = 147, KP = 148, CO = 64;     // taps, padded taps, output channels
constexpr int PWD = 38;                        // patch row pitch (floats): 2 * 16 + 5 = 37 columns

struct StemArgs {
    const float* X; int64_t batch_pitch;       // [B][D][H][W] of one modality, samples batch_pitch floats apart
    const float* Wk;                           // forward: [64][148] (co-major, taps contiguous, column 147 zero)
    float* Y; int64_t ldy;                     // forward: output rows [B*D*Ho*Wo][64]; weight gradient: dY (read)
    float* ws;                                 // weight gradient: per-wave slabs [grid * 4][64][148]
    int B, D, H, W, Ho, Wo, nth, ntw, ntiles, TH;      // TH = output rows per tile (16 forward, 8 weight gradient)
    FastDiv dT0, dT1, dT2;                     // tile -> (b, d, th, tw)
};

__device__ __forceinline__ void tile_origin(const StemArgs& p, int tile, int& b, int& d, int& oh0, int& ow0) {
    uint32_t t = (uint32_t)tile;
    b = (int)fdiv(t, p.dT0);
    t -= (uint32_t)b * p.dT0.d;
    d = (int)fdiv(t, p.dT1);
    t -= (uint32_t)d * p.dT1.d;
    const int ith = (int)fdiv(t, p.dT2), itw = (int)t - ith * (int)p.dT2.d;
    oh0 = ith * p.TH;
    ow0 = itw * 16;
}
// patch[td][ph][pw] = x[b][d - 1 + td][2 oh0 - 3 + ph][2 ow0 - 3 + pw] (zero outside), ph < 2 TH + 5, pw < 37
template <int TH>
__device__ __forceinline__ void stage_patch(const StemArgs& p, float* patch, int b, int d, int oh0, int ow0, int tid) {
    constexpr int PH = 2 * TH + 5;
    const float* __restrict__ xb = p.X + (int64_t)b * p.batch_pitch;
    const int total = 3 * PH * PWD;
    for (int i0 = tid; i0 < total; i0 += 256 * 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * 256, total - 1);
            const int row = i / PWD, pw = i - row * PWD;
            const int td = row / PH, ph = row - td * PH;
            const int sd = d - 1 + td, sh = 2 * oh0 - 3 + ph, sw = 2 * ow0 - 3 + pw;
            const bool ok = sd >= 0 && sd < p.D && sh >= 0 && sh < p.H && sw >= 0 && sw < p.W && pw < 37;
            const float x = xb[ok ? ((int64_t)sd * p.H + sh) * p.W + sw : 0];      // unconditional load, select afterwards
            v[u] = ok ? x : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * 256 < total) patch[i0 + u * 256] = v[u];
    }
}
// LDS offset of tap k inside the patch (k >= 147: tap 0, its weight row is zero / its gradient column is dropped)
__device__ __forceinline__ int tap_offset(int k, int PH) {
    if (k >= KT) k = 0;
    const int td = k / 49, r = k - td * 49, th = r / 7, tw = r - th * 7;
    return (td * PH + th) * PWD + tw;
}

// ------------------------------------------------------------------------------------------------ forward
// workgroup = 16 x 16 output voxels of one (b, d) plane x 64 channels; wave w owns output rows 4 w .. 4 w + 3 (two 32-voxel MFMA tiles)
__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(StemArgs p) {
    constexpr int PH = 2 * 16 + 5, PB = CO + 4;
    __shared__ __attribute__((aligned(16))) float patch[3 * PH * PWD];
    __shared__ __attribute__((aligned(16))) float wl[KP * PB];        // [k][co]
    __shared__ int koff[KP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < CO * (KP / 4); i += 256) {                   // weights -> LDS, transposed to k-major
        const int co = i / (KP / 4), k4 = i - co * (KP / 4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(p.Wk + (int64_t)co * KP + k4 * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) wl[(k4 * 4 + e) * PB + co] = v[e];
    }
    if (tid < KP) koff[tid] = tap_offset(tid, PH);
    // voxel offsets of this lane's rows in its two MFMA tiles: tile t covers output rows 2 t, 2 t + 1 (16 voxels each)
    int abase[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) abase[t] = (2 * (2 * (2 * wave + t) + (li >> 4))) * PWD + 2 * (li & 15);

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        int b, d, oh0, ow0;
        tile_origin(p, tile, b, d, oh0, ow0);
        __syncthreads();                                               // the previous tile's readers are done
        stage_patch<16>(p, patch, b, d, oh0, ow0, tid);
        __syncthreads();
        f32x16 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
#pragma unroll 2
        for (int s = 0; s < KP / 2; ++s) {
            const int k = 2 * s + lh, ko = koff[k];
            const float a0 = patch[ko + abase[0]], a1 = patch[ko + abase[1]];
            const float b0 = wl[k * PB + li], b1 = wl[k * PB + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        // C/D map: column = lane & 31 (channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (voxel inside the 32-voxel tile)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int v = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int oh = oh0 + 2 * (2 * wave + t) + (v >> 4), ow = ow0 + (v & 15);
                if (oh < p.Ho && ow < p.Wo) {
                    float* __restrict__ dst = p.Y + ((((int64_t)b * p.D + d) * p.Ho + oh) * p.Wo + ow) * p.ldy;
                    dst[li] = acc[t][0][r];
                    dst[32 + li] = acc[t][1][r];
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[co][k] = sum over output voxels of dY[voxel][co] * xpad[voxel, tap k].  Workgroup tile = 8 x 16 output voxels; wave w contracts
// over its 32 voxels (output rows 2 w, 2 w + 1) into ALL 2 x 5 output tiles (64 channels x 160 tap columns) kept in accumulators for
// the whole persistent loop; every wave writes its own slab, summed in a fixed order by corrif_slab_reduce (deterministic).
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(StemArgs p) {
    constexpr int PH = 2 * 8 + 5, PD = CO + 4;
    __shared__ __attribute__((aligned(16))) float patch[3 * PH * PWD];
    __shared__ __attribute__((aligned(16))) float dys[128 * PD];       // [voxel][co]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    int kcol[5];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) kcol[kt] = tap_offset(32 * kt + li, PH);
    f32x16 acc[2][5];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int kt = 0; kt < 5; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][kt][r] = 0.f;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        int b, d, oh0, ow0;
        tile_origin(p, tile, b, d, oh0, ow0);
        __syncthreads();
        stage_patch<8>(p, patch, b, d, oh0, ow0, tid);
        // dY tile: 128 voxels x 16 float4
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + 256 * u, v = i >> 4, c4 = i & 15;
            const int oh = oh0 + (v >> 4), ow = ow0 + (v & 15);
            const bool ok = oh < p.Ho && ow < p.Wo;
            f32x4 g = *reinterpret_cast<const f32x4*>(p.Y + (ok ? ((((int64_t)b * p.D + d) * p.Ho + oh) * p.Wo + ow) * p.ldy + c4 * 4 : 0));
            if (!ok) g = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&dys[v * PD + c4 * 4]) = g;
        }
        __syncthreads();
#pragma unroll 2
        for (int s = 0; s < 16; ++s) {
            const int v = 32 * wave + 2 * s + lh;                      // this half-wave's voxel of the pair
            const int vo = (2 * (v >> 4)) * PWD + 2 * (v & 15);
            const float a0 = dys[v * PD + li], a1 = dys[v * PD + 32 + li];
            float x[5];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) x[kt] = patch[kcol[kt] + vo];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                acc[0][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, x[kt], acc[0][kt], 0, 0, 0);
                acc[1][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, x[kt], acc[1][kt], 0, 0, 0);
            }
        }
    }
    float* __restrict__ out = p.ws + ((int64_t)blockIdx.x * 4 + wave) * CO * KP;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int kt = 0; kt < 5; ++kt) {
            const int k = 32 * kt + li;
            if (k >= KP) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[(int64_t)co * KP + k] = k < KT ? acc[c][kt][r] : 0.f;
            }
        }
}

static bool stem_geom_ok(int32_t B, int32_t D, int32_t H, int32_t W) {
    return B > 0 && D > 0 && H > 0 && W > 0 && (int64_t)B * D * ((H + 1) / 2) * ((W + 1) / 2) < ((int64_t)1 << 31);
}
static void fill_args(StemArgs& a, int32_t B, int32_t D, int32_t H, int32_t W, int TH) {
    a.B = B; a.D = D; a.H = H; a.W = W;
    a.Ho = (H + 2 * 3 - 7) / 2 + 1; a.Wo = (W + 2 * 3 - 7) / 2 + 1;
    a.TH = TH;
    a.nth = (a.Ho + TH - 1) / TH; a.ntw = (a.Wo + 15) / 16;
    a.ntiles = B * D * a.nth * a.ntw;
    a.dT0 = make_fastdiv((uint32_t)(D * a.nth * a.ntw));
    a.dT1 = make_fastdiv((uint32_t)(a.nth * a.ntw));
    a.dT2 = make_fastdiv((uint32_t)a.ntw);
}

}  // namespace corrif_stem
using namespace corrif_stem;

extern "C" int corrif_stem_supported(int32_t Co, int32_t kd, int32_t kh, int32_t kw, int32_t sd, int32_t sh, int32_t sw, int32_t pd, int32_t ph,
                                     int32_t pw) {
    return Co == CO && kd == 3 && kh == 7 && kw == 7 && sd == 1 && sh == 2 && sw == 2 && pd == 1 && ph == 3 && pw == 3;
}

extern "C" int corrif_stem_fwd(const float* x, int64_t batch_pitch, const float* wk, float* y, int64_t ldy, int32_t B, int32_t D, int32_t H, int32_t W,
                               void* stream) {
    if (!x || !wk || !y || !stem_geom_ok(B, D, H, W) || ldy < CO || batch_pitch < (int64_t)D * H * W) return CORRIF_EINVAL;
    if (((uintptr_t)wk & 15)) return CORRIF_EUNSUPPORTED;
    StemArgs a{};
    a.X = x; a.batch_pitch = batch_pitch; a.Wk = wk; a.Y = y; a.ldy = ldy;
    fill_args(a, B, D, H, W, 16);
    const unsigned grid = a.ntiles < 512 ? (unsigned)a.ntiles : 512u;
    hipLaunchKernelGGL(stem_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

extern "C" size_t corrif_stem_wgrad_workspace(void) { return (size_t)512 * 4 * CO * KP * sizeof(float); }

/* dwk: [64][148] in the forward's weight layout (column 147 = 0) */
extern "C" int corrif_stem_wgrad(const float* x, int64_t batch_pitch, const float* dy, int64_t lddy, float* dwk, float* ws, int32_t B, int32_t D,
                                 int32_t H, int32_t W, void* stream) {
    if (!x || !dy || !dwk || !ws || !stem_geom_ok(B, D, H, W) || lddy < CO || batch_pitch < (int64_t)D * H * W) return CORRIF_EINVAL;
    if ((lddy & 3) || ((uintptr_t)dy & 15)) return CORRIF_EUNSUPPORTED;
    StemArgs a{};
    a.X = x; a.batch_pitch = batch_pitch; a.Y = const_cast<float*>(dy); a.ldy = lddy; a.ws = ws;
    fill_args(a, B, D, H, W, 8);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, a);     // every workgroup writes its slabs (zeros if it owns no tile)
    CORRIF_CHECK_LAUNCH();
    return corrif_slab_reduce(ws, dwk, (int64_t)CO * KP, 512 * 4, stream);
}
