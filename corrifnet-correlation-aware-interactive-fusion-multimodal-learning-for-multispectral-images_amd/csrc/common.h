// Shared device/host helpers for the corrif gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/corrif.h"

#define CORRIF_CHECK_LAUNCH()                                   \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return CORRIF_ELAUNCH;           \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Division of a 31-bit unsigned numerator by a runtime constant: q = (n * M) >> sh, 64-bit product.
// M = floor(2^(31+s)/d) + 1, s = ceil(log2 d)   (exact for n < 2^31).
struct FastDiv {
    uint32_t M;
    uint32_t sh;
    uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    if (d == 0) d = 1;
    uint32_t s = 0;
    while ((1u << s) < d) ++s;
    f.M = (uint32_t)((((uint64_t)1) << (31 + s)) / d + 1);
    f.sh = 31 + s;
    f.d = d;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) { return (uint32_t)(((uint64_t)n * f.M) >> f.sh); }

// Device-side gather geometry (CorrifGeom + derived constants).
struct DevGeom {
    int is_gemm;
    int Rd, Rh, Rw, Sd, Sh, Sw, kd, kh, kw;
    int mul_d, mul_h, mul_w, off_d, off_h, off_w, div_d, div_h, div_w, dir, clamp, ntaps;
    FastDiv dRS, dRhw, dRw;   // rows -> (n, d, h, w)
    FastDiv dKhw, dKw;        // tap  -> (td, th, tw)
    int64_t sample_pitch;     // Sd*Sh*Sw*ld  (floats)
};
static inline DevGeom make_devgeom(const CorrifGeom& g, int64_t ld) {
    DevGeom d;
    d.is_gemm = g.is_gemm;
    d.Rd = g.Rd; d.Rh = g.Rh; d.Rw = g.Rw; d.Sd = g.Sd; d.Sh = g.Sh; d.Sw = g.Sw;
    d.kd = g.kd; d.kh = g.kh; d.kw = g.kw;
    d.mul_d = g.mul_d; d.mul_h = g.mul_h; d.mul_w = g.mul_w;
    d.off_d = g.off_d; d.off_h = g.off_h; d.off_w = g.off_w;
    d.div_d = g.div_d; d.div_h = g.div_h; d.div_w = g.div_w;
    d.dir = g.dir; d.clamp = g.clamp; d.ntaps = g.ntaps;
    d.dRS = make_fastdiv((uint32_t)(g.Rd * g.Rh * g.Rw));
    d.dRhw = make_fastdiv((uint32_t)(g.Rh * g.Rw));
    d.dRw = make_fastdiv((uint32_t)g.Rw);
    d.dKhw = make_fastdiv((uint32_t)(g.kh * g.kw));
    d.dKw = make_fastdiv((uint32_t)g.kw);
    d.sample_pitch = g.src_batch_pitch ? g.src_batch_pitch : (int64_t)g.Sd * g.Sh * g.Sw * ld;
    return d;
}
static inline bool geom_ok(const CorrifGeom& g) {
    if (g.is_gemm) return true;
    if (g.Rd <= 0 || g.Rh <= 0 || g.Rw <= 0 || g.Sd <= 0 || g.Sh <= 0 || g.Sw <= 0) return false;
    if (g.Rd >= 1024 || g.Rh >= 1024 || g.Rw >= 1024) return false;      // 10/10/10-bit packing of (d,h,w)
    if (g.kd <= 0 || g.kh <= 0 || g.kw <= 0) return false;
    if (g.div_d <= 0 || g.div_h <= 0 || g.div_w <= 0) return false;
    if (g.dir != 1 && g.dir != -1) return false;
    if (g.clamp && (g.div_d != 1 || g.div_h != 1 || g.div_w != 1)) return false;
    return true;
}

// row -> sample index and packed (d,h,w)
__device__ __forceinline__ void decode_row(uint32_t row, const DevGeom& g, uint32_t& n, uint32_t& packed) {
    n = fdiv(row, g.dRS);
    uint32_t rem = row - n * g.dRS.d;
    uint32_t d = fdiv(rem, g.dRhw);
    rem -= d * g.dRhw.d;
    uint32_t h = fdiv(rem, g.dRw);
    uint32_t w = rem - h * g.dRw.d;
    packed = (d << 20) | (h << 10) | w;
}
// one axis of the gather; returns false if the tap falls outside (zero padding)
__device__ __forceinline__ bool axis_src(int r, int t, int mul, int dir, int off, int dv, int S, int clamp, int& s) {
    int num = r * mul + dir * t + off;
    if (dv != 1) {
        if (num < 0) return false;
        int q = num / dv;
        if (q * dv != num) return false;
        num = q;
    }
    if (clamp) {
        s = min(max(num, 0), S - 1);
        return true;
    }
    s = num;
    return num >= 0 && num < S;
}
// voxel offset (in voxels, per sample) of the source for packed row coords and tap; false = zero
__device__ __forceinline__ bool gather_voxel(uint32_t packed, int td, int th, int tw, const DevGeom& g, int& vox) {
    int rd = packed >> 20, rh = (packed >> 10) & 1023, rw = packed & 1023;
    int sd, sh, sw;
    bool ok = axis_src(rd, td, g.mul_d, g.dir, g.off_d, g.div_d, g.Sd, g.clamp, sd);
    ok = axis_src(rh, th, g.mul_h, g.dir, g.off_h, g.div_h, g.Sh, g.clamp, sh) && ok;
    ok = axis_src(rw, tw, g.mul_w, g.dir, g.off_w, g.div_w, g.Sw, g.clamp, sw) && ok;
    vox = (sd * g.Sh + sh) * g.Sw + sw;
    return ok;
}

// XCD-aware bijective block remap: blocks that share blockIdx % 8 (one XCD's L2 under round-robin
// dispatch) get a contiguous range of tile ids.  Speed only; any placement is correct.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
    uint32_t q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float kInvSqrt2Pi = 0.39894228040143267794f;
    float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    return cdf + x * kInvSqrt2Pi * expf(-0.5f * x * x);
}

// ---- depth axis of a nearest-neighbour up-sampling Ds -> D slices (F.interpolate(nearest), same float arithmetic as nearest_fwd_kernel)
// and the depth classes of the decoder's compact skip branch (misc.hip: depth_bcast_add / depth_class_reduce; conv3_patch.hip: fused add)
struct DAxis { int in, out; float scale; };      // the depth axis of nearest_fwd_kernel (same float arithmetic)
static DAxis make_daxis(int in, int out) { DAxis a; a.in = in; a.out = out; a.scale = (float)in / (float)out; return a; }
__device__ __forceinline__ int depth_src(const DAxis& a, int o) {
    int s = (int)floorf((float)o * a.scale);
    return s < a.in - 1 ? s : a.in - 1;
}
__device__ __forceinline__ int depth_class(int d, const DAxis& a) {
    const int k = depth_src(a, d);
    const bool first = d == 0 || depth_src(a, d - 1) != k, last = d == a.out - 1 || depth_src(a, d + 1) != k;
    return 3 * k + (first ? 0 : (last ? 2 : 1));
}
// first / last slice of block k (the slices whose source is k)
__device__ __forceinline__ void depth_block(const DAxis& a, int k, int& lo, int& hi) {
    const float inv = 1.0f / a.scale;
    lo = (int)floorf((float)k * inv) - 1;
    hi = (int)ceilf((float)(k + 1) * inv) + 1;
    if (lo < 0) lo = 0;
    if (hi > a.out - 1) hi = a.out - 1;
    while (lo <= hi && depth_src(a, lo) != k) ++lo;
    while (hi >= lo && depth_src(a, hi) != k) --hi;
}
