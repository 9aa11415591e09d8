// HBM-bound stages of the MMVit4 path: resampling, pooling, padding adjoint, softmax, dropout, the
// inter-modal correlation block, head (+sigmoid), loss, Jaccard, Adam, weight re-layouts and small
// element-wise helpers.  All channels-last, float4 per lane where the layout allows, grid-stride,
// deterministic (no float atomics anywhere).
#include "common.h"

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
static inline unsigned nblocks(int64_t n, int per = 256, int64_t cap = 16384) {
    int64_t b = (n + per - 1) / per;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}
// flat index -> (b, d, h, w, c4) with 32-bit multiply-shift divisions (the index-heavy resampling kernels were VALU-bound on
// 64-bit '/' and '%': nearest 128^3 ran at 1.6 TB/s)
struct Dec5 { FastDiv c, w, h, d; };
static inline Dec5 make_dec5(int D, int H, int W, int C4) {
    Dec5 q;
    q.c = make_fastdiv((uint32_t)C4); q.w = make_fastdiv((uint32_t)W); q.h = make_fastdiv((uint32_t)H); q.d = make_fastdiv((uint32_t)D);
    return q;
}
__device__ __forceinline__ void dec5(uint32_t i, const Dec5& q, int& b, int& d, int& h, int& w, int& c) {
    uint32_t v = fdiv(i, q.c);
    c = (int)(i - v * q.c.d);
    uint32_t t = fdiv(v, q.w);
    w = (int)(v - t * q.w.d);
    v = t;
    t = fdiv(v, q.h);
    h = (int)(v - t * q.h.d);
    v = t;
    t = fdiv(v, q.d);
    d = (int)(v - t * q.d.d);
    b = (int)t;
}
#define GRID_STRIDE(i, n) for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (int64_t)gridDim.x * blockDim.x)

extern "C" int corrif_abi_version(void) { return CORRIF_ABI_VERSION; }
extern "C" const char* corrif_build_arch(void) { return "gfx950"; }

// ------------------------------------------------------------------ element-wise helpers
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t b_n, float* __restrict__ y, int64_t n4) {
    GRID_STRIDE(i, n4) {
        f32x4 av = reinterpret_cast<const f32x4*>(a)[i];
        f32x4 bv = reinterpret_cast<const f32x4*>(b)[b_n ? (i % (b_n >> 2)) : i];
        reinterpret_cast<f32x4*>(y)[i] = av + bv;
    }
}
__global__ void add_scalar_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
    GRID_STRIDE(i, n) y[i] = a[i] + b[i];
}
extern "C" int corrif_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
    if (!a || !b || !y || n <= 0) return CORRIF_EINVAL;
    if ((n & 3) || !al16(a) || !al16(b) || !al16(y)) {      // odd sizes / unaligned views (gradient buckets): scalar path
        hipLaunchKernelGGL(add_scalar_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n);
        CORRIF_CHECK_LAUNCH();
        return CORRIF_OK;
    }
    hipLaunchKernelGGL(add_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, (int64_t)0, y, n / 4);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_add_bcast_rows(const float* a, const float* b, int64_t b_n, float* y, int64_t n, void* stream) {
    if (!a || !b || !y || n <= 0 || b_n <= 0) return CORRIF_EINVAL;
    if ((n & 3) || (b_n & 3) || !al16(a) || !al16(b) || !al16(y)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(add_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, a, b, b_n, y, n / 4);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
// ------------------------------------------------------------------ depth-class broadcast (decoder skip branch on a compact depth grid)
// A tensor that is the nearest-neighbour up-sampling of Ds depth slices to D slices (F.interpolate(nearest): source slice of d is
// min(floor(d * Ds / D), Ds - 1) in float arithmetic, mmvit4.py:271-286) is constant along depth inside each BLOCK of consecutive slices
// that share a source - D / Ds slices when Ds divides D, floor or ceil of it otherwise (the reference-native 3 bands, 12 bands) - so a
// 3x3x3 convolution of it (replicate padded) takes only three distinct values per block: at the block's first slice, at its last slice
// and everywhere in between (blocks are at least two slices long: D >= 2 Ds).  The decoder therefore evaluates the skip branch's share
// of d*_c2 on a grid of 3 * Ds slices (class c = 3 k + {0 first, 1 interior, 2 last} of block k) and these two kernels broadcast it
// into / reduce it out of the full-depth tensor:
//     y[b, d, s, :] += ys[b, cls(d), s, :]        g_ys[b, c, s, :] = sum over d with cls(d) = c of g[b, d, s, :]
__global__ void depth_bcast_add_kernel(float* __restrict__ y, int64_t ldy, const float* __restrict__ ys, int64_t lds, int64_t total, DAxis ad,
                                       int C4, FastDiv dC, FastDiv dS, FastDiv dD, int S) {
    GRID_STRIDE(i, total) {        // i = ((b * D + d) * S + s) * C4 + c
        uint32_t t = (uint32_t)i;
        const uint32_t row = fdiv(t, dC), c = t - row * dC.d;
        const uint32_t bd = fdiv(row, dS), sp = row - bd * dS.d;
        const uint32_t b = fdiv(bd, dD), d = bd - b * dD.d;
        const int64_t srow = ((int64_t)b * (3 * ad.in) + depth_class((int)d, ad)) * S + sp;
        f32x4 v = *reinterpret_cast<const f32x4*>(y + (int64_t)row * ldy + c * 4);
        v += *reinterpret_cast<const f32x4*>(ys + srow * lds + c * 4);
        *reinterpret_cast<f32x4*>(y + (int64_t)row * ldy + c * 4) = v;
    }
}
__global__ void depth_class_reduce_kernel(const float* __restrict__ g, int64_t ldg, float* __restrict__ out, int64_t ldo, int64_t total, DAxis ad,
                                          int C4, FastDiv dC, FastDiv dS, FastDiv dK, int S) {
    GRID_STRIDE(i, total) {        // i = ((b * Ds + k) * S + s) * C4 + c ; fixed summation order over the interior slices of block k
        uint32_t t = (uint32_t)i;
        const uint32_t row = fdiv(t, dC), c = t - row * dC.d;
        const uint32_t bk = fdiv(row, dS), sp = row - bk * dS.d;
        const uint32_t b = fdiv(bk, dK), k = bk - b * dK.d;
        int lo, hi;
        depth_block(ad, (int)k, lo, hi);
        const float* __restrict__ src = g + (((int64_t)b * ad.out) * S + sp) * ldg + c * 4;
        const int64_t step = (int64_t)S * ldg;
        float* __restrict__ dst = out + (((int64_t)b * 3 * dK.d + 3 * k) * S + sp) * ldo + c * 4;
        *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<const f32x4*>(src + lo * step);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int r = lo + 1; r < hi; ++r) acc += *reinterpret_cast<const f32x4*>(src + r * step);
        *reinterpret_cast<f32x4*>(dst + (int64_t)S * ldo) = acc;
        *reinterpret_cast<f32x4*>(dst + 2 * (int64_t)S * ldo) = *reinterpret_cast<const f32x4*>(src + (int64_t)hi * step);
    }
}
static bool depth_args_ok(const void* a, int64_t lda, const void* b, int64_t ldb, int32_t B, int32_t D, int32_t S, int32_t C, int32_t Ds) {
    if (!a || !b || B <= 0 || D <= 0 || S <= 0 || C <= 0 || Ds < 1 || D < 2 * Ds) return false;      // every block has a first and a last slice
    if ((C & 3) || (lda & 3) || (ldb & 3) || lda < C || ldb < C || !al16(a) || !al16(b)) return false;
    return (int64_t)B * D * S * (C / 4) < ((int64_t)1 << 31);
}
extern "C" int corrif_depth_bcast_add(float* y, int64_t ldy, const float* ys, int64_t lds, int32_t B, int32_t D, int32_t S, int32_t C, int32_t Ds,
                                      void* stream) {
    if (!depth_args_ok(y, ldy, ys, lds, B, D, S, C, Ds)) return CORRIF_EINVAL;
    const int64_t total = (int64_t)B * D * S * (C / 4);
    hipLaunchKernelGGL(depth_bcast_add_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, y, ldy, ys, lds, total, make_daxis(Ds, D),
                       C / 4, make_fastdiv((uint32_t)(C / 4)), make_fastdiv((uint32_t)S), make_fastdiv((uint32_t)D), (int)S);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_depth_class_reduce(const float* g, int64_t ldg, float* out, int64_t ldo, int32_t B, int32_t D, int32_t S, int32_t C, int32_t Ds,
                                         void* stream) {
    if (!depth_args_ok(g, ldg, out, ldo, B, D, S, C, Ds)) return CORRIF_EINVAL;
    const int64_t total = (int64_t)B * Ds * S * (C / 4);
    hipLaunchKernelGGL(depth_class_reduce_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, g, ldg, out, ldo, total,
                       make_daxis(Ds, D), C / 4, make_fastdiv((uint32_t)(C / 4)), make_fastdiv((uint32_t)S), make_fastdiv((uint32_t)Ds), (int)S);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    GRID_STRIDE(i, n) y[i] = gelu_erf(x[i]);
}
extern "C" int corrif_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return CORRIF_EINVAL;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t n) {
    GRID_STRIDE(i, n) dx[i] = dy[i] * gelu_erf_grad(x[i]);
}
extern "C" int corrif_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
    if (!dy || !x || !dx || n <= 0) return CORRIF_EINVAL;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, n);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, int64_t n) {
    GRID_STRIDE(i, n) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
extern "C" int corrif_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    if (!dy || !y || !dx || n <= 0) return CORRIF_EINVAL;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, n);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, int64_t n) {
    const float k = s[0];
    GRID_STRIDE(i, n) y[i] = x[i] * k;
}
extern "C" int corrif_scale_dev(const float* x, const float* scalar, float* y, int64_t n, void* stream) {
    if (!x || !scalar || !y || n <= 0) return CORRIF_EINVAL;
    hipLaunchKernelGGL(scale_dev_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, scalar, y, n);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float alpha) {
    GRID_STRIDE(i, n) y[i] = x[i] * alpha;
}
__global__ void scale4_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4, float alpha) {
    GRID_STRIDE(i, n4) reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(x)[i] * alpha;
}
extern "C" int corrif_scale(const float* x, float* y, int64_t n, float alpha, void* stream) {
    if (!x || !y || n <= 0) return CORRIF_EINVAL;
    if (!(n & 3) && al16(x) && al16(y)) hipLaunchKernelGGL(scale4_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4, alpha);
    else hipLaunchKernelGGL(scale_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, alpha);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
__global__ void fill_kernel(float* __restrict__ y, int64_t n, float v) {
    GRID_STRIDE(i, n) y[i] = v;
}
__global__ void fill4_kernel(float* __restrict__ y, int64_t n4, float v) {
    const f32x4 v4 = {v, v, v, v};
    GRID_STRIDE(i, n4) reinterpret_cast<f32x4*>(y)[i] = v4;
}
extern "C" int corrif_fill(float* y, int64_t n, float value, void* stream) {
    if (!y || n <= 0) return CORRIF_EINVAL;
    if (!(n & 3) && al16(y)) hipLaunchKernelGGL(fill4_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, y, n / 4, value);
    else hipLaunchKernelGGL(fill_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, y, n, value);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
__global__ void copy2d_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst, int64_t ldd, int64_t rows, int C4,
                              int acc) {
    int64_t total = rows * C4;
    GRID_STRIDE(i, total) {
        int64_t r = i / C4;
        int c = (int)(i - r * C4);
        f32x4 v = *reinterpret_cast<const f32x4*>(src + r * lds + c * 4);
        f32x4* d = reinterpret_cast<f32x4*>(dst + r * ldd + c * 4);
        if (acc) v += *d;
        *d = v;
    }
}
extern "C" int corrif_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t C, int32_t accumulate,
                             void* stream) {
    if (!src || !dst || rows <= 0 || C <= 0) return CORRIF_EINVAL;
    if ((C & 3) || (lds & 3) || (ldd & 3) || !al16(src) || !al16(dst)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(copy2d_kernel, dim3(nblocks(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, src, lds, dst, ldd, rows, C / 4,
                       (int)accumulate);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
// dst[g][i] = src_g[i]: the parameter tensors of G same-shaped modules gathered into one stacked buffer (grouped launches), one launch
struct GroupPtrs { const float* p[4]; };
__global__ void stack_groups_kernel(GroupPtrs src, float* __restrict__ dst, int64_t n) {
    const int g = blockIdx.y;
    const float* __restrict__ s = g == 0 ? src.p[0] : g == 1 ? src.p[1] : g == 2 ? src.p[2] : src.p[3];
    float* __restrict__ d = dst + (int64_t)g * n;
    GRID_STRIDE(i, n) d[i] = s[i];
}
extern "C" int corrif_stack_groups(const float* const* srcs, int32_t G, float* dst, int64_t n, void* stream) {
    if (!srcs || !dst || G < 1 || G > 4 || n <= 0) return CORRIF_EINVAL;
    GroupPtrs gp;
    for (int i = 0; i < 4; ++i) gp.p[i] = i < G ? srcs[i] : nullptr;
    for (int i = 0; i < G; ++i)
        if (!gp.p[i]) return CORRIF_EINVAL;
    hipLaunchKernelGGL(stack_groups_kernel, dim3(nblocks(n), (unsigned)G), dim3(256), 0, (hipStream_t)stream, gp, dst, n);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
__global__ void sum_groups_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t ge, int groups) {
    GRID_STRIDE(i, ge) {
        float s = 0.f;
        for (int g = 0; g < groups; ++g) s += x[(int64_t)g * ge + i];
        out[i] = s;
    }
}
extern "C" int corrif_sum_groups(const float* x, float* out, int64_t group_elems, int32_t groups, void* stream) {
    if (!x || !out || group_elems <= 0 || groups <= 0) return CORRIF_EINVAL;
    hipLaunchKernelGGL(sum_groups_kernel, dim3(nblocks(group_elems)), dim3(256), 0, (hipStream_t)stream, x, out, group_elems, (int)groups);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ weight re-layouts
__global__ void repack_kernel(const float* __restrict__ in, float* __restrict__ out, int O, int I, int T, int mode, int64_t ldo) {
    int64_t n = (int64_t)O * I * T;
    GRID_STRIDE(e, n) {
        if (mode == 0) {          // out[o][t][i] (row pitch ldo) = w[o][i][t] ; e enumerates the output
            int i = (int)(e % I);
            int64_t r = e / I;
            int t = (int)(r % T), o = (int)(r / T);
            out[(int64_t)o * ldo + (int64_t)t * I + i] = in[((int64_t)o * I + i) * T + t];
        } else if (mode == 1) {   // out[t][o][i] = w[o][i][t]
            int i = (int)(e % I);
            int64_t r = e / I;
            int o = (int)(r % O), t = (int)(r / O);
            out[e] = in[((int64_t)o * I + i) * T + t];
        } else if (mode == 2) {   // w[o][i][t] = in[o][t][i] (row pitch ldo)
            int t = (int)(e % T);
            int64_t r = e / T;
            int i = (int)(r % I), o = (int)(r / I);
            out[e] = in[(int64_t)o * ldo + (int64_t)t * I + i];
        } else if (mode == 3) {   // out[ch][o][t][c] = w[o][ch*cc + c][t]
            const int cc = (int)ldo;
            int c = (int)(e % cc);
            int64_t r = e / cc;
            int t = (int)(r % T); r /= T;
            int o = (int)(r % O), ch = (int)(r / O);
            out[e] = in[((int64_t)o * I + ch * cc + c) * T + t];
        } else {                  // out[ch][i][t'][c] = w[ch*cc + c][i][T-1-t']
            const int cc = (int)ldo;
            int c = (int)(e % cc);
            int64_t r = e / cc;
            int t = (int)(r % T); r /= T;
            int i = (int)(r % I), ch = (int)(r / I);
            out[e] = in[((int64_t)(ch * cc + c) * I + i) * T + (T - 1 - t)];
        }
    }
}
extern "C" int corrif_weight_repack(const float* in, float* out, int32_t O, int32_t I, int32_t T, int32_t mode, int64_t ldo, void* stream) {
    if (!in || !out || O <= 0 || I <= 0 || T <= 0 || mode < 0 || mode > 4) return CORRIF_EINVAL;
    if ((mode == 0 || mode == 2) && ldo < (int64_t)I * T) return CORRIF_EINVAL;
    if (mode == 3 && (ldo <= 0 || I % ldo)) return CORRIF_EINVAL;
    if (mode == 4 && (ldo <= 0 || O % ldo)) return CORRIF_EINVAL;
    hipLaunchKernelGGL(repack_kernel, dim3(nblocks((int64_t)O * I * T)), dim3(256), 0, (hipStream_t)stream, in, out, (int)O, (int)I, (int)T,
                       (int)mode, ldo);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ max-pool (1,3,3)/(1,2,2) pad (0,1,1)
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int8_t* __restrict__ idx, int BD, int H, int W,
                                   int Ho, int Wo, int C4, Dec5 q) {
    int64_t total = (int64_t)BD * Ho * Wo * C4;
    GRID_STRIDE(i, total) {
        int c, wo, ho, d0, b0;
        dec5((uint32_t)i, q, b0, d0, ho, wo, c);        // q.d extent = BD, so b0 == 0 and d0 = bd
        const int64_t bd = d0;
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {-1, -1, -1, -1};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                int h = 2 * ho - 1 + kh, w = 2 * wo - 1 + kw;
                if (h < 0 || h >= H || w < 0 || w >= W) continue;
                f32x4 xv = *reinterpret_cast<const f32x4*>(x + ((bd * H + h) * W + w) * (int64_t)(C4 * 4) + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (bi[e] < 0 || xv[e] > best[e] || xv[e] != xv[e]) { best[e] = xv[e]; bi[e] = kh * 3 + kw; }
            }
        *reinterpret_cast<f32x4*>(y + i * 4) = best;
        char4 pk = make_char4((char)bi[0], (char)bi[1], (char)bi[2], (char)bi[3]);
        *reinterpret_cast<char4*>(idx + i * 4) = pk;
    }
}
__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const int8_t* __restrict__ idx, float* __restrict__ dx, int BD, int H,
                                   int W, int Ho, int Wo, int C4, Dec5 q) {
    int64_t total = (int64_t)BD * H * W * C4;
    GRID_STRIDE(i, total) {
        int c, w, h, d0, b0;
        dec5((uint32_t)i, q, b0, d0, h, w, c);
        const int64_t bd = d0;
        f32x4 acc = {0, 0, 0, 0};
        // outputs whose window covers (h, w): 2*ho-1+kh == h  -> ho in {(h+1)/2 (kh = h+1-2ho)}, kh in 0..2
        for (int kh = 0; kh < 3; ++kh) {
            int t = h + 1 - kh;
            if (t < 0 || (t & 1)) continue;
            int ho = t >> 1;
            if (ho >= Ho) continue;
            for (int kw = 0; kw < 3; ++kw) {
                int u = w + 1 - kw;
                if (u < 0 || (u & 1)) continue;
                int wo = u >> 1;
                if (wo >= Wo) continue;
                int64_t o = (((bd * Ho + ho) * Wo + wo) * C4 + c) * 4;
                char4 id = *reinterpret_cast<const char4*>(idx + o);
                f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
                int code = kh * 3 + kw;
                if (id.x == code) acc[0] += g[0];
                if (id.y == code) acc[1] += g[1];
                if (id.z == code) acc[2] += g[2];
                if (id.w == code) acc[3] += g[3];
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = acc;
    }
}
extern "C" int corrif_maxpool133_fwd(const float* x, float* y, int8_t* idx, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                                     void* stream) {
    if (!x || !y || !idx || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return CORRIF_EINVAL;
    if ((C & 3) || !al16(x) || !al16(y) || ((uintptr_t)idx & 3)) return CORRIF_EUNSUPPORTED;
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    int64_t total = (int64_t)B * D * Ho * Wo * (C / 4);
    if (total >= ((int64_t)1 << 31)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, idx, B * D, (int)H, (int)W, Ho, Wo,
                       C / 4, make_dec5(B * D, Ho, Wo, C / 4));
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_maxpool133_bwd(const float* dy, const int8_t* idx, float* dx, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C,
                                     void* stream) {
    if (!dy || !dx || !idx || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return CORRIF_EINVAL;
    if ((C & 3) || !al16(dy) || !al16(dx) || ((uintptr_t)idx & 3)) return CORRIF_EUNSUPPORTED;
    int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    int64_t total = (int64_t)B * D * H * W * (C / 4);
    if (total >= ((int64_t)1 << 31)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, B * D, (int)H, (int)W, Ho, Wo,
                       C / 4, make_dec5(B * D, H, W, C / 4));
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ trilinear (align_corners=True), ATen index arithmetic
struct Axis { int in, out; float scale; };
static Axis make_axis(int in, int out) {
    Axis a;
    a.in = in; a.out = out;
    a.scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;      // area_pixel_compute_scale, align_corners
    return a;
}
__device__ __forceinline__ void axis_taps(const Axis& a, int o, int& i0, int& i1, float& l0, float& l1) {
    float r = a.scale * (float)o;
    i0 = (int)r;
    if (i0 > a.in - 1) i0 = a.in - 1;
    i1 = i0 + ((i0 < a.in - 1) ? 1 : 0);
    l1 = r - (float)i0;
    l0 = 1.0f - l1;
}
__global__ void trilinear_fwd_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int B, int C4, Axis ad,
                                     Axis ah, Axis aw, Dec5 q) {
    int64_t total = (int64_t)B * ad.out * ah.out * aw.out * C4;
    GRID_STRIDE(i, total) {
        int c, wo, ho, dd, bi;
        dec5((uint32_t)i, q, bi, dd, ho, wo, c);
        const int64_t b = bi;
        int d0, d1, h0, h1, w0, w1;
        float ld0, ld1, lh0, lh1, lw0, lw1;
        axis_taps(ad, dd, d0, d1, ld0, ld1);
        axis_taps(ah, ho, h0, h1, lh0, lh1);
        axis_taps(aw, wo, w0, w1, lw0, lw1);
        const float* xb = x + b * ad.in * ah.in * aw.in * ldx + c * 4;
        auto at = [&](int d, int h, int w) { return *reinterpret_cast<const f32x4*>(xb + ((int64_t)(d * ah.in + h) * aw.in + w) * ldx); };
        // same association as ATen's upsample_trilinear3d
        f32x4 o = ld0 * (lh0 * (lw0 * at(d0, h0, w0) + lw1 * at(d0, h0, w1)) + lh1 * (lw0 * at(d0, h1, w0) + lw1 * at(d0, h1, w1))) +
                  ld1 * (lh0 * (lw0 * at(d1, h0, w0) + lw1 * at(d1, h0, w1)) + lh1 * (lw0 * at(d1, h1, w0) + lw1 * at(d1, h1, w1)));
        int64_t orow = ((b * ad.out + dd) * ah.out + ho) * aw.out + wo;
        *reinterpret_cast<f32x4*>(y + orow * ldy + c * 4) = o;
    }
}
// weight with which output index o reads input index i along one axis (0 if it does not)
__device__ __forceinline__ float axis_weight(const Axis& a, int o, int i) {
    int i0, i1;
    float l0, l1;
    axis_taps(a, o, i0, i1, l0, l1);
    float w = 0.f;
    if (i0 == i) w += l0;
    if (i1 == i) w += l1;
    return w;
}
// conservative output range that can touch input index i
__device__ __forceinline__ void axis_range(const Axis& a, int i, int& lo, int& hi) {
    if (a.out == 1 || a.scale == 0.f) { lo = 0; hi = a.out - 1; return; }
    float inv = 1.0f / a.scale;
    lo = (int)floorf(((float)i - 1.0f) * inv) - 1;
    hi = (int)ceilf(((float)i + 1.0f) * inv) + 1;
    if (lo < 0) lo = 0;
    if (hi > a.out - 1) hi = a.out - 1;
}
// the outputs that read input index i along one axis and their weights, scanned ONCE per axis (up to NT of them: 2-5 for the x2
// up-samplings, 1-2 for the down-samplings to 8^3); returns the count, or -1 when there are more than NT (generic fall-back below)
template <int NT>
__device__ __forceinline__ int axis_tap_list(const Axis& a, int i, int (&idx)[NT], float (&wt)[NT]) {
    int lo, hi, n = 0;
    axis_range(a, i, lo, hi);
    for (int o = lo; o <= hi; ++o) {
        const float w = axis_weight(a, o, i);
        if (w == 0.f) continue;
        if (n == NT) return -1;
        idx[n] = o;
        wt[n] = w;
        ++n;
    }
    return n;
}
__global__ void trilinear_bwd_kernel(const float* __restrict__ dy, int64_t lddy, float* __restrict__ dx, int64_t lddx, int B, int C4,
                                     Axis ad, Axis ah, Axis aw, Dec5 q) {
    constexpr int NT = 6;
    int64_t total = (int64_t)B * ad.in * ah.in * aw.in * C4;
    GRID_STRIDE(i, total) {
        int c, wi, hi_, di, bi;
        dec5((uint32_t)i, q, bi, di, hi_, wi, c);
        const int64_t b = bi;
        f32x4 acc = {0, 0, 0, 0};
        const float* gb = dy + b * ad.out * ah.out * aw.out * lddy + c * 4;
        int id[NT], ih[NT], iw[NT];
        float fd[NT], fh[NT], fw[NT];
        const int nd = axis_tap_list<NT>(ad, di, id, fd), nh = axis_tap_list<NT>(ah, hi_, ih, fh), nw = axis_tap_list<NT>(aw, wi, iw, fw);
        if (nd >= 0 && nh >= 0 && nw >= 0) {
            // same summation order as the generic loop (ascending output index per axis, weight product (wd * wh) * ww): bit-identical
            for (int a = 0; a < nd; ++a)
                for (int e = 0; e < nh; ++e) {
                    const float wdh = fd[a] * fh[e];
                    const float* row = gb + (int64_t)(id[a] * ah.out + ih[e]) * aw.out * lddy;
                    for (int f = 0; f < nw; ++f) acc += (wdh * fw[f]) * *reinterpret_cast<const f32x4*>(row + (int64_t)iw[f] * lddy);
                }
        } else {
            int dlo, dhi, hlo, hhi, wlo, whi;
            axis_range(ad, di, dlo, dhi);
            axis_range(ah, hi_, hlo, hhi);
            axis_range(aw, wi, wlo, whi);
            for (int dd = dlo; dd <= dhi; ++dd) {
                float wd = axis_weight(ad, dd, di);
                if (wd == 0.f) continue;
                for (int ho = hlo; ho <= hhi; ++ho) {
                    float wh = axis_weight(ah, ho, hi_);
                    if (wh == 0.f) continue;
                    for (int wo = wlo; wo <= whi; ++wo) {
                        float ww = axis_weight(aw, wo, wi);
                        if (ww == 0.f) continue;
                        f32x4 g = *reinterpret_cast<const f32x4*>(gb + ((int64_t)(dd * ah.out + ho) * aw.out + wo) * lddy);
                        acc += (wd * wh * ww) * g;
                    }
                }
            }
        }
        int64_t irow = ((b * ad.in + di) * ah.in + hi_) * aw.in + wi;
        *reinterpret_cast<f32x4*>(dx + irow * lddx + c * 4) = acc;
    }
}
// One axis of the adjoint at a time (the x2 up-samplings of the decoder, mmvit4.py:243): the gather above reads up to 4^3 incoming
// gradients per input voxel through the L1 (measured 20.5 GB of fetches for 4.8 GB of tensors at 64^3 -> 128^3); the interpolation
// separates, so three streaming passes (D, then H, then W) read every incoming value exactly once.  A thread owns one float4 column of
// one (outer, inner) position and walks the output axis in ascending order, carrying the two input cells the current output touches -
// the same ascending-output summation order per input cell as the gather.
__global__ void trilinear_axis_adjoint_kernel(const float* __restrict__ src, float* __restrict__ dst, Axis a, FastDiv inner4, uint32_t total) {
    GRID_STRIDE(i, total) {
        const uint32_t o = fdiv((uint32_t)i, inner4), j = (uint32_t)i - o * inner4.d;
        const int64_t st = inner4.d;
        const f32x4* s = reinterpret_cast<const f32x4*>(src) + (int64_t)o * a.out * st + j;
        f32x4* d = reinterpret_cast<f32x4*>(dst) + (int64_t)o * a.in * st + j;
        f32x4 cur = {0, 0, 0, 0}, nxt = {0, 0, 0, 0};
        int p = 0;
#pragma unroll 4
        for (int t = 0; t < a.out; ++t) {
            int i0, i1;
            float l0, l1;
            axis_taps(a, t, i0, i1, l0, l1);
            const f32x4 g = __builtin_nontemporal_load(s + (int64_t)t * st);
            while (p < i0) {
                d[(int64_t)p * st] = cur;
                cur = nxt;
                nxt = f32x4{0, 0, 0, 0};
                ++p;
            }
            if (i1 == i0) {
                cur += (l0 + l1) * g;
            } else {
                cur += l0 * g;
                nxt += l1 * g;
            }
        }
        for (; p < a.in; ++p) {
            d[(int64_t)p * st] = cur;
            cur = nxt;
            nxt = f32x4{0, 0, 0, 0};
        }
    }
}
static bool resample_ok(const void* a, int64_t lda, const void* b, int64_t ldb, int B, int C, int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
    return a && b && B > 0 && C > 0 && !(C & 3) && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0 && !(lda & 3) && !(ldb & 3) &&
           al16(a) && al16(b);
}
extern "C" int corrif_trilinear_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t B, int32_t C, int32_t Di, int32_t Hi,
                                    int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, void* stream) {
    if (!resample_ok(x, ldx, y, ldy, B, C, Di, Hi, Wi, Do, Ho, Wo)) return CORRIF_EINVAL;
    int64_t total = (int64_t)B * Do * Ho * Wo * (C / 4);
    if (total >= ((int64_t)1 << 31)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(trilinear_fwd_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, (int)B, C / 4,
                       make_axis(Di, Do), make_axis(Hi, Ho), make_axis(Wi, Wo), make_dec5(Do, Ho, Wo, C / 4));
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_trilinear_bwd(const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t B, int32_t C, int32_t Di, int32_t Hi,
                                    int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, void* stream) {
    if (!resample_ok(dy, lddy, dx, lddx, B, C, Di, Hi, Wi, Do, Ho, Wo)) return CORRIF_EINVAL;
    int64_t total = (int64_t)B * Di * Hi * Wi * (C / 4);
    if (total >= ((int64_t)1 << 31)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(trilinear_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx, lddx, (int)B, C / 4,
                       make_axis(Di, Do), make_axis(Hi, Ho), make_axis(Wi, Wo), make_dec5(Di, Hi, Wi, C / 4));
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ nearest (ATen: src = min(floor(dst * in/out), in-1))
struct NAxis { int in, out; float scale; };
static NAxis make_naxis(int in, int out) { NAxis a; a.in = in; a.out = out; a.scale = (float)in / (float)out; return a; }
__device__ __forceinline__ int nearest_src(const NAxis& a, int o) {
    int s = (int)floorf((float)o * a.scale);
    return s < a.in - 1 ? s : a.in - 1;
}
__global__ void nearest_fwd_kernel(const float* __restrict__ x, int64_t ldx, float* __restrict__ y, int64_t ldy, int B, int C4, NAxis ad,
                                   NAxis ah, NAxis aw, Dec5 q) {
    int64_t total = (int64_t)B * ad.out * ah.out * aw.out * C4;
    GRID_STRIDE(i, total) {
        int c, wo, ho, dd, bi;
        dec5((uint32_t)i, q, bi, dd, ho, wo, c);
        const int64_t b = bi;
        int64_t irow = ((b * ad.in + nearest_src(ad, dd)) * ah.in + nearest_src(ah, ho)) * aw.in + nearest_src(aw, wo);
        int64_t orow = ((b * ad.out + dd) * ah.out + ho) * aw.out + wo;
        *reinterpret_cast<f32x4*>(y + orow * ldy + c * 4) = *reinterpret_cast<const f32x4*>(x + irow * ldx + c * 4);
    }
}
__device__ __forceinline__ void nearest_range(const NAxis& a, int i, int& lo, int& hi) {
    float inv = 1.0f / a.scale;
    lo = (int)floorf((float)i * inv) - 1;
    hi = (int)ceilf((float)(i + 1) * inv) + 1;
    if (lo < 0) lo = 0;
    if (hi > a.out - 1) hi = a.out - 1;
    while (lo <= hi && nearest_src(a, lo) != i) ++lo;
    while (hi >= lo && nearest_src(a, hi) != i) --hi;
}
__global__ void nearest_bwd_kernel(const float* __restrict__ dy, int64_t lddy, float* __restrict__ dx, int64_t lddx, int B, int C4, NAxis ad,
                                   NAxis ah, NAxis aw, Dec5 q) {
    int64_t total = (int64_t)B * ad.in * ah.in * aw.in * C4;
    GRID_STRIDE(i, total) {
        int c, wi, hi_, di, bi;
        dec5((uint32_t)i, q, bi, di, hi_, wi, c);
        const int64_t b = bi;
        int dlo, dhi, hlo, hhi, wlo, whi;
        nearest_range(ad, di, dlo, dhi);
        nearest_range(ah, hi_, hlo, hhi);
        nearest_range(aw, wi, wlo, whi);
        f32x4 acc = {0, 0, 0, 0};
        const float* gb = dy + b * ad.out * ah.out * aw.out * lddy + c * 4;
        for (int dd = dlo; dd <= dhi; ++dd)
            for (int ho = hlo; ho <= hhi; ++ho)
                for (int wo = wlo; wo <= whi; ++wo)
                    acc += *reinterpret_cast<const f32x4*>(gb + ((int64_t)(dd * ah.out + ho) * aw.out + wo) * lddy);
        int64_t irow = ((b * ad.in + di) * ah.in + hi_) * aw.in + wi;
        *reinterpret_cast<f32x4*>(dx + irow * lddx + c * 4) = acc;
    }
}
static bool sep_ok(int B, int C, int Di, int Hi, int Wi, int Do, int Ho, int Wo) {
    if (B <= 0 || C <= 0 || (C & 3) || Di <= 0 || Hi <= 0 || Wi <= 0 || Do < Di || Ho < Hi || Wo < Wi) return false;
    return (int64_t)B * Di * Ho * Wo * (C / 4) < ((int64_t)1 << 31) && (int64_t)B * Do * Ho * Wo * C < ((int64_t)1 << 40);
}
extern "C" int64_t corrif_trilinear_bwd_sep_workspace(int32_t B, int32_t C, int32_t Di, int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho,
                                                      int32_t Wo) {
    if (!sep_ok(B, C, Di, Hi, Wi, Do, Ho, Wo)) return -1;
    return ((int64_t)B * Di * Ho * Wo * C + (int64_t)B * Di * Hi * Wo * C) * (int64_t)sizeof(float);
}
extern "C" int corrif_trilinear_bwd_sep(const float* dy, float* dx, float* ws, int32_t B, int32_t C, int32_t Di, int32_t Hi, int32_t Wi,
                                        int32_t Do, int32_t Ho, int32_t Wo, void* stream) {
    if (!dy || !dx || !ws || !al16(dy) || !al16(dx) || !al16(ws)) return CORRIF_EINVAL;
    if (!sep_ok(B, C, Di, Hi, Wi, Do, Ho, Wo)) return CORRIF_EUNSUPPORTED;
    float* t1 = ws;                                           // [B, Di, Ho, Wo, C]
    float* t2 = ws + (int64_t)B * Di * Ho * Wo * C;           // [B, Di, Hi, Wo, C]
    const int C4 = C / 4;
    struct Pass { const float* s; float* d; Axis a; int64_t outer, inner4; };
    const Pass ps[3] = {{dy, t1, make_axis(Di, Do), (int64_t)B, (int64_t)Ho * Wo * C4},
                        {t1, t2, make_axis(Hi, Ho), (int64_t)B * Di, (int64_t)Wo * C4},
                        {t2, dx, make_axis(Wi, Wo), (int64_t)B * Di * Hi, (int64_t)C4}};
    for (const Pass& p : ps) {
        const int64_t total = p.outer * p.inner4;
        hipLaunchKernelGGL(trilinear_axis_adjoint_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, p.s, p.d, p.a,
                           make_fastdiv((uint32_t)p.inner4), (uint32_t)total);
        CORRIF_CHECK_LAUNCH();
    }
    return CORRIF_OK;
}
extern "C" int corrif_nearest_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t B, int32_t C, int32_t Di, int32_t Hi, int32_t Wi,
                                  int32_t Do, int32_t Ho, int32_t Wo, void* stream) {
    if (!resample_ok(x, ldx, y, ldy, B, C, Di, Hi, Wi, Do, Ho, Wo)) return CORRIF_EINVAL;
    int64_t total = (int64_t)B * Do * Ho * Wo * (C / 4);
    if (total >= ((int64_t)1 << 31)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(nearest_fwd_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, (int)B, C / 4,
                       make_naxis(Di, Do), make_naxis(Hi, Ho), make_naxis(Wi, Wo), make_dec5(Do, Ho, Wo, C / 4));
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_nearest_bwd(const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t B, int32_t C, int32_t Di, int32_t Hi,
                                  int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, void* stream) {
    if (!resample_ok(dy, lddy, dx, lddx, B, C, Di, Hi, Wi, Do, Ho, Wo)) return CORRIF_EINVAL;
    int64_t total = (int64_t)B * Di * Hi * Wi * (C / 4);
    if (total >= ((int64_t)1 << 31)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(nearest_bwd_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, dy, lddy, dx, lddx, (int)B, C / 4,
                       make_naxis(Di, Do), make_naxis(Hi, Ho), make_naxis(Wi, Wo), make_dec5(Di, Hi, Wi, C / 4));
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ adjoint of replicate padding (pad 1 on every side)
__global__ void pad_fold_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int64_t lddx, int B, int D, int H, int W, int C4,
                                Dec5 q) {
    int64_t total = (int64_t)B * D * H * W * C4;
    const int Dp = D + 2, Hp = H + 2, Wp = W + 2;
    GRID_STRIDE(i, total) {
        int c, w, h, d, bi;
        dec5((uint32_t)i, q, bi, d, h, w, c);
        const int64_t b = bi;
        int d0 = d == 0 ? 0 : d + 1, d1 = d == D - 1 ? D + 1 : d + 1;
        int h0 = h == 0 ? 0 : h + 1, h1 = h == H - 1 ? H + 1 : h + 1;
        int w0 = w == 0 ? 0 : w + 1, w1 = w == W - 1 ? W + 1 : w + 1;
        f32x4 acc = {0, 0, 0, 0};
        for (int pd = d0; pd <= d1; ++pd)
            for (int ph = h0; ph <= h1; ++ph)
                for (int pw = w0; pw <= w1; ++pw)
                    acc += *reinterpret_cast<const f32x4*>(dxp + ((((b * Dp + pd) * Hp + ph) * Wp + pw) * C4 + c) * 4);
        *reinterpret_cast<f32x4*>(dx + (((b * D + d) * H + h) * W + w) * lddx + c * 4) = acc;
    }
}
extern "C" int corrif_pad_fold(const float* dxp, float* dx, int64_t lddx, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, void* stream) {
    if (!dxp || !dx || B <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return CORRIF_EINVAL;
    if ((C & 3) || (lddx & 3) || !al16(dxp) || !al16(dx)) return CORRIF_EUNSUPPORTED;
    int64_t total = (int64_t)B * D * H * W * (C / 4);
    if (total >= ((int64_t)1 << 31)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(pad_fold_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, dxp, dx, lddx, (int)B, (int)D, (int)H, (int)W,
                       C / 4, make_dec5(D, H, W, C / 4));
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ Philox4x32-10 (shared by dropout and the fused softmax+dropout)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32_10(uint64_t ctr, uint64_t seed, uint32_t (&out)[4]) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) out[e] = c[e];
}
// ------------------------------------------------------------------ row softmax (one wave per row, row kept in registers)
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
template <int NV>   // n = NV*256
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int64_t rows, float scale) {
    constexpr int n = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k] = *reinterpret_cast<const f32x4*>(s + row * n + (k * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[k][e] *= scale; m = fmaxf(m, v[k][e]); }
    }
    m = wave_max_f(m);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[k][e] = expf(v[k][e] - m); sum += v[k][e]; }
    const float inv = 1.0f / wave_sum_f(sum);
#pragma unroll
    for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4*>(s + row * n + (k * 64 + lane) * 4) = v[k] * inv;
}
// softmax + attention dropout in one pass: P (kept for the backward) and P' = P * keep/(1-p) (fed to the P.V product)
template <int NV>
__global__ __launch_bounds__(256) void softmax_dropout_rows_kernel(float* __restrict__ s, float* __restrict__ pd, int64_t rows, float scale,
                                                                   float pdrop, float inv_keep, uint64_t seed, uint64_t offset4) {
    constexpr int n = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[NV];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k] = *reinterpret_cast<const f32x4*>(s + row * n + (k * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[k][e] *= scale; m = fmaxf(m, v[k][e]); }
    }
    m = wave_max_f(m);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[k][e] = expf(v[k][e] - m); sum += v[k][e]; }
    const float inv = 1.0f / wave_sum_f(sum);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int64_t idx4 = (row * n) / 4 + k * 64 + lane;
        f32x4 pv = v[k] * inv;
        *reinterpret_cast<f32x4*>(s + idx4 * 4) = pv;
        uint32_t r[4];
        philox4x32_10(offset4 + (uint64_t)idx4, seed, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = (float)(r[e] >> 8) * (1.0f / 16777216.0f);
            pv[e] = (u >= pdrop) ? pv[e] * inv_keep : 0.f;
        }
        *reinterpret_cast<f32x4*>(pd + idx4 * 4) = pv;
    }
}
// backward of the pair: g = dP' * keep/(1-p) ; dS = scale * P * (g - sum(P*g)), in place on the dP' buffer
template <int NV>
__global__ __launch_bounds__(256) void softmax_dropout_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ g, int64_t rows,
                                                                       float scale, float pdrop, float inv_keep, uint64_t seed,
                                                                       uint64_t offset4) {
    constexpr int n = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 pv[NV], gv[NV];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int64_t idx4 = (row * n) / 4 + k * 64 + lane;
        pv[k] = *reinterpret_cast<const f32x4*>(p + idx4 * 4);
        gv[k] = *reinterpret_cast<const f32x4*>(g + idx4 * 4);
        uint32_t r[4];
        philox4x32_10(offset4 + (uint64_t)idx4, seed, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = (float)(r[e] >> 8) * (1.0f / 16777216.0f);
            gv[k][e] = (u >= pdrop) ? gv[k][e] * inv_keep : 0.f;
            dot += pv[k][e] * gv[k][e];
        }
    }
    dot = wave_sum_f(dot);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = scale * pv[k][e] * (gv[k][e] - dot);
        *reinterpret_cast<f32x4*>(g + row * n + (k * 64 + lane) * 4) = o;
    }
}
template <int NV>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ g, int64_t rows, float scale) {
    constexpr int n = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 pv[NV], gv[NV];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        pv[k] = *reinterpret_cast<const f32x4*>(p + row * n + (k * 64 + lane) * 4);
        gv[k] = *reinterpret_cast<const f32x4*>(g + row * n + (k * 64 + lane) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) dot += pv[k][e] * gv[k][e];
    }
    dot = wave_sum_f(dot);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = scale * pv[k][e] * (gv[k][e] - dot);
        *reinterpret_cast<f32x4*>(g + row * n + (k * 64 + lane) * 4) = o;
    }
}
extern "C" int corrif_softmax_rows(float* s, int64_t rows, int32_t n, float scale, void* stream) {
    if (!s || rows <= 0) return CORRIF_EINVAL;
    if (!al16(s)) return CORRIF_EUNSUPPORTED;
    dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (n == 256) hipLaunchKernelGGL((softmax_rows_kernel<1>), grid, dim3(256), 0, st, s, rows, scale);
    else if (n == 512) hipLaunchKernelGGL((softmax_rows_kernel<2>), grid, dim3(256), 0, st, s, rows, scale);
    else if (n == 1024) hipLaunchKernelGGL((softmax_rows_kernel<4>), grid, dim3(256), 0, st, s, rows, scale);
    else if (n == 1536) hipLaunchKernelGGL((softmax_rows_kernel<6>), grid, dim3(256), 0, st, s, rows, scale);      // MMVit2's 3 x 512 multimodal tokens
    else if (n == 2048) hipLaunchKernelGGL((softmax_rows_kernel<8>), grid, dim3(256), 0, st, s, rows, scale);
    else return CORRIF_EUNSUPPORTED;
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_softmax_dropout_rows(float* s, float* pd, int64_t rows, int32_t n, float scale, float p, uint64_t seed, uint64_t offset,
                                           void* stream) {
    if (!s || !pd || rows <= 0 || !(p >= 0.f && p < 1.f)) return CORRIF_EINVAL;
    if (!al16(s) || !al16(pd) || (offset & 3)) return CORRIF_EUNSUPPORTED;
    dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const float ik = 1.0f / (1.0f - p);
    if (n == 512) hipLaunchKernelGGL((softmax_dropout_rows_kernel<2>), grid, dim3(256), 0, st, s, pd, rows, scale, p, ik, seed, offset / 4);
    else if (n == 1024) hipLaunchKernelGGL((softmax_dropout_rows_kernel<4>), grid, dim3(256), 0, st, s, pd, rows, scale, p, ik, seed, offset / 4);
    else if (n == 1536) hipLaunchKernelGGL((softmax_dropout_rows_kernel<6>), grid, dim3(256), 0, st, s, pd, rows, scale, p, ik, seed, offset / 4);      // MMVit2's 3 x 512 multimodal tokens
    else if (n == 2048) hipLaunchKernelGGL((softmax_dropout_rows_kernel<8>), grid, dim3(256), 0, st, s, pd, rows, scale, p, ik, seed, offset / 4);
    else return CORRIF_EUNSUPPORTED;
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_softmax_dropout_rows_bwd(const float* pr, float* dpd_to_ds, int64_t rows, int32_t n, float scale, float p, uint64_t seed,
                                               uint64_t offset, void* stream) {
    if (!pr || !dpd_to_ds || rows <= 0 || !(p >= 0.f && p < 1.f)) return CORRIF_EINVAL;
    if (!al16(pr) || !al16(dpd_to_ds) || (offset & 3)) return CORRIF_EUNSUPPORTED;
    dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    const float ik = 1.0f / (1.0f - p);
    if (n == 512) hipLaunchKernelGGL((softmax_dropout_rows_bwd_kernel<2>), grid, dim3(256), 0, st, pr, dpd_to_ds, rows, scale, p, ik, seed, offset / 4);
    else if (n == 1024) hipLaunchKernelGGL((softmax_dropout_rows_bwd_kernel<4>), grid, dim3(256), 0, st, pr, dpd_to_ds, rows, scale, p, ik, seed, offset / 4);
    else if (n == 1536) hipLaunchKernelGGL((softmax_dropout_rows_bwd_kernel<6>), grid, dim3(256), 0, st, pr, dpd_to_ds, rows, scale, p, ik, seed, offset / 4);      // MMVit2's 3 x 512 multimodal tokens
    else if (n == 2048) hipLaunchKernelGGL((softmax_dropout_rows_bwd_kernel<8>), grid, dim3(256), 0, st, pr, dpd_to_ds, rows, scale, p, ik, seed, offset / 4);
    else return CORRIF_EUNSUPPORTED;
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_softmax_rows_bwd(const float* p, float* dp_to_ds, int64_t rows, int32_t n, float scale, void* stream) {
    if (!p || !dp_to_ds || rows <= 0) return CORRIF_EINVAL;
    if (!al16(p) || !al16(dp_to_ds)) return CORRIF_EUNSUPPORTED;
    dim3 grid((unsigned)((rows + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    if (n == 256) hipLaunchKernelGGL((softmax_rows_bwd_kernel<1>), grid, dim3(256), 0, st, p, dp_to_ds, rows, scale);
    else if (n == 512) hipLaunchKernelGGL((softmax_rows_bwd_kernel<2>), grid, dim3(256), 0, st, p, dp_to_ds, rows, scale);
    else if (n == 1024) hipLaunchKernelGGL((softmax_rows_bwd_kernel<4>), grid, dim3(256), 0, st, p, dp_to_ds, rows, scale);
    else if (n == 1536) hipLaunchKernelGGL((softmax_rows_bwd_kernel<6>), grid, dim3(256), 0, st, p, dp_to_ds, rows, scale);      // MMVit2's 3 x 512 multimodal tokens
    else if (n == 2048) hipLaunchKernelGGL((softmax_rows_bwd_kernel<8>), grid, dim3(256), 0, st, p, dp_to_ds, rows, scale);
    else return CORRIF_EUNSUPPORTED;
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ dropout, Philox4x32-10 counter stream
__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n4, float p, float inv_keep, uint64_t seed,
                               uint64_t offset4) {
    GRID_STRIDE(i, n4) {
        uint32_t r[4];
        philox4x32_10(offset4 + (uint64_t)i, seed, r);
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = (float)(r[e] >> 8) * (1.0f / 16777216.0f);      // 24-bit uniform in [0,1)
            v[e] = (u >= p) ? v[e] * inv_keep : 0.f;
        }
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}
extern "C" int corrif_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream) {
    if (!x || !y || n <= 0 || !(p >= 0.f && p < 1.f)) return CORRIF_EINVAL;
    if ((n & 3) || (offset & 3) || !al16(x) || !al16(y)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(dropout_kernel, dim3(nblocks(n / 4)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4, p, 1.0f / (1.0f - p), seed,
                       offset / 4);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ inter-modal correlation (mmvit4.py:481-491)
struct QkvPtrs { const float* p[3]; };
struct OutPtrs { float* p[3]; };
__device__ __forceinline__ void softmax3(float s0, float s1, float s2, float (&w)[3]) {
    const float k = 0.57735026918962576451f;      // 1/sqrt(3)
    s0 *= k; s1 *= k; s2 *= k;
    float m = fmaxf(s0, fmaxf(s1, s2));
    float e0 = expf(s0 - m), e1 = expf(s1 - m), e2 = expf(s2 - m);
    float inv = 1.0f / (e0 + e1 + e2);
    w[0] = e0 * inv; w[1] = e1 * inv; w[2] = e2 * inv;
}
__global__ void intercorr_fwd_kernel(QkvPtrs in, int64_t ldq, OutPtrs out, int64_t ldo, int B, int S, int C) {
    const int C4 = C / 4;
    int64_t total = (int64_t)B * S * C4;
    GRID_STRIDE(i, total) {
        int c = (int)(i % C4) * 4;
        int64_t v = i / C4;
        int s = (int)(v % S);
        int bp = (int)(v / S);
        f32x4 acc[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[m] = (f32x4){0, 0, 0, 0};
#pragma unroll
        for (int ip = 0; ip < 3; ++ip) {
            int lin = 3 * bp + ip;
            int ik = lin / B, b = lin - ik * B;               // (i, b) = divmod(3 b' + i', B)
            int64_t rb = ((int64_t)b * S + s) * ldq + c;
            f32x4 k0 = *reinterpret_cast<const f32x4*>(in.p[0] + rb + C);
            f32x4 k1 = *reinterpret_cast<const f32x4*>(in.p[1] + rb + C);
            f32x4 k2 = *reinterpret_cast<const f32x4*>(in.p[2] + rb + C);
            f32x4 vv = *reinterpret_cast<const f32x4*>(in.p[ip] + ((int64_t)bp * S + s) * ldq + c + 2 * C);
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                f32x4 q = *reinterpret_cast<const f32x4*>(in.p[m] + rb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float w[3];
                    softmax3(q[e] * k0[e], q[e] * k1[e], q[e] * k2[e], w);
                    float ws = ik == 0 ? w[0] : (ik == 1 ? w[1] : w[2]);
                    acc[m][e] += ws * vv[e];
                }
            }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) *reinterpret_cast<f32x4*>(out.p[m] + ((int64_t)bp * S + s) * ldo + c) = acc[m];
    }
}
struct CPtrs { const float* p[3]; };
__global__ void intercorr_bwd_kernel(QkvPtrs in, int64_t ldq, CPtrs dout, int64_t ldo, OutPtrs dq, int B, int S, int C) {
    const int C4 = C / 4;
    int64_t total = (int64_t)B * S * C4;
    GRID_STRIDE(i, total) {
        int c = (int)(i % C4) * 4;
        int64_t v = i / C4;
        int s = (int)(v % S);
        int b = (int)(v / S);
        const int64_t rb = ((int64_t)b * S + s) * ldq + c;
        // ---- phase 1: this thread as the SOURCE sample b of the softmax: dq_m[b], dk_i[b]
        f32x4 k[3], dk[3];
#pragma unroll
        for (int ik = 0; ik < 3; ++ik) {
            k[ik] = *reinterpret_cast<const f32x4*>(in.p[ik] + rb + C);
            dk[ik] = (f32x4){0, 0, 0, 0};
        }
        // (b', i') for each key index i: 3b' + i' = i*B + b
        int bp_[3], ip_[3];
#pragma unroll
        for (int ik = 0; ik < 3; ++ik) { int lin = ik * B + b; bp_[ik] = lin / 3; ip_[ik] = lin - 3 * bp_[ik]; }
        f32x4 vsel[3];
#pragma unroll
        for (int ik = 0; ik < 3; ++ik)
            vsel[ik] = *reinterpret_cast<const f32x4*>(in.p[ip_[ik]] + ((int64_t)bp_[ik] * S + s) * ldq + c + 2 * C);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            f32x4 q = *reinterpret_cast<const f32x4*>(in.p[m] + rb);
            f32x4 g[3];
#pragma unroll
            for (int ik = 0; ik < 3; ++ik) g[ik] = *reinterpret_cast<const f32x4*>(dout.p[m] + ((int64_t)bp_[ik] * S + s) * ldo + c);
            f32x4 dqv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w[3];
                softmax3(q[e] * k[0][e], q[e] * k[1][e], q[e] * k[2][e], w);
                float dw0 = g[0][e] * vsel[0][e], dw1 = g[1][e] * vsel[1][e], dw2 = g[2][e] * vsel[2][e];
                float dot = w[0] * dw0 + w[1] * dw1 + w[2] * dw2;
                const float kk = 0.57735026918962576451f;
                float ds0 = w[0] * (dw0 - dot) * kk, ds1 = w[1] * (dw1 - dot) * kk, ds2 = w[2] * (dw2 - dot) * kk;
                dqv[e] = ds0 * k[0][e] + ds1 * k[1][e] + ds2 * k[2][e];
                dk[0][e] += ds0 * q[e];
                dk[1][e] += ds1 * q[e];
                dk[2][e] += ds2 * q[e];
            }
            *reinterpret_cast<f32x4*>(dq.p[m] + rb) = dqv;
        }
#pragma unroll
        for (int ik = 0; ik < 3; ++ik) *reinterpret_cast<f32x4*>(dq.p[ik] + rb + C) = dk[ik];
        // ---- phase 2: this thread as the DESTINATION sample b' = b: dv_i'[b'] = sum_m w_m[i][bsrc] * dout_m[b']
        f32x4 g2[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) g2[m] = *reinterpret_cast<const f32x4*>(dout.p[m] + ((int64_t)b * S + s) * ldo + c);
#pragma unroll
        for (int ip = 0; ip < 3; ++ip) {
            int lin = 3 * b + ip;
            int ik = lin / B, bs = lin - ik * B;
            const int64_t rs = ((int64_t)bs * S + s) * ldq + c;
            f32x4 k0 = *reinterpret_cast<const f32x4*>(in.p[0] + rs + C);
            f32x4 k1 = *reinterpret_cast<const f32x4*>(in.p[1] + rs + C);
            f32x4 k2 = *reinterpret_cast<const f32x4*>(in.p[2] + rs + C);
            f32x4 dv = {0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                f32x4 q = *reinterpret_cast<const f32x4*>(in.p[m] + rs);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float w[3];
                    softmax3(q[e] * k0[e], q[e] * k1[e], q[e] * k2[e], w);
                    float ws = ik == 0 ? w[0] : (ik == 1 ? w[1] : w[2]);
                    dv[e] += ws * g2[m][e];
                }
            }
            *reinterpret_cast<f32x4*>(dq.p[ip] + rb + 2 * C) = dv;
        }
    }
}
extern "C" int corrif_intercorr_fwd(const float* qkv0, const float* qkv1, const float* qkv2, int64_t ldq, float* out0, float* out1,
                                    float* out2, int64_t ldo, int32_t B, int32_t S, int32_t C, void* stream) {
    if (!qkv0 || !qkv1 || !qkv2 || !out0 || !out1 || !out2 || B <= 0 || S <= 0 || C <= 0) return CORRIF_EINVAL;
    if ((C & 3) || (ldq & 3) || (ldo & 3) || ldq < 3 * (int64_t)C || !al16(qkv0) || !al16(qkv1) || !al16(qkv2) || !al16(out0) || !al16(out1) ||
        !al16(out2))
        return CORRIF_EUNSUPPORTED;
    QkvPtrs in = {{qkv0, qkv1, qkv2}};
    OutPtrs out = {{out0, out1, out2}};
    hipLaunchKernelGGL(intercorr_fwd_kernel, dim3(nblocks((int64_t)B * S * (C / 4))), dim3(256), 0, (hipStream_t)stream, in, ldq, out, ldo,
                       (int)B, (int)S, (int)C);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_intercorr_bwd(const float* qkv0, const float* qkv1, const float* qkv2, int64_t ldq, const float* do0, const float* do1,
                                    const float* do2, int64_t ldo, float* dqkv0, float* dqkv1, float* dqkv2, int32_t B, int32_t S, int32_t C,
                                    void* stream) {
    if (!qkv0 || !qkv1 || !qkv2 || !do0 || !do1 || !do2 || !dqkv0 || !dqkv1 || !dqkv2 || B <= 0 || S <= 0 || C <= 0) return CORRIF_EINVAL;
    if ((C & 3) || (ldq & 3) || (ldo & 3) || ldq < 3 * (int64_t)C || !al16(qkv0) || !al16(qkv1) || !al16(qkv2) || !al16(do0) || !al16(do1) ||
        !al16(do2) || !al16(dqkv0) || !al16(dqkv1) || !al16(dqkv2))
        return CORRIF_EUNSUPPORTED;
    QkvPtrs in = {{qkv0, qkv1, qkv2}};
    CPtrs dout = {{do0, do1, do2}};
    OutPtrs dq = {{dqkv0, dqkv1, dqkv2}};
    hipLaunchKernelGGL(intercorr_bwd_kernel, dim3(nblocks((int64_t)B * S * (C / 4))), dim3(256), 0, (hipStream_t)stream, in, ldq, dout, ldo, dq,
                       (int)B, (int)S, (int)C);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ head: 1x1x1 conv 8->3 + sigmoid, NCDHW output
__global__ void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                float* __restrict__ pred, int B, int HW) {
    int64_t total = (int64_t)B * HW;
    GRID_STRIDE(i, total) {
        f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8), b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        int64_t bb = i / HW, hw = i - bb * HW;
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float acc = bias[o];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += a[e] * w[o * 8 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += b[e] * w[o * 8 + 4 + e];
            pred[(bb * 3 + o) * HW + hw] = 1.0f / (1.0f + expf(-acc));
        }
    }
}
// dx + per-block partials of dw[3][8], db[3] (27 values) in double
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ pred,
                                                       const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ dx,
                                                       double* __restrict__ part, int B, int HW) {
    __shared__ double red[4][27];
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0;
    int64_t total = (int64_t)B * HW;
    GRID_STRIDE(i, total) {
        f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 8), b = *reinterpret_cast<const f32x4*>(x + i * 8 + 4);
        float xv[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        int64_t bb = i / HW, hw = i - bb * HW;
        float d[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            float p = pred[(bb * 3 + o) * HW + hw];
            float gl = dpred[(bb * 3 + o) * HW + hw] * p * (1.0f - p);
            acc[24 + o] += (double)gl;
#pragma unroll
            for (int e = 0; e < 8; ++e) { d[e] += gl * w[o * 8 + e]; acc[o * 8 + e] += (double)(gl * xv[e]); }
        }
        *reinterpret_cast<f32x4*>(dx + i * 8) = (f32x4){d[0], d[1], d[2], d[3]};
        *reinterpret_cast<f32x4*>(dx + i * 8 + 4) = (f32x4){d[4], d[5], d[6], d[7]};
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        double v = acc[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wv][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 27) part[(int64_t)blockIdx.x * 27 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void head_bwd_final_kernel(const double* __restrict__ part, int nb, float* __restrict__ dw, float* __restrict__ db) {
    int k = threadIdx.x;
    if (k >= 27) return;
    double s = 0;
    for (int j = 0; j < nb; ++j) s += part[(int64_t)j * 27 + k];
    if (k < 24) dw[k] = (float)s; else db[k - 24] = (float)s;
}
static unsigned head_blocks(int64_t n) { return nblocks(n, 256, 1024); }
extern "C" size_t corrif_head_workspace(int32_t B, int32_t HW) { return (size_t)head_blocks((int64_t)B * HW) * 27 * sizeof(double); }
extern "C" int corrif_head_fwd(const float* x, const float* w, const float* b, float* pred, int32_t B, int32_t HW, void* stream) {
    if (!x || !w || !b || !pred || B <= 0 || HW <= 0) return CORRIF_EINVAL;
    if (!al16(x)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL(head_fwd_kernel, dim3(nblocks((int64_t)B * HW)), dim3(256), 0, (hipStream_t)stream, x, w, b, pred, (int)B, (int)HW);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_head_bwd(const float* dpred, const float* pred, const float* x, const float* w, float* dx, float* dw, float* db,
                               double* ws, int32_t B, int32_t HW, void* stream) {
    if (!dpred || !pred || !x || !w || !dx || !dw || !db || !ws || B <= 0 || HW <= 0) return CORRIF_EINVAL;
    if (!al16(x) || !al16(dx)) return CORRIF_EUNSUPPORTED;
    unsigned nb = head_blocks((int64_t)B * HW);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(nb), dim3(256), 0, s, dpred, pred, x, w, dx, ws, (int)B, (int)HW);
    CORRIF_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_bwd_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, (int)nb, dw, db);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ BCE-with-logits (mean) on the sigmoided prediction
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ x, const float* __restrict__ t, int64_t n, float inv_n,
                                                  float* __restrict__ dx, double* __restrict__ part) {
    __shared__ double red[4];
    double acc = 0;
    GRID_STRIDE(i, n) {
        float xv = x[i], tv = t[i];
        // aten::binary_cross_entropy_with_logits: (1-t)*x + log1p(exp(-|x|)) + max(-x, 0)
        float l = (1.0f - tv) * xv + log1pf(expf(-fabsf(xv))) + fmaxf(-xv, 0.f);
        acc += (double)l;
        if (dx) dx[i] = (1.0f / (1.0f + expf(-xv)) - tv) * inv_n;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void bce_final_kernel(const double* __restrict__ part, int nb, double inv_n, float* __restrict__ loss) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0;
        for (int k = 0; k < nb; ++k) s += part[k];
        loss[0] = (float)(s * inv_n);
    }
}
extern "C" size_t corrif_bce_workspace(int64_t n) { return (size_t)nblocks(n, 256, 1024) * sizeof(double); }
extern "C" int corrif_bce_logits_mean(const float* pred, const float* target, int64_t n, float* loss, float* dpred, double* ws, void* stream) {
    if (!pred || !target || !loss || !ws || n <= 0) return CORRIF_EINVAL;
    unsigned nb = nblocks(n, 256, 1024);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_kernel, dim3(nb), dim3(256), 0, s, pred, target, n, (float)(1.0 / (double)n), dpred, ws);
    CORRIF_CHECK_LAUNCH();
    hipLaunchKernelGGL(bce_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, (int)nb, 1.0 / (double)n, loss);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ Jaccard2 / Jaccard / F1 (F5_JACCARD2.py)
// partial sums per block, fp32 (exact for 0/1 data: every partial is an integer < 2^24):
//   0: sum y   1: sum yp*y   2: sum (1-yp)*y   3: sum (1-y)*yp   4: sum (1-yp)*(1-y)
__global__ __launch_bounds__(256) void jaccard_partial_kernel(const float* __restrict__ y, const float* __restrict__ yp, int64_t n,
                                                              float* __restrict__ part) {
    __shared__ float red[4][5];
    float a[5] = {0, 0, 0, 0, 0};
    GRID_STRIDE(i, n) {
        float t = y[i], p = yp[i];
        a[0] += t;
        a[1] += p * t;
        a[2] += (1.0f - p) * t;
        a[3] += (1.0f - t) * p;
        a[4] += (1.0f - p) * (1.0f - t);
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float v = a[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 5) part[(int64_t)blockIdx.x * 5 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void jaccard_final_kernel(const float* __restrict__ part, int nb, float eps, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < nb; ++k)
        for (int j = 0; j < 5; ++j) s[j] += (double)part[(int64_t)k * 5 + j];
    float sy = (float)s[0], tp = (float)s[1], fp = (float)s[2], fn = (float)s[3], tpc = (float)s[4];
    // Jaccard (no complement): F5_JACCARD2.py:4-9
    out[1] = (tp + eps) / (tp + fp + fn + eps);
    // Jaccard2 / JaccardAndF1: complement both when the mask is empty (F5_JACCARD2.py:12-14, 23-25):
    //   y' = 1-y, yp' = 1-yp  =>  TP' = sum (1-yp)(1-y), "FP'" = sum yp*(1-y) = fn, "FN'" = sum y*(1-yp) = fp
    float TP = tp, FP = fp, FN = fn;
    if (sy == 0.f) { TP = tpc; FP = fn; FN = fp; }
    out[0] = (TP + eps) / (TP + FP + FN + eps);
    float recall = TP / (TP + FN + eps), prec = TP / (TP + FP + eps);
    out[2] = 2.0f * (recall * prec) / (recall + prec + eps);
}
extern "C" size_t corrif_jaccard_workspace(int64_t n) { return (size_t)nblocks(n, 256, 2048) * 5 * sizeof(float); }
extern "C" int corrif_jaccard(const float* y, const float* y_pred, int64_t n, float eps, float* out, float* ws, void* stream) {
    if (!y || !y_pred || !out || !ws || n <= 0) return CORRIF_EINVAL;
    unsigned nb = nblocks(n, 256, 2048);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(jaccard_partial_kernel, dim3(nb), dim3(256), 0, s, y, y_pred, n, ws);
    CORRIF_CHECK_LAUNCH();
    hipLaunchKernelGGL(jaccard_final_kernel, dim3(1), dim3(64), 0, s, (const float*)ws, (int)nb, eps, out);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam, no amsgrad)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                            float step_size, float beta1, float beta2, float eps, float wd, float bc2_sqrt) {
    GRID_STRIDE(i, n) {
        float gv = g[i];
        if (wd != 0.f) gv += wd * p[i];
        float mv = m[i] + (1.0f - beta1) * (gv - m[i]);        // lerp form used by torch
        float vv = beta2 * v[i] + (1.0f - beta2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] -= step_size * (mv / denom);
    }
}
extern "C" int corrif_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int32_t step, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return CORRIF_EINVAL;
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)((double)lr / bc1), beta1, beta2, eps,
                       weight_decay, (float)sqrt(bc2));
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ tiny-channel 1x1x1 convolution (Ci, Co <= 16)
// d1_out / d2_out (8->8, 16->16 on 128^3 / 64^3 grids, mmvit4.py:233,236) have 2-4 FLOP/B: pure HBM streams.  One thread
// per voxel, weights broadcast from LDS, float4 I/O.  The same kernel is the data gradient (transposed weights).
template <int CI, int CO>
__global__ __launch_bounds__(256) void conv1x1_small_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                            int w_transposed, const float* __restrict__ bias, float* __restrict__ y,
                                                            int64_t ldy, int64_t rows) {
    __shared__ float ws[CO * CI + CO];
    for (int i = threadIdx.x; i < CO * CI; i += 256) {
        int co = i / CI, ci = i - co * CI;
        ws[i] = w_transposed ? w[ci * CO + co] : w[i];          // ws[co][ci]
    }
    for (int i = threadIdx.x; i < CO; i += 256) ws[CO * CI + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    GRID_STRIDE(r, rows) {
        float xv[CI];
#pragma unroll
        for (int c = 0; c < CI; c += 4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
            xv[c] = v[0]; xv[c + 1] = v[1]; xv[c + 2] = v[2]; xv[c + 3] = v[3];
        }
#pragma unroll
        for (int o = 0; o < CO; o += 4) {
            f32x4 acc;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = ws[CO * CI + o + e];
#pragma unroll
                for (int c = 0; c < CI; ++c) a = fmaf(xv[c], ws[(o + e) * CI + c], a);
                acc[e] = a;
            }
            *reinterpret_cast<f32x4*>(y + r * ldy + o) = acc;
        }
    }
}
// dW[co][ci] = sum_r dy[r][co] x[r][ci], db[co] = sum_r dy[r][co]: per-thread register tile, wave butterfly, block partials (double)
template <int CI, int CO>   // blockIdx.y selects a slice of 8 output channels (keeps the register tile at 8*CI + 8 floats)
__global__ __launch_bounds__(256) void conv1x1_small_wgrad_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                                  int64_t lddy, int64_t rows, double* __restrict__ part) {
    constexpr int OT = 8, NA = OT * CI + OT, NT = CO * CI + CO;
    __shared__ float red[4][NA];
    const int o0 = blockIdx.y * OT;
    float acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.f;
    int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        float xv[CI], gv[OT];
#pragma unroll
        for (int c = 0; c < CI; c += 4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ldx + c);
            xv[c] = v[0]; xv[c + 1] = v[1]; xv[c + 2] = v[2]; xv[c + 3] = v[3];
        }
#pragma unroll
        for (int o = 0; o < OT; o += 4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(dy + r * lddy + o0 + o);
            gv[o] = v[0]; gv[o + 1] = v[1]; gv[o + 2] = v[2]; gv[o + 3] = v[3];
        }
#pragma unroll
        for (int o = 0; o < OT; ++o) {
#pragma unroll
            for (int c = 0; c < CI; ++c) acc[o * CI + c] = fmaf(gv[o], xv[c], acc[o * CI + c]);
            acc[OT * CI + o] += gv[o];
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[wv][i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NA; i += 256) {
        const double s = (double)red[0][i] + (double)red[1][i] + (double)red[2][i] + (double)red[3][i];
        const int dst = i < OT * CI ? (o0 * CI + i) : (CO * CI + o0 + (i - OT * CI));     // dW[o0*CI + ...] | db[o0 + ...]
        part[(int64_t)blockIdx.x * NT + dst] = s;
    }
}
__global__ void conv1x1_small_wgrad_final_kernel(const double* __restrict__ part, int nb, int n_w, int n_b, float* __restrict__ dw,
                                                 float* __restrict__ db) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_w + n_b) return;
    double s = 0;
    for (int k = 0; k < nb; ++k) s += part[(int64_t)k * (n_w + n_b) + i];
    if (i < n_w) dw[i] = (float)s;
    else if (db) db[i - n_w] = (float)s;
}
static int small1x1_blocks(int64_t rows) { int64_t b = (rows + 2047) / 2048; return (int)(b > 512 ? 512 : (b < 1 ? 1 : b)); }
extern "C" int corrif_conv1x1_small_supported(int32_t Ci, int32_t Co) { return (Ci == Co) && (Ci == 8 || Ci == 16); }
extern "C" size_t corrif_conv1x1_small_workspace(int64_t rows, int32_t Ci, int32_t Co) {
    return (size_t)small1x1_blocks(rows) * (Ci * Co + Co) * sizeof(double);
}
extern "C" int corrif_conv1x1_small_fwd(const float* x, int64_t ldx, const float* w, int32_t w_transposed, const float* bias, float* y,
                                        int64_t ldy, int64_t rows, int32_t Ci, int32_t Co, void* stream) {
    if (!x || !w || !y || rows <= 0) return CORRIF_EINVAL;
    if (!corrif_conv1x1_small_supported(Ci, Co) || (ldx & 3) || (ldy & 3) || !al16(x) || !al16(y)) return CORRIF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    unsigned nb = nblocks(rows, 256, 16384);
    if (Ci == 8) hipLaunchKernelGGL((conv1x1_small_kernel<8, 8>), dim3(nb), dim3(256), 0, s, x, ldx, w, (int)w_transposed, bias, y, ldy, rows);
    else hipLaunchKernelGGL((conv1x1_small_kernel<16, 16>), dim3(nb), dim3(256), 0, s, x, ldx, w, (int)w_transposed, bias, y, ldy, rows);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_conv1x1_small_wgrad(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dw, float* db, double* ws,
                                          int64_t rows, int32_t Ci, int32_t Co, void* stream) {
    if (!x || !dy || !dw || !ws || rows <= 0) return CORRIF_EINVAL;
    if (!corrif_conv1x1_small_supported(Ci, Co) || (ldx & 3) || (lddy & 3) || !al16(x) || !al16(dy)) return CORRIF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    int nb = small1x1_blocks(rows);
    if (Ci == 8) hipLaunchKernelGGL((conv1x1_small_wgrad_kernel<8, 8>), dim3(nb, 1), dim3(256), 0, s, x, ldx, dy, lddy, rows, ws);
    else hipLaunchKernelGGL((conv1x1_small_wgrad_kernel<16, 16>), dim3(nb, 2), dim3(256), 0, s, x, ldx, dy, lddy, rows, ws);
    CORRIF_CHECK_LAUNCH();
    const int n = Ci * Co + Co;
    hipLaunchKernelGGL(conv1x1_small_wgrad_final_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const double*)ws, nb, Ci * Co, (int)Co, dw, db);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ multi-tensor Adam: one launch for all parameters
// table entry per tensor: {p, g, m, v, n}; block b works on chunk blk_off[b] (1024 elements) of tensor blk_tensor[b]
struct AdamEntry { float* p; const float* g; float* m; float* v; int64_t n; };
__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamEntry* __restrict__ tab, const int* __restrict__ blk_tensor,
                                                         const int64_t* __restrict__ blk_off, float step_size, float beta1, float beta2,
                                                         float eps, float wd, float bc2_sqrt) {
    const AdamEntry e = tab[blk_tensor[blockIdx.x]];
    const int64_t base = blk_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i >= e.n) break;
        float gv = e.g[i];
        if (wd != 0.f) gv += wd * e.p[i];
        const float mv = e.m[i] + (1.0f - beta1) * (gv - e.m[i]);
        const float vv = beta2 * e.v[i] + (1.0f - beta2) * gv * gv;
        e.m[i] = mv;
        e.v[i] = vv;
        e.p[i] -= step_size * (mv / (sqrtf(vv) / bc2_sqrt + eps));
    }
}
extern "C" int corrif_adam_multi(const void* table, const int32_t* blk_tensor, const int64_t* blk_off, int32_t nblocks_, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int32_t step, void* stream) {
    if (!table || !blk_tensor || !blk_off || nblocks_ <= 0 || step < 1) return CORRIF_EINVAL;
    double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nblocks_), dim3(256), 0, (hipStream_t)stream, (const AdamEntry*)table, (const int*)blk_tensor,
                       blk_off, (float)((double)lr / bc1), beta1, beta2, eps, weight_decay, (float)sqrt(bc2));
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ multi-tensor gather: the gradients of one all-reduce bucket in ONE launch
// table entry per tensor: {src (NULL: the segment is zero-filled), dst offset in the flat bucket, n}; block b copies chunk blk_off[b] (1024
// elements) of tensor blk_tensor[b].  Replaces autograd's per-parameter in-place accumulation into bucket views (one stock ATen add per
// parameter and step) and the zero fill of the buckets: the producers write ordinary gradient tensors, the bucket is assembled once.
struct GatherEntry { const float* src; int64_t dst_off; int64_t n; };
__global__ __launch_bounds__(256) void gather_multi_kernel(const GatherEntry* __restrict__ tab, const int* __restrict__ blk_tensor,
                                                           const int64_t* __restrict__ blk_off, float* __restrict__ dst) {
    const GatherEntry e = tab[blk_tensor[blockIdx.x]];
    const int64_t base = blk_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        if (i >= e.n) break;
        dst[e.dst_off + i] = e.src ? e.src[i] : 0.f;
    }
}
extern "C" int corrif_gather_multi(const void* table, const int32_t* blk_tensor, const int64_t* blk_off, int32_t nblocks_, float* dst, void* stream) {
    if (!table || !blk_tensor || !blk_off || !dst || nblocks_ <= 0) return CORRIF_EINVAL;
    hipLaunchKernelGGL(gather_multi_kernel, dim3(nblocks_), dim3(256), 0, (hipStream_t)stream, (const GatherEntry*)table, (const int*)blk_tensor,
                       blk_off, dst);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------ input pipeline (F8_IMAGES4.py:36-88), SURVEY 8(f) N3
// raw patches are pixel-interleaved (HWC): rgb [N, HW, 3], all20 [N, HW, 20]; the network wants planar [N, 3 modalities, 3 bands, HW]
// with the per-band TRAINING-SET mean removed: modality 0 = R,G,B; 1 = bands 9,10,11; 2 = bands 12,13,14 of the 20-band cube.
__device__ __forceinline__ void prep_pixel(const float* __restrict__ rgb, const float* __restrict__ all20, int64_t px, float (&v)[9]) {
    v[0] = rgb[px * 3]; v[1] = rgb[px * 3 + 1]; v[2] = rgb[px * 3 + 2];
#pragma unroll
    for (int c = 0; c < 6; ++c) v[3 + c] = all20[px * 20 + 9 + c];
}
__global__ __launch_bounds__(256) void prep_means_partial_kernel(const float* __restrict__ rgb, const float* __restrict__ all20,
                                                                 const int* __restrict__ trind, int ntr, int HW, double* __restrict__ part) {
    __shared__ double red[4][9];
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int64_t total = (int64_t)ntr * HW;
    GRID_STRIDE(i, total) {
        const int s = (int)(i / HW);
        const int64_t px = (int64_t)trind[s] * HW + (i - (int64_t)s * HW);
        float v[9];
        prep_pixel(rgb, all20, px, v);
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[c] += (double)v[c];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
        double x = acc[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = x;
    }
    __syncthreads();
    if (threadIdx.x < 9) part[(int64_t)blockIdx.x * 9 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
__global__ void prep_means_final_kernel(const double* __restrict__ part, int nb, double inv_n, float* __restrict__ means) {
    int c = threadIdx.x;
    if (c >= 9) return;
    double s = 0;
    for (int k = 0; k < nb; ++k) s += part[(int64_t)k * 9 + c];
    means[c] = (float)(s * inv_n);
}
__global__ void prep_stack_kernel(const float* __restrict__ rgb, const float* __restrict__ all20, const float* __restrict__ masks,
                                  const float* __restrict__ means, float* __restrict__ images, float* __restrict__ targets, int N, int HW) {
    const int64_t total = (int64_t)N * HW;
    float mu[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) mu[c] = means[c];
    GRID_STRIDE(i, total) {
        const int64_t n = i / HW, p = i - n * HW;
        float v[9];
        prep_pixel(rgb, all20, i, v);
#pragma unroll
        for (int c = 0; c < 9; ++c) images[(n * 9 + c) * HW + p] = v[c] - mu[c];       // [n][modality][band][pixel]
        if (masks) {
            const float m = masks[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) targets[(n * 3 + c) * HW + p] = m;              // mask repeated over 3 channels (:88)
        }
    }
}
extern "C" size_t corrif_prep_workspace(int32_t ntr, int32_t HW) { return (size_t)nblocks((int64_t)ntr * HW, 256, 1024) * 9 * sizeof(double); }
extern "C" int corrif_prep_means(const float* rgb, const float* all20, const int32_t* trind, int32_t ntr, int32_t HW, float* means, double* ws,
                                 void* stream) {
    if (!rgb || !all20 || !trind || !means || !ws || ntr <= 0 || HW <= 0) return CORRIF_EINVAL;
    const unsigned nb = nblocks((int64_t)ntr * HW, 256, 1024);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(prep_means_partial_kernel, dim3(nb), dim3(256), 0, s, rgb, all20, (const int*)trind, (int)ntr, (int)HW, ws);
    CORRIF_CHECK_LAUNCH();
    hipLaunchKernelGGL(prep_means_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, (int)nb, 1.0 / ((double)ntr * HW), means);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_prep_stack(const float* rgb, const float* all20, const float* masks, const float* means, float* images, float* targets,
                                 int32_t N, int32_t HW, void* stream) {
    if (!rgb || !all20 || !means || !images || N <= 0 || HW <= 0 || (masks && !targets)) return CORRIF_EINVAL;
    hipLaunchKernelGGL(prep_stack_kernel, dim3(nblocks((int64_t)N * HW)), dim3(256), 0, (hipStream_t)stream, rgb, all20, masks, means, images,
                       targets, (int)N, (int)HW);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
