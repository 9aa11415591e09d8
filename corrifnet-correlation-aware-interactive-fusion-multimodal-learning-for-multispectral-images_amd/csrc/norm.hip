// BatchNorm3d / InstanceNorm3d (channels-last, statistics over rows) and LayerNorm, forward + backward.
// HBM-bound: every pass streams float4 per lane; per-channel partial sums are kept in double so that the
// one-pass E[x^2]-E[x]^2 variance loses nothing to cancellation; partials are combined in a fixed order
// (deterministic, no atomics).
#include "common.h"

struct NormGeo {
    int C4;          // float4 chunks per row
    int CT;          // chunks per block tile  (min(C4,256))
    int RL;          // row lanes per block    (256/CT)
    int ctiles;      // ceil(C4/CT)
    int chunks;      // row chunks per group
    int64_t rows_per_chunk;
};
static NormGeo norm_geo(int64_t rows_per_group, int G, int C) {
    NormGeo n;
    n.C4 = C / 4;
    n.CT = n.C4 < 256 ? n.C4 : 256;
    n.RL = 256 / n.CT;
    n.ctiles = (n.C4 + n.CT - 1) / n.CT;
    int64_t want = 2048 / ((int64_t)G * n.ctiles);
    if (want < 1) want = 1;
    int64_t maxc = (rows_per_group + 4 * n.RL - 1) / (4 * n.RL);   // >= 4 rows per row lane
    if (maxc < 1) maxc = 1;
    n.chunks = (int)(want < maxc ? want : maxc);
    n.rows_per_chunk = (rows_per_group + n.chunks - 1) / n.chunks;
    return n;
}
extern "C" size_t corrif_norm_workspace(int64_t rows_per_group, int32_t G, int32_t C) {
    NormGeo n = norm_geo(rows_per_group, G, C);
    return (size_t)G * n.chunks * C * 2 * sizeof(double) + (size_t)G * C * 2 * sizeof(float) + 64;
}
// The grouped (`_g`) entries chunk EVERY group exactly like a standalone launch with G = 1 would (norm_geo(rows, 1, C)): the partial sums
// and their fixed summation order - hence mean, rstd and every gradient - are bit-identical to running the G norms one by one.
extern "C" size_t corrif_norm_workspace_g(int64_t rows_per_group, int32_t G, int32_t C) {
    NormGeo n = norm_geo(rows_per_group, 1, C);
    return (size_t)G * n.chunks * C * 2 * sizeof(double) + (size_t)G * C * 2 * sizeof(float) + 64;
}

// partial[((g*C + c)*chunks + chunk)*2 + {0,1}]
template <int MODE>   // 0: sum x', sum x'^2 ; 1: sum g, sum g*xhat
__global__ __launch_bounds__(256) void norm_partial_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ dy,
                                                           int64_t lddy, const float* __restrict__ y, int64_t ldy,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           double* __restrict__ part, int64_t rows_per_group, int C, int flags,
                                                           NormGeo n) {
    __shared__ double red[256 * 8];
    const int tid = threadIdx.x;
    const int cl = tid % n.CT, rl = tid / n.CT;
    const int chunk = blockIdx.x, ct = blockIdx.y, g = blockIdx.z;
    const int c4 = ct * n.CT + cl;
    const bool active = (rl < n.RL) && (c4 < n.C4);
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
    if (active) {
        const int64_t r0 = (int64_t)chunk * n.rows_per_chunk;
        int64_t r1 = r0 + n.rows_per_chunk;
        if (r1 > rows_per_group) r1 = rows_per_group;
        const int64_t gbase = (int64_t)g * rows_per_group;
        f32x4 mu = {0, 0, 0, 0}, rs = {1, 1, 1, 1};
        if (MODE == 1) {
            mu = *reinterpret_cast<const f32x4*>(mean + (int64_t)g * C + c4 * 4);
            rs = *reinterpret_cast<const f32x4*>(rstd + (int64_t)g * C + c4 * 4);
        }
        for (int64_t r = r0 + rl; r < r1; r += n.RL) {
            const int64_t row = gbase + r;
            f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ldx + c4 * 4);
            if (flags & CORRIF_NORM_RELU_IN) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s0[e] += (double)v[e]; s1[e] += (double)v[e] * (double)v[e]; }
            } else {
                f32x4 gv = *reinterpret_cast<const f32x4*>(dy + row * lddy + c4 * 4);
                if (flags & CORRIF_NORM_RELU_OUT) {
                    f32x4 yv = *reinterpret_cast<const f32x4*>(y + row * ldy + c4 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) gv[e] = yv[e] > 0.f ? gv[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xh = (v[e] - mu[e]) * rs[e];
                    s0[e] += (double)gv[e];
                    s1[e] += (double)gv[e] * (double)xh;
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = s0[e]; red[tid * 8 + 4 + e] = s1[e]; }
    __syncthreads();
    if (rl == 0 && c4 < n.C4) {
        for (int k = 1; k < n.RL; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s0[e] += red[(k * n.CT + cl) * 8 + e]; s1[e] += red[(k * n.CT + cl) * 8 + 4 + e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            double* o = part + (((int64_t)g * C + c4 * 4 + e) * n.chunks + chunk) * 2;
            o[0] = s0[e];
            o[1] = s1[e];
        }
    }
}

// one wave per (group, channel): lanes stride over the chunk partials, fixed-order butterfly in double
__device__ __forceinline__ void wave_sum2(double& a, double& b) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
}
struct RunPtrs { float* mean[4]; float* var[4]; };     // running statistics of up to 4 groups (grouped BatchNorm: one nn.BatchNorm3d per group)
__global__ __launch_bounds__(256) void norm_stats_final_kernel(const double* __restrict__ part, int chunks, int G, int C,
                                                               int64_t rows_per_group, float eps, float* __restrict__ mean,
                                                               float* __restrict__ rstd, RunPtrs run, float momentum) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= G * C) return;
    const int c = i % C;
    double s = 0, q = 0;
    const double* o = part + (int64_t)i * chunks * 2;
    for (int k = lane; k < chunks; k += 64) { s += o[k * 2]; q += o[k * 2 + 1]; }
    wave_sum2(s, q);
    if (lane != 0) return;
    double m = s / (double)rows_per_group;
    double var = q / (double)rows_per_group - m * m;
    if (var < 0) var = 0;
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    if (run.mean[0]) {   // G <= 4 ; momentum update with the unbiased variance (nn.BatchNorm3d training)
        const int gi = i / C;              // explicit selects: a dynamic index into the by-value struct would go through scratch
        float* const running_mean = gi == 0 ? run.mean[0] : gi == 1 ? run.mean[1] : gi == 2 ? run.mean[2] : run.mean[3];
        float* const running_var = gi == 0 ? run.var[0] : gi == 1 ? run.var[1] : gi == 2 ? run.var[2] : run.var[3];
        double unb = rows_per_group > 1 ? var * (double)rows_per_group / (double)(rows_per_group - 1) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
}

// sums[(g*C + c)*2 + {0,1}] = (sum g, sum g*xhat) ; optional dgamma/dbeta ; colsum (bias gradient)
__global__ __launch_bounds__(256) void norm_bwd_final_kernel(const double* __restrict__ part, int chunks, int G, int C,
                                                             float* __restrict__ sums, float* dgamma, float* dbeta, float* colsum,
                                                             int per_group) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= G * C) return;
    const int c = i % C;
    double s = 0, q = 0;
    const double* o = part + (int64_t)i * chunks * 2;
    for (int k = lane; k < chunks; k += 64) { s += o[k * 2]; q += o[k * 2 + 1]; }
    wave_sum2(s, q);
    if (lane != 0) return;
    if (sums) { sums[i * 2] = (float)s; sums[i * 2 + 1] = (float)q; }
    const int oc = per_group ? i : c;        // per-group affine parameters / column sums: [G][C]
    if (dgamma) dgamma[oc] = (float)q;
    if (dbeta) dbeta[oc] = (float)s;
    if (colsum) colsum[oc] = (float)s;
}

__global__ void eval_rstd_kernel(const float* __restrict__ var, float eps, float* __restrict__ rstd, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) rstd[i] = 1.0f / sqrtf(var[i] + eps);
}


// element-wise passes: block = (row block, channel tile, group); thread (cl, rl) owns chunk c4 and rows rl, rl+RL, ...
struct ApplyGeo { int CT, RL, C4; int64_t rows_per_block; };
static ApplyGeo apply_geo(int64_t rows_per_group, int C, int& nblk, int& ctiles) {
    ApplyGeo a;
    a.C4 = C / 4;
    a.CT = a.C4 < 256 ? a.C4 : 256;
    a.RL = 256 / a.CT;
    ctiles = (a.C4 + a.CT - 1) / a.CT;
    int64_t per = (int64_t)a.RL * 8;
    int64_t nb = (rows_per_group + per - 1) / per;
    if (nb > 16384) nb = 16384;
    if (nb < 1) nb = 1;
    a.rows_per_block = (rows_per_group + nb - 1) / nb;
    nblk = (int)((rows_per_group + a.rows_per_block - 1) / a.rows_per_block);
    return a;
}

__global__ __launch_bounds__(256) void norm_apply_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ res, int64_t ldr,
                                                         float* __restrict__ y, int64_t ldy, int64_t rows_per_group, int C, int flags,
                                                         ApplyGeo a, int64_t ags) {
    const int tid = threadIdx.x, cl = tid % a.CT, rl = tid / a.CT;
    const int c4 = blockIdx.y * a.CT + cl, g = blockIdx.z;
    if (rl >= a.RL || c4 >= a.C4) return;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
    int64_t r1 = r0 + a.rows_per_block;
    if (r1 > rows_per_group) r1 = rows_per_group;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + (int64_t)g * C + c4 * 4);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + (int64_t)g * C + c4 * 4);
    f32x4 ga = {1, 1, 1, 1}, be = {0, 0, 0, 0};
    if (gamma) ga = *reinterpret_cast<const f32x4*>(gamma + g * ags + c4 * 4);
    if (beta) be = *reinterpret_cast<const f32x4*>(beta + g * ags + c4 * 4);
    for (int64_t r = r0 + rl; r < r1; r += a.RL) {
        const int64_t row = (int64_t)g * rows_per_group + r;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ldx + c4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float xv = (flags & CORRIF_NORM_RELU_IN) ? fmaxf(v[e], 0.f) : v[e];
            o[e] = (xv - mu[e]) * rs[e] * ga[e] + be[e];
        }
        if (res) {
            f32x4 rv = *reinterpret_cast<const f32x4*>(res + row * ldr + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += rv[e];
        }
        if (flags & CORRIF_NORM_RELU_OUT) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(y + row * ldy + c4 * 4) = o;
    }
}

__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const float* __restrict__ dy, int64_t lddy, const float* __restrict__ y,
                                                             int64_t ldy, const float* __restrict__ x, int64_t ldx,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ sums,
                                                             float* __restrict__ dx, int64_t lddx, float* __restrict__ dres,
                                                             int64_t lddres, int64_t rows_per_group, int C, int flags, int frozen,
                                                             ApplyGeo a, int64_t ags) {
    const int tid = threadIdx.x, cl = tid % a.CT, rl = tid / a.CT;
    const int c4 = blockIdx.y * a.CT + cl, g = blockIdx.z;
    if (rl >= a.RL || c4 >= a.C4) return;
    const int64_t r0 = (int64_t)blockIdx.x * a.rows_per_block;
    int64_t r1 = r0 + a.rows_per_block;
    if (r1 > rows_per_group) r1 = rows_per_group;
    const float invn = 1.0f / (float)rows_per_group;
    f32x4 mu = {0, 0, 0, 0}, rs = {1, 1, 1, 1}, ga = {1, 1, 1, 1}, sg = {0, 0, 0, 0}, sgx = {0, 0, 0, 0};
    if (dx) {
        mu = *reinterpret_cast<const f32x4*>(mean + (int64_t)g * C + c4 * 4);
        rs = *reinterpret_cast<const f32x4*>(rstd + (int64_t)g * C + c4 * 4);
        if (gamma) ga = *reinterpret_cast<const f32x4*>(gamma + g * ags + c4 * 4);
        if (!frozen) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sg[e] = sums[((int64_t)g * C + c4 * 4 + e) * 2] * invn;
                sgx[e] = sums[((int64_t)g * C + c4 * 4 + e) * 2 + 1] * invn;
            }
        }
    }
    for (int64_t r = r0 + rl; r < r1; r += a.RL) {
        const int64_t row = (int64_t)g * rows_per_group + r;
        f32x4 gv = *reinterpret_cast<const f32x4*>(dy + row * lddy + c4 * 4);
        if (flags & CORRIF_NORM_RELU_OUT) {
            f32x4 yv = *reinterpret_cast<const f32x4*>(y + row * ldy + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) gv[e] = yv[e] > 0.f ? gv[e] : 0.f;
        }
        if (dres) *reinterpret_cast<f32x4*>(dres + row * lddres + c4 * 4) = gv;
        if (dx) {
            f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ldx + c4 * 4);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xv = (flags & CORRIF_NORM_RELU_IN) ? fmaxf(v[e], 0.f) : v[e];
                float t = gv[e];
                if (!frozen) t = gv[e] - sg[e] - (xv - mu[e]) * rs[e] * sgx[e];
                float d = ga[e] * rs[e] * t;
                if ((flags & CORRIF_NORM_RELU_IN) && !(v[e] > 0.f)) d = 0.f;
                o[e] = d;
            }
            *reinterpret_cast<f32x4*>(dx + row * lddx + c4 * 4) = o;
        }
    }
}

static bool norm_args_ok(int64_t rows_per_group, int G, int C) {
    if (rows_per_group <= 0 || G <= 0 || G > 65535 || C <= 0 || (C & 3)) return false;
    if (rows_per_group * G >= ((int64_t)1 << 31)) return false;
    return true;
}
static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

static int run_ptrs(RunPtrs& r, float* const* rm, float* const* rv, int G) {
    for (int i = 0; i < 4; ++i) { r.mean[i] = nullptr; r.var[i] = nullptr; }
    if (!rm) return CORRIF_OK;
    if (!rv || G > 4) return CORRIF_EINVAL;
    for (int i = 0; i < G; ++i) {
        if (!rm[i] || !rv[i]) return CORRIF_EINVAL;
        r.mean[i] = rm[i];
        r.var[i] = rv[i];
    }
    return CORRIF_OK;
}

static int norm_stats_core(const float* x, int64_t ldx, int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, float eps,
                           float* mean, float* rstd, float* const* running_means, float* const* running_vars, float momentum,
                           double* ws, void* stream, int geoG) {
    if (!x || !mean || !rstd || !ws || !norm_args_ok(rows_per_group, G, C)) return CORRIF_EINVAL;
    if ((ldx & 3) || !al16(x)) return CORRIF_EUNSUPPORTED;
    RunPtrs run;
    if (run_ptrs(run, running_means, running_vars, G) != CORRIF_OK) return CORRIF_EINVAL;
    NormGeo n = norm_geo(rows_per_group, geoG, C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((norm_partial_kernel<0>), dim3(n.chunks, n.ctiles, G), dim3(256), 0, s, x, ldx, (const float*)nullptr, (int64_t)0,
                       (const float*)nullptr, (int64_t)0, (const float*)nullptr, (const float*)nullptr, ws, rows_per_group, (int)C,
                       (int)flags, n);
    CORRIF_CHECK_LAUNCH();
    hipLaunchKernelGGL(norm_stats_final_kernel, dim3((G * C + 3) / 4), dim3(256), 0, s, (const double*)ws, n.chunks, (int)G, (int)C,
                       rows_per_group, eps, mean, rstd, run, momentum);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_norm_stats_g(const float* x, int64_t ldx, int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, float eps,
                                   float* mean, float* rstd, float* const* running_means, float* const* running_vars, float momentum,
                                   double* ws, void* stream) {
    return norm_stats_core(x, ldx, rows_per_group, G, C, flags, eps, mean, rstd, running_means, running_vars, momentum, ws, stream, 1);
}
extern "C" int corrif_norm_stats(const float* x, int64_t ldx, int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, float eps,
                                 float* mean, float* rstd, float* running_mean, float* running_var, float momentum, double* ws,
                                 void* stream) {
    if (running_mean && (G != 1 || !running_var)) return CORRIF_EINVAL;
    return norm_stats_core(x, ldx, rows_per_group, G, C, flags, eps, mean, rstd, running_mean ? &running_mean : nullptr,
                           running_mean ? &running_var : nullptr, momentum, ws, stream, G);
}

extern "C" int corrif_norm_stats_finalize_g(const double* part, int32_t chunks, int32_t G, int32_t C, int64_t rows_per_group, float eps,
                                            float* mean, float* rstd, float* const* running_means, float* const* running_vars,
                                            float momentum, void* stream) {
    if (!part || !mean || !rstd || chunks <= 0 || !norm_args_ok(rows_per_group, G, C)) return CORRIF_EINVAL;
    RunPtrs run;
    if (run_ptrs(run, running_means, running_vars, G) != CORRIF_OK) return CORRIF_EINVAL;
    hipLaunchKernelGGL(norm_stats_final_kernel, dim3((G * C + 3) / 4), dim3(256), 0, (hipStream_t)stream, part, (int)chunks, (int)G, (int)C,
                       rows_per_group, eps, mean, rstd, run, momentum);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_norm_stats_finalize(const double* part, int32_t chunks, int32_t G, int32_t C, int64_t rows_per_group, float eps, float* mean,
                                          float* rstd, float* running_mean, float* running_var, float momentum, void* stream) {
    if (running_mean && (G != 1 || !running_var)) return CORRIF_EINVAL;
    return corrif_norm_stats_finalize_g(part, chunks, G, C, rows_per_group, eps, mean, rstd, running_mean ? &running_mean : nullptr,
                                        running_mean ? &running_var : nullptr, momentum, stream);
}

extern "C" int corrif_norm_eval_rstd(const float* running_var, float eps, float* rstd, int32_t C, void* stream) {
    if (!running_var || !rstd || C <= 0) return CORRIF_EINVAL;
    hipLaunchKernelGGL(eval_rstd_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, running_var, eps, rstd, (int)C);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

extern "C" int corrif_norm_apply_g(const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, const float* residual, int64_t ldr, float* y, int64_t ldy,
                                   int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, int64_t affine_gstride, void* stream) {
    if (!x || !mean || !rstd || !y || !norm_args_ok(rows_per_group, G, C)) return CORRIF_EINVAL;
    if ((ldx & 3) || (ldy & 3) || (ldr & 3) || !al16(x) || !al16(y) || !al16(residual) || !al16(mean) || !al16(rstd) ||
        !al16(gamma) || !al16(beta) || (affine_gstride & 3))
        return CORRIF_EUNSUPPORTED;
    int nblk, ctiles;
    ApplyGeo a = apply_geo(rows_per_group, C, nblk, ctiles);
    hipLaunchKernelGGL(norm_apply_kernel, dim3(nblk, ctiles, G), dim3(256), 0, (hipStream_t)stream, x, ldx, mean, rstd, gamma, beta,
                       residual, ldr, y, ldy, rows_per_group, (int)C, (int)flags, a, affine_gstride);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_norm_apply(const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma,
                                 const float* beta, const float* residual, int64_t ldr, float* y, int64_t ldy,
                                 int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, void* stream) {
    return corrif_norm_apply_g(x, ldx, mean, rstd, gamma, beta, residual, ldr, y, ldy, rows_per_group, G, C, flags, 0, stream);
}

static int norm_bwd_core(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                         const float* mean, const float* rstd, const float* gamma, float* dx, int64_t lddx, float* dres,
                         int64_t lddres, float* dgamma, float* dbeta, int64_t rows_per_group, int32_t G, int32_t C,
                         int32_t flags, int32_t frozen, int64_t affine_gstride, double* ws, void* stream, int geoG) {
    if (!dy || !norm_args_ok(rows_per_group, G, C) || !ws) return CORRIF_EINVAL;
    if ((flags & CORRIF_NORM_RELU_OUT) && !y) return CORRIF_EINVAL;
    if (dx && (!x || !mean || !rstd)) return CORRIF_EINVAL;
    if ((dgamma || dbeta) && ((G != 1 && affine_gstride != C) || !x || !mean || !rstd)) return CORRIF_EINVAL;
    if ((lddy & 3) || (ldy & 3) || (ldx & 3) || (lddx & 3) || (lddres & 3) || !al16(dy) || !al16(y) || !al16(x) || !al16(dx) ||
        !al16(dres) || !al16(mean) || !al16(rstd) || !al16(gamma) || (affine_gstride & 3))
        return CORRIF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    NormGeo n = norm_geo(rows_per_group, geoG, C);
    // sums live behind the double partials in the same workspace
    float* sums = reinterpret_cast<float*>(ws + (size_t)G * n.chunks * C * 2);
    const bool need_sums = (!frozen && dx) || dgamma || dbeta;
    if (need_sums) {
        hipLaunchKernelGGL((norm_partial_kernel<1>), dim3(n.chunks, n.ctiles, G), dim3(256), 0, s, x, ldx, dy, lddy, y, ldy, mean, rstd, ws,
                           rows_per_group, (int)C, (int)flags, n);
        CORRIF_CHECK_LAUNCH();
        hipLaunchKernelGGL(norm_bwd_final_kernel, dim3((G * C + 3) / 4), dim3(256), 0, s, (const double*)ws, n.chunks, (int)G, (int)C,
                           sums, dgamma, dbeta, (float*)nullptr, affine_gstride ? 1 : 0);
        CORRIF_CHECK_LAUNCH();
    }
    if (dx || dres) {
        int nblk, ctiles;
        ApplyGeo a = apply_geo(rows_per_group, C, nblk, ctiles);
        hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(nblk, ctiles, G), dim3(256), 0, s, dy, lddy, y, ldy, x, ldx, mean, rstd, gamma,
                           (const float*)sums, dx, lddx, dres, lddres, rows_per_group, (int)C, (int)flags, (int)frozen, a, affine_gstride);
        CORRIF_CHECK_LAUNCH();
    }
    return CORRIF_OK;
}
extern "C" int corrif_norm_bwd_g(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                                 const float* mean, const float* rstd, const float* gamma, float* dx, int64_t lddx, float* dres,
                                 int64_t lddres, float* dgamma, float* dbeta, int64_t rows_per_group, int32_t G, int32_t C,
                                 int32_t flags, int32_t frozen, int64_t affine_gstride, double* ws, void* stream) {
    return norm_bwd_core(dy, lddy, y, ldy, x, ldx, mean, rstd, gamma, dx, lddx, dres, lddres, dgamma, dbeta, rows_per_group, G, C, flags, frozen,
                         affine_gstride, ws, stream, 1);
}
extern "C" int corrif_norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                               const float* mean, const float* rstd, const float* gamma, float* dx, int64_t lddx, float* dres,
                               int64_t lddres, float* dgamma, float* dbeta, int64_t rows_per_group, int32_t G, int32_t C,
                               int32_t flags, int32_t frozen, double* ws, void* stream) {
    return norm_bwd_core(dy, lddy, y, ldy, x, ldx, mean, rstd, gamma, dx, lddx, dres, lddres, dgamma, dbeta, rows_per_group, G, C, flags,
                         frozen, 0, ws, stream, G);
}

extern "C" int corrif_norm_bwd_pre_g(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx, const float* mean,
                                     const float* rstd, const float* gamma, float* dx, int64_t lddx, float* dres, int64_t lddres, float* dgamma,
                                     float* dbeta, int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, int64_t affine_gstride,
                                     const double* part, int32_t chunks, double* ws, void* stream) {
    if (!dy || !x || !mean || !rstd || !part || chunks <= 0 || !ws || !norm_args_ok(rows_per_group, G, C)) return CORRIF_EINVAL;
    if ((flags & CORRIF_NORM_RELU_OUT) && !y) return CORRIF_EINVAL;
    if (G != 1 && affine_gstride != C) return CORRIF_EINVAL;
    if ((lddy & 3) || (ldy & 3) || (ldx & 3) || (lddx & 3) || (lddres & 3) || !al16(dy) || !al16(y) || !al16(x) || !al16(dx) ||
        !al16(dres) || !al16(mean) || !al16(rstd) || !al16(gamma) || (affine_gstride & 3))
        return CORRIF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    float* sums = reinterpret_cast<float*>(ws);
    hipLaunchKernelGGL(norm_bwd_final_kernel, dim3((G * C + 3) / 4), dim3(256), 0, s, part, (int)chunks, (int)G, (int)C, sums, dgamma, dbeta,
                       (float*)nullptr, affine_gstride ? 1 : 0);
    CORRIF_CHECK_LAUNCH();
    if (dx || dres) {
        int nblk, ctiles;
        ApplyGeo a = apply_geo(rows_per_group, C, nblk, ctiles);
        hipLaunchKernelGGL(norm_bwd_apply_kernel, dim3(nblk, ctiles, G), dim3(256), 0, s, dy, lddy, y, ldy, x, ldx, mean, rstd, gamma,
                           (const float*)sums, dx, lddx, dres, lddres, rows_per_group, (int)C, (int)flags, 0, a, affine_gstride);
        CORRIF_CHECK_LAUNCH();
    }
    return CORRIF_OK;
}
extern "C" int corrif_norm_bwd_pre(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx, const float* mean,
                                   const float* rstd, const float* gamma, float* dx, int64_t lddx, float* dres, int64_t lddres, float* dgamma,
                                   float* dbeta, int64_t rows, int32_t C, int32_t flags, const double* part, int32_t chunks, double* ws,
                                   void* stream) {
    return corrif_norm_bwd_pre_g(dy, lddy, y, ldy, x, ldx, mean, rstd, gamma, dx, lddx, dres, lddres, dgamma, dbeta, rows, 1, C, flags, 0, part,
                                 chunks, ws, stream);
}

// column sums (bias gradients): the statistics partial pass (sum, sum^2) + the wave-per-channel final
extern "C" size_t corrif_col_sum_workspace(int64_t rows, int32_t C) { return corrif_norm_workspace(rows, 1, C); }
/* out[g][c] = sum over the rows of group g (G groups of `rows` consecutive rows each) */
extern "C" int corrif_col_sum_g(const float* x, int64_t ld, int64_t rows, int32_t G, int32_t C, float* out, double* ws, void* stream) {
    if (!x || !out || !ws || !norm_args_ok(rows, G, C)) return CORRIF_EINVAL;
    if ((ld & 3) || !al16(x)) return CORRIF_EUNSUPPORTED;
    NormGeo n = norm_geo(rows, 1, C);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((norm_partial_kernel<0>), dim3(n.chunks, n.ctiles, G), dim3(256), 0, s, x, ld, (const float*)nullptr, (int64_t)0,
                       (const float*)nullptr, (int64_t)0, (const float*)nullptr, (const float*)nullptr, ws, rows, (int)C, 0, n);
    CORRIF_CHECK_LAUNCH();
    hipLaunchKernelGGL(norm_bwd_final_kernel, dim3((G * C + 3) / 4), dim3(256), 0, s, (const double*)ws, n.chunks, (int)G, (int)C, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, out, 1);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_col_sum(const float* x, int64_t ld, int64_t rows, int32_t C, float* out, double* ws, void* stream) {
    return corrif_col_sum_g(x, ld, rows, 1, C, out, ws, stream);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over C (multiple of 256, <= 1024): one wave per row, C/64 values per lane in registers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int NV>   // float4 per lane; C = NV*256
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos, int64_t pos_rows,
                                                            float* __restrict__ xsum, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, float eps) {
    constexpr int C = NV * 256;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        v[k] = *reinterpret_cast<const f32x4*>(x + row * C + (k * 64 + lane) * 4);
        if (pos) {
            f32x4 pv = *reinterpret_cast<const f32x4*>(pos + (row % pos_rows) * C + (k * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[k][e] += pv[e];
            if (xsum) *reinterpret_cast<f32x4*>(xsum + row * C + (k * 64 + lane) * 4) = v[k];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[k][e];
    }
    const float mu = wave_sum(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { float d = v[k][e] - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) * (1.0f / C) + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + (k * 64 + lane) * 4);
        f32x4 be = *reinterpret_cast<const f32x4*>(beta + (k * 64 + lane) * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[k][e] - mu) * rs * ga[e] + be[e];
        *reinterpret_cast<f32x4*>(y + row * C + (k * 64 + lane) * 4) = o;
    }
}

// dx per row + per-block partial (dgamma, dbeta) over a slab of rows
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* __restrict__ dx,
                                                            double* __restrict__ part, int64_t rows, int rows_per_block) {
    constexpr int C = NV * 256;
    __shared__ float red[4][2][C];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x4 ga[NV], ag[NV], ab[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        ga[k] = *reinterpret_cast<const f32x4*>(gamma + (k * 64 + lane) * 4);
        ag[k] = (f32x4){0, 0, 0, 0};
        ab[k] = (f32x4){0, 0, 0, 0};
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (int64_t row = r0 + w; row < r1; row += 4) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 g[NV], xh[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            g[k] = *reinterpret_cast<const f32x4*>(dy + row * C + (k * 64 + lane) * 4);
            f32x4 xv = *reinterpret_cast<const f32x4*>(x + row * C + (k * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[k][e] = (xv[e] - mu) * rs;
                ag[k][e] += g[k][e] * xh[k][e];
                ab[k][e] += g[k][e];
                float gg = g[k][e] * ga[k][e];
                s1 += gg;
                s2 += gg * xh[k][e];
            }
        }
        s1 = wave_sum(s1) * (1.0f / C);
        s2 = wave_sum(s2) * (1.0f / C);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = rs * (g[k][e] * ga[k][e] - s1 - xh[k][e] * s2);
            *reinterpret_cast<f32x4*>(dx + row * C + (k * 64 + lane) * 4) = o;
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            red[w][0][(k * 64 + lane) * 4 + e] = ag[k][e];
            red[w][1][(k * 64 + lane) * 4 + e] = ab[k][e];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double a = 0, b = 0;
        for (int ww = 0; ww < 4; ++ww) { a += red[ww][0][c]; b += red[ww][1][c]; }
        part[((int64_t)blockIdx.x * C + c) * 2] = a;
        part[((int64_t)blockIdx.x * C + c) * 2 + 1] = b;
    }
}
// one wave per channel: the lanes stride over the per-block partials, fixed-order butterfly in double (the serial loop of round 1-2,
// one thread per channel over up to 1024 partials, took 157 us per call - launch-latency class work that sat exposed 8x per step)
__global__ __launch_bounds__(256) void layernorm_bwd_final_kernel(const double* __restrict__ part, int nblk, int C, float* dgamma, float* dbeta) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double a = 0, b = 0;
    for (int k = lane; k < nblk; k += 64) { a += part[((int64_t)k * C + c) * 2]; b += part[((int64_t)k * C + c) * 2 + 1]; }
    wave_sum2(a, b);
    if (lane == 0) {
        dgamma[c] = (float)a;
        dbeta[c] = (float)b;
    }
}
static int ln_blocks(int64_t rows, int& rpb) {
    int64_t nb = (rows + 31) / 32;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    rpb = (int)((rows + nb - 1) / nb);
    return (int)((rows + rpb - 1) / rpb);
}
extern "C" size_t corrif_layernorm_workspace(int64_t rows, int32_t C) {
    int rpb;
    int nb = ln_blocks(rows, rpb);
    return (size_t)nb * C * 2 * sizeof(double);
}
extern "C" int corrif_layernorm_fwd(const float* x, const float* pos, int64_t pos_rows, float* xsum, const float* gamma,
                                    const float* beta, float* y, float* mean, float* rstd, int64_t rows, int32_t C, float eps,
                                    void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0) return CORRIF_EINVAL;
    if (pos && pos_rows <= 0) return CORRIF_EINVAL;
    if (C != 512) return CORRIF_EUNSUPPORTED;
    if (!al16(x) || !al16(pos) || !al16(xsum) || !al16(y) || !al16(gamma) || !al16(beta)) return CORRIF_EUNSUPPORTED;
    hipLaunchKernelGGL((layernorm_fwd_kernel<2>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, pos, pos_rows, xsum,
                       gamma, beta, y, mean, rstd, rows, eps);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
extern "C" int corrif_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx,
                                    double* ws, float* dgamma, float* dbeta, int64_t rows, int32_t C, void* stream) {
    if (!dy || !x || !mean || !rstd || !gamma || !dx || !ws || !dgamma || !dbeta || rows <= 0) return CORRIF_EINVAL;
    if (C != 512) return CORRIF_EUNSUPPORTED;
    if (!al16(dy) || !al16(x) || !al16(dx) || !al16(gamma)) return CORRIF_EUNSUPPORTED;
    int rpb;
    int nb = ln_blocks(rows, rpb);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((layernorm_bwd_kernel<2>), dim3(nb), dim3(256), 0, s, dy, x, mean, rstd, gamma, dx, ws, rows, rpb);
    CORRIF_CHECK_LAUNCH();
    hipLaunchKernelGGL(layernorm_bwd_final_kernel, dim3((C + 3) / 4), dim3(256), 0, s, (const double*)ws, nb, (int)C, dgamma, dbeta);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
