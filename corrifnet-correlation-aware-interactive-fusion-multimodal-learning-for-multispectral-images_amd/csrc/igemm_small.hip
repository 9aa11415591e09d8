// Narrow-channel variants of the implicit GEMM for the decoder's 8/16-channel layers (H5 of SURVEY section 7).
//
// A 32x32 MFMA tile wastes 75 % of the matrix core when Cout = 8.  v_mfma_f32_4x4x1_16b_f32 performs 16
// independent 4x4 outer products per instruction at the same FLOP rate (64 FLOP/clk/SIMD): block b = lane>>2,
// A[b][i] and B[b][j] come from lane 4b+i / 4b+j, and D[b][i][j] lands in VGPR i of lane 4b+j.
//
//   forward / data-gradient (N = Cout <= 16):  A operand = weights (i = 4 output channels, same for all blocks),
//       B operand = activations (j = voxel: lane l of a wave owns voxel l of its 64-row slab).  After the K loop
//       lane l holds out[voxel l][4 channels] per channel group -> one float4 store per lane, no waste.
//   weight gradient (M = Cout <= 16):  A operand = dY[row][4 channels] (broadcast), B operand = gathered
//       X[row][j], j = 64 consecutive (tap, ci) columns across the lanes; K = 1 row per instruction.
#include "igemm_args.h"

// ------------------------------------------------------------------------------------------------ forward-type
template <int NG>   // channel groups of 4 (N <= 4*NG)
__global__ __launch_bounds__(256) void smalln_fwd_kernel(GemmArgs p) {
    constexpr int BM = 256, AI = BM / 32;
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[16 * LDS_PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = tile * BM;
    const int z = blockIdx.z, zo = z / p.Zi, zi = z - zo * p.Zi;
    const float* __restrict__ A = p.A + zo * p.sA_o + zi * p.sA_i;
    const float* __restrict__ B = p.B + zo * p.sB_o + zi * p.sB_i;
    float* __restrict__ C = p.C + zo * p.sC_o + zi * p.sC_i;

    const int kc = tid & 7, ar = tid >> 3;
    int64_t a_base[AI];
    uint32_t a_pack[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        int row = m0 + ar + 32 * i;
        if (row < p.M) {
            if (p.g.is_gemm) {
                a_base[i] = (int64_t)row * p.lda;
                a_pack[i] = 0;
            } else {
                uint32_t n, pk;
                decode_row((uint32_t)row, p.g, n, pk);
                a_base[i] = (int64_t)n * p.g.sample_pitch;
                a_pack[i] = pk;
            }
        } else {
            a_base[i] = 0;
            a_pack[i] = 0xFFFFFFFFu;
        }
    }
    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 ra[AI], rb;
    const int nk = (p.K + BK - 1) / BK;
    auto load_tile = [&](int kt) {
        const int q = kt * 8 + kc, k = q * 4;
        const bool kin = k < p.K;
        int c = k, td = 0, th = 0, tw = 0;
        if (!p.g.is_gemm && kin) {
            int tap = q / p.Cs4;
            c = (q - tap * p.Cs4) * 4;
            td = (int)fdiv((uint32_t)tap, p.g.dKhw);
            int rem = tap - td * (int)p.g.dKhw.d;
            th = (int)fdiv((uint32_t)rem, p.g.dKw);
            tw = rem - th * (int)p.g.dKw.d;
        }
        // unconditional loads (masked lanes read the always-mapped first 16 bytes), zero fill by select: see igemm.hip
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            bool ok = kin && a_pack[i] != 0xFFFFFFFFu;
            int64_t off = a_base[i] + k;
            if (!p.g.is_gemm) {
                int vox;
                ok = gather_voxel(a_pack[i], td, th, tw, p.g, vox) && ok;
                off = a_base[i] + (int64_t)vox * p.lda + c;
            }
            const f32x4 v = *reinterpret_cast<const f32x4*>(A + (ok ? off : 0));
            ra[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        {
            bool ok;
            int64_t off;
            if (p.b_layout == 0) {          // [N][K]: chunk (n = tid>>3, kc)
                const int n = tid >> 3;
                ok = tid < 128 && kin && n < p.N;
                off = (int64_t)n * p.ldb + k;
            } else {                        // [K][N]: chunk (k = tid>>2, n4 = tid&3)
                const int kg = kt * BK + (tid >> 2), n = (tid & 3) * 4;
                ok = tid < 128 && kg < p.K && n < p.N;
                off = (int64_t)kg * p.ldb + n;
            }
            const f32x4 v = *reinterpret_cast<const f32x4*>(B + (ok ? off : 0));
            rb = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < AI; ++i) *reinterpret_cast<f32x4*>(&As[(ar + 32 * i) * LDS_PITCH + kc * 4]) = ra[i];
        if (tid < 128) {
            if (p.b_layout == 0) {
                *reinterpret_cast<f32x4*>(&Bs[(tid >> 3) * LDS_PITCH + kc * 4]) = rb;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[((tid & 3) * 4 + e) * LDS_PITCH + (tid >> 2)] = rb[e];
            }
        }
    };

    const int xrow = wave * 64 + lane, wrow = lane & 3;
    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        store_tile();
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const f32x4 x4 = *reinterpret_cast<const f32x4*>(&As[xrow * LDS_PITCH + kk * 4]);
            f32x4 w4[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) w4[g] = *reinterpret_cast<const f32x4*>(&Bs[(g * 4 + wrow) * LDS_PITCH + kk * 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(w4[g][e], x4[e], acc[g], 0, 0, 0);
        }
        __syncthreads();
    }

    const int row = m0 + xrow;
    if (row < p.M) {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int col = g * 4;
            if (col >= p.N) continue;
            f32x4 v = acc[g];
            if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + zo * p.zs_bias + col);
            if (p.addend) v += *reinterpret_cast<const f32x4*>(p.addend + zo * p.zs_add + (int64_t)row * p.ld_add + col);
            if (p.act == CORRIF_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            } else if (p.act == CORRIF_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            *reinterpret_cast<f32x4*>(C + (int64_t)row * p.ldc + col) = v;
        }
    }
}

int launch_smalln_fwd(const GemmArgs& a, int Z, hipStream_t s) {
    if (a.bias && (((uintptr_t)a.bias & 15) || (a.zs_bias & 3))) return CORRIF_EUNSUPPORTED;
    dim3 grid((unsigned)((a.M + 255) / 256), 1, Z);
    const int ng = (a.N + 3) / 4;
    if (ng == 1) hipLaunchKernelGGL((smalln_fwd_kernel<1>), grid, dim3(256), 0, s, a);
    else if (ng == 2) hipLaunchKernelGGL((smalln_fwd_kernel<2>), grid, dim3(256), 0, s, a);
    else if (ng == 3) hipLaunchKernelGGL((smalln_fwd_kernel<3>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((smalln_fwd_kernel<4>), grid, dim3(256), 0, s, a);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}

// ------------------------------------------------------------------------------------------------ weight-gradient type
template <int NG>   // M <= 4*NG output channels, J tile = 256 columns (4 waves x 64)
__global__ __launch_bounds__(256) void smallm_wgrad_kernel(WgradArgs p) {
    constexpr int BN = 256, PA = 20, PB = BN + 4;
    __shared__ __attribute__((aligned(16))) float As[32 * PA];
    __shared__ __attribute__((aligned(16))) float Bs[32 * PB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * BN;
    int split = 0, z = 0;
    if (p.splits > 1) { z = blockIdx.z / p.splits; split = blockIdx.z - z * p.splits; } else z = blockIdx.z;
    const int zo = z / p.Zi, zi = z - zo * p.Zi;
    const float* __restrict__ A = p.A + zo * p.sA_o + zi * p.sA_i;
    const float* __restrict__ B = p.B + zo * p.sB_o + zi * p.sB_i;
    const int r_begin = split * p.rows_per_split;
    const int r_end = min(p.R, r_begin + p.rows_per_split);

    // B staging: thread owns row-in-tile br = tid>>3 and 8 column chunks (tid&7) + 8*i  -> one row decode per tile
    const int br = tid >> 3, jc0 = tid & 7;
    uint32_t tapk[8];        // packed (td<<20 | th<<10 | tw), 0xFFFFFFFF = column outside N
    int cch[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int j = n0 + (jc0 + 8 * i) * 4;
        tapk[i] = 0xFFFFFFFFu;
        cch[i] = j;
        if (j < p.N) {
            if (p.g.is_gemm) {
                tapk[i] = 0;
            } else {
                int tap = j / p.Cs;
                cch[i] = j - tap * p.Cs;
                int td = (int)fdiv((uint32_t)tap, p.g.dKhw);
                int rem = tap - td * (int)p.g.dKhw.d;
                int th = (int)fdiv((uint32_t)rem, p.g.dKw);
                int tw = rem - th * (int)p.g.dKw.d;
                tapk[i] = ((uint32_t)td << 20) | ((uint32_t)th << 10) | (uint32_t)tw;
            }
        }
    }
    const int arow = tid >> 2, am = (tid & 3) * 4;       // A staging: tid < 128

    f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra, rb[8];
    auto load_tile = [&](int r0) {
        {
            const int row = r0 + arow;
            const bool ok = tid < 128 && row < r_end && am < p.M;
            const f32x4 v = *reinterpret_cast<const f32x4*>(A + (ok ? (int64_t)row * p.lda + am : 0));
            ra = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const int row = r0 + br;
        uint32_t n = 0, pk = 0;
        const bool rin = row < r_end;
        if (!p.g.is_gemm) decode_row((uint32_t)(rin ? row : 0), p.g, n, pk);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bool ok = rin && tapk[i] != 0xFFFFFFFFu;
            int64_t off = (int64_t)row * p.ldb + cch[i];
            if (!p.g.is_gemm) {
                int vox;
                ok = gather_voxel(pk, (int)(tapk[i] >> 20), (int)((tapk[i] >> 10) & 1023), (int)(tapk[i] & 1023), p.g, vox) && ok;
                off = (int64_t)n * p.g.sample_pitch + (int64_t)vox * p.ldb + cch[i];
            }
            const f32x4 v = *reinterpret_cast<const f32x4*>(B + (ok ? off : 0));
            rb[i] = ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tile = [&]() {
        if (tid < 128) *reinterpret_cast<f32x4*>(&As[arow * PA + am]) = ra;
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<f32x4*>(&Bs[br * PB + (jc0 + 8 * i) * 4]) = rb[i];
    };

    const int xcol = wave * 64 + lane, wsel = lane & 3;
    if (r_begin < r_end) {
        load_tile(r_begin);
        for (int r0 = r_begin; r0 < r_end; r0 += 32) {
            store_tile();
            __syncthreads();
            if (r0 + 32 < r_end) load_tile(r0 + 32);
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const float xb = Bs[r * PB + xcol];
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(As[r * PA + g * 4 + wsel], xb, acc[g], 0, 0, 0);
            }
            __syncthreads();
        }
    }
    float* __restrict__ C;
    int64_t ldc;
    if (p.splits > 1) { C = p.ws + ((int64_t)z * p.splits + split) * p.M * p.N; ldc = p.N; }
    else { C = p.C + zo * p.sC_o + zi * p.sC_i; ldc = p.ldc; }
    const int col = n0 + xcol;
    if (col < p.N) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = g * 4 + i;
                if (m < p.M) C[(int64_t)m * ldc + col] = acc[g][i];
            }
    }
}

int launch_smallm_wgrad(const WgradArgs& a, int grid_z, hipStream_t s) {
    dim3 grid((unsigned)((a.N + 255) / 256), 1, grid_z);
    const int ng = (a.M + 3) / 4;
    if (ng == 1) hipLaunchKernelGGL((smallm_wgrad_kernel<1>), grid, dim3(256), 0, s, a);
    else if (ng == 2) hipLaunchKernelGGL((smallm_wgrad_kernel<2>), grid, dim3(256), 0, s, a);
    else if (ng == 3) hipLaunchKernelGGL((smallm_wgrad_kernel<3>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((smallm_wgrad_kernel<4>), grid, dim3(256), 0, s, a);
    CORRIF_CHECK_LAUNCH();
    return CORRIF_OK;
}
