// Launch descriptors shared by the implicit-GEMM kernel files.
#pragma once
#include "common.h"

#define BK 32
#define LDS_PITCH (BK + 4)   // floats; 144-byte rows keep the ds_read_b128 lane groups conflict-free

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias; const float* addend;
    int64_t lda, ldb, ldc, ld_add;
    int M, N, K, Cs4, act, b_layout, Zi;
    int64_t sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
    DevGeom g;
    int ntap_sel; int8_t tap_sel[28];
    int out_map, OD, OH, OW, om_d, om_h, om_w, oo_d, oo_h, oo_w;
    double* stats_part; int stats_relu; int stats_rpg, stats_chunks;
    // stream-K split (igemm.hip): G persistent workgroups share sk_tiles * sk_nk (tile, K tile) iterations; partial accumulators go to
    // sk_ws[2 * G][BM * BN]; sk_tiles = sk_tiles_mn * Z
    float* sk_ws; int sk_G, sk_nk, sk_tiles, sk_tiles_mn, sk_allowed;
    const float* addend2; int64_t ld_add2;      // second epilogue addend (a tensor with three consumers: two gradients ride along)
    // backward statistics of the BatchNorm that produced this GEMM's input (data-gradient launches): with bs_x set, stats_part receives
    // per column and 64-row block (sum g', sum g' * xhat), g' = the value this epilogue stores masked by bs_y > 0, xhat = (bs_x - mean) * rstd
    const float* bs_x; int64_t bs_ldx; const float* bs_y; int64_t bs_ldy; const float* bs_mean; const float* bs_rstd;
    // grouped launches (Z = the three modality encoders' identical-shape layers in ONE launch): per OUTER batch index zo the epilogue
    // operands move by these strides (floats; zs_stats in doubles)
    int64_t zs_bias, zs_add, zs_add2, zs_stats, zs_bsx, zs_bsy, zs_bsstat;
    int f32_mfma;       // 1 = the v_mfma_f32_32x32x2_f32 main loop even where the split-bf16 loop exists (A/B, see SPLIT in igemm.hip)
};


struct WgradArgs {
    const float* A; const float* B; float* C; float* ws;
    int64_t lda, ldb, ldc;
    int R, M, N, Cs, splits, rows_per_split, Zi;
    int64_t sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
    DevGeom g;
    int Z;              // batch count; with splits > 1 the grid enumerates (z, split, tile) and the slabs are ws[z][split][M][N]
};


// small-N / small-M variants on v_mfma_f32_4x4x1_16b_f32 (igemm_small.hip)
int launch_smalln_fwd(const GemmArgs& a, int Z, hipStream_t s);     // N <= 16, N % 4 == 0
int launch_smallm_wgrad(const WgradArgs& a, int grid_z, hipStream_t s);   // M <= 16, M % 4 == 0
