/* corrif.h - C-ABI of the MI355X (gfx950) kernels behind the CorrIFNet `MMVit4` hot path.
 *
 * The reference (a pure-Python PyTorch repository) has no FFI layer: its "operator API" for this
 * path is the nn.Module protocol of `mmvit4.MMVit4` (mmvit4.py:391-532) whose forward/backward is
 * executed by ATen ops.  Each entry point below replaces the ATen op(s) named in its comment, for
 * the call sites cited (file:line in the upstream repository).  The reference-side binding a
 * maintainer would add is the ctypes stub shown in INTEGRATION.md (the host mirror in this repo is
 * `<package>/corrif_hip.py`).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless noted; fp32 data.
 *  - activations are channels-last: a tensor [B, D, H, W, C] is a matrix of B*D*H*W rows; `ld*`
 *    arguments are the row pitch in floats (>= channel count) so that channel slices of a wider
 *    (concatenated) buffer can be read / written in place.
 *  - the caller owns all memory (inputs, outputs, workspaces); kernels never allocate or free.
 *  - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no implicit sync.
 *  - return 0 on success, CORRIF_EINVAL (-1) bad argument, CORRIF_EUNSUPPORTED (-2),
 *    CORRIF_ELAUNCH (-3) launch failure (hipGetLastError != hipSuccess).  No C++ exception
 *    crosses this boundary.
 */
#ifndef CORRIF_H
#define CORRIF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CORRIF_OK 0
#define CORRIF_EINVAL (-1)
#define CORRIF_EUNSUPPORTED (-2)
#define CORRIF_ELAUNCH (-3)

#define CORRIF_ABI_VERSION 7   /* 7: split-bf16 main loops (CorrifGemm.f32_mfma replaces no_ksplit, CorrifWgrad.f32_mfma), corrif_trilinear_bwd_sep  6: grouped launches (CorrifGemm.zs_*, corrif_wgrad with Z x splits, per-group affine norms)  2: CorrifConv3Patch.fold  3: CorrifGemm.ws / no_split (stream-K), corrif_scale, corrif_fill  4: CorrifGemm.addend2, corrif_flash_attn_*  5: CorrifGemm.bstats_*, corrif_norm_bwd_pre, corrif_stem_*, corrif_depth_* */
int corrif_abi_version(void);
/* name of the gfx target the library was built for ("gfx950") - host-only call */
const char* corrif_build_arch(void);

/* ------------------------------------------------------------------------------------------
 * Geometry of an implicit-GEMM gather: how GEMM row r = (n, rd, rh, rw) and filter tap
 * t = (td, th, tw) address a source voxel.  For each axis
 *      num = r*mul + dir*t + off ;  valid iff num % div == 0 ; s = num / div
 * and then s must lie in [0, S) (zero padding) or is clamped into it (replicate padding).
 *   forward conv   : mul = stride, dir = +1, off = -pad, div = 1
 *   data gradient  : mul = 1,      dir = -1, off = +pad, div = stride   (transposed conv)
 * is_gemm = 1 skips all of it: source row = GEMM row (1x1x1 stride-1 convs, nn.Linear, bmm).
 * ------------------------------------------------------------------------------------------ */
typedef struct CorrifGeom {
    int32_t is_gemm;
    int32_t Rd, Rh, Rw;        /* per-sample grid the GEMM rows enumerate            */
    int32_t Sd, Sh, Sw;        /* per-sample grid of the source tensor               */
    int32_t kd, kh, kw;        /* filter taps                                        */
    int32_t mul_d, mul_h, mul_w;
    int32_t off_d, off_h, off_w;
    int32_t div_d, div_h, div_w;
    int32_t dir;               /* +1 / -1                                            */
    int32_t clamp;             /* 0 zero padding, 1 replicate padding                */
    int32_t ntaps;             /* valid taps (<= kd*kh*kw; K may be padded past it)  */
    int64_t src_batch_pitch;   /* floats between samples of the source; 0 = dense    */
} CorrifGeom;

/* activation codes for epilogues */
#define CORRIF_ACT_NONE 0
#define CORRIF_ACT_RELU 1
#define CORRIF_ACT_GELU 2      /* exact erf GELU (F.gelu default, mmvit4.py:341-345) */

/* C[z][M,N] = act( gather(A)[M,K] * B^T + bias[N] + addend[M,N] )   K = taps*Cs
 * Replaces aten::convolution forward for every nn.Conv3d with Cin % 4 == 0 on the path
 * (mmvit4.py:32,72,131-135,161-168,398-426,237-264; Cs = 1: the stem, mmvit4.py:120), aten::convolution_backward's data gradient
 * (geom.dir = -1), aten::addmm/mm of nn.Linear (mmvit4.py:301,303,351,354) and their input
 * gradients, and aten::bmm (q@k^T, attn@v: mmvit4.py:309,312) through the z batch.
 *   b_layout 0: B is [N][K] (K contiguous, pitch ldb)     1: B is [K][N] (N contiguous, pitch ldb)
 *   z = zo*Zi + zi ; operand X is offset by zo*sX_o + zi*sX_i floats (X in A,B,C).
 * FP32-input MFMA (v_mfma_f32_32x32x2_f32): results are exact-f32 fma chains. */
typedef struct CorrifGemm {
    const float* A; int64_t lda; int32_t Cs;     /* channels per tap (multiple of 4, or 1)  */
    const float* B; int64_t ldb; int32_t b_layout;
    float* C; int64_t ldc;
    const float* bias;                            /* [N] or NULL                             */
    const float* addend; int64_t ld_add;          /* [M][N] or NULL                          */
    int32_t M, N, K; int32_t act;
    int32_t Z, Zi;                                /* batch count (>=1), inner batch extent   */
    int64_t sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;
    CorrifGeom g;
    /* optional tap subset: K enumerates (tap_sel[0..ntap_sel), channel) instead of all kd*kh*kw taps (ntap_sel = 0: all).
     * Used by the parity-class data gradient of stride-2 convolutions, which only visits the taps that can be non-zero. */
    int32_t ntap_sel; int8_t tap_sel[28];
    /* optional output row map: GEMM row r = (n, rd, rh, rw) on the g.R* grid is stored at voxel
     * (n, rd*om_d + oo_d, rh*om_h + oo_h, rw*om_w + oo_w) of an OD x OH x OW grid (out_map = 0: row r itself).
     * Requires is_gemm = 0.  bias/addend are not combined with it. */
    int32_t out_map, OD, OH, OW, om_d, om_h, om_w, oo_d, oo_h, oo_w;
    /* optional fused statistics for the BatchNorm / InstanceNorm that follows (mmvit4.py:41-45,173,206-208): the epilogue also
     * accumulates, per output channel and per 64-row block, sum and sum of squares of the stored values (after max(.,0) when
     * stats_relu) into stats_part[((g*N + c)*chunks + chunk)*2 + {0,1}] (doubles), g = row / stats_rows_per_group,
     * chunks = ceil(stats_rows_per_group / 64); corrif_norm_stats_finalize turns them into mean / rstd.  Saves the
     * separate statistics pass over the conv output.  stats_rows_per_group must be a multiple of 64 unless it equals M. */
    double* stats_part; int64_t stats_rows_per_group; int32_t stats_relu;
    /* Stream-K split (ABI 3).  When the tile grid would fill the resident workgroup slots of the chip unevenly (e.g. 392 tiles on
     * 512 slots), the launch becomes a persistent grid whose workgroups each take an equal share of the (tile, K tile) iterations;
     * tiles cut by a share boundary are completed by a second, fixed-order reduction kernel (deterministic, no atomics).  The caller
     * provides `ws` with corrif_gemm_fwd_workspace(p) bytes (0 = this launch is not split; 16-byte aligned); no_split = 1 forces the
     * one-workgroup-per-tile launch (A/B measurements). */
    float* ws; int32_t no_split;
    /* second epilogue addend (ABI 4), only with `addend`: C = act(A.B + bias + addend + addend2).  Used by the data gradient of a
     * 1x1x1 convolution whose input has three consumers (a ResNet layer output feeds the next block's conv1, its downsample conv and
     * the encoder's adapt conv, mmvit4.py:176-186,204-212): the other two gradients ride along instead of two accumulation passes. */
    const float* addend2; int64_t ld_add2;
    /* backward statistics of a BatchNorm in the epilogue of the data-gradient GEMM that produces the gradient of its output (ABI 5;
     * mmvit4.py:204-212: conv -> BN -> ReLU -> conv): with bstats_x set, stats_part receives per column and 64-row block
     * (sum g', sum g' * xhat) instead of the forward statistics, g' = the stored value (addends included) where bstats_y > 0 (NULL: no
     * ReLU after that BatchNorm) and xhat = (bstats_x - bstats_mean) * bstats_rstd; bstats_x / bstats_y are [M][N] like C.
     * corrif_norm_bwd_pre consumes the partials: the separate reduction pass over (dy, x, y) of corrif_norm_bwd disappears. */
    const float* bstats_x; int64_t bstats_ldx; const float* bstats_y; int64_t bstats_ldy; const float* bstats_mean; const float* bstats_rstd;
    /* grouped launch (ABI 6): the three modality encoders (mmvit4.py:442-447 runs the same `Encoder` on x[:,0], x[:,1], x[:,2]) execute
     * their identical-shape layers as ONE launch with Z = 3 (Zi = 1).  A / B / C move by sA_o / sB_o / sC_o per group as before (stacked
     * activations: rows * ld; a concat buffer filled in place: the group's channel offset); the epilogue operands move by these strides
     * per outer batch index (floats; zs_stats in doubles): bias, addend, addend2, stats_part, bstats_x, bstats_y, bstats_mean / _rstd.
     * With Z > 1 the fused statistics are per group: stats_rows_per_group must equal M. */
    int64_t zs_bias, zs_add, zs_add2, zs_stats, zs_bsx, zs_bsy, zs_bsstat;
    /* Main loop of the 128-row tiles (the encoder's shapes): by default every fp32 operand is split exactly into three bf16 terms while it
     * is staged and the product is formed from six bf16 MFMA products with fp32 accumulation ("bf16x6": dropped terms <= 2^-25 |xy|; one
     * fp32 rounding per 16 products of the K sum - 0.36x the error of the fp32-input MFMA chain against fp64 at K = 2304-4608, and
     * 1.4-1.5x its speed; csrc/igemm_fwd.h, tools/split_lab.hip).  f32_mfma = 1 selects the v_mfma_f32_32x32x2_f32 loop instead (A/B). */
    int32_t f32_mfma;
} CorrifGemm;
int corrif_gemm_fwd(const CorrifGemm* p, void* stream);
int corrif_gemm_fwd_is_split(const CorrifGemm* p);      /* 1 = this launch takes the split-bf16 main loop (six bf16 MFMAs per fp32 product), 0 = v_mfma_f32_* */
size_t corrif_gemm_fwd_workspace(const CorrifGemm* p);   /* bytes of CorrifGemm.ws this launch needs; queries the device's CU count */

/* The encoder stem (mmvit4.py:120,172: Conv3d(1, 64, (3,7,7), stride (1,2,2), padding (1,3,3), bias=False) = inflate_conv of the
 * ResNet-50 conv1) on one modality plane x[b] = x + b*batch_pitch of the NCDHW input ([D][H][W] floats): patch-staged MFMA kernels
 * for the Cin = 1 case, where the implicit-GEMM loader would gather scalars.  wk / dwk: [64][148] (output channel major, the 147 taps
 * (td, th, tw) contiguous, column 147 zero) = corrif_weight_repack mode 0 / 2 with ldo = 148; y / dy: channels-last rows of 64.
 * ws: corrif_stem_wgrad_workspace() bytes.  corrif_stem_supported tells whether a layer has this geometry (else: corrif_gemm_fwd). */
int corrif_stem_supported(int32_t Co, int32_t kd, int32_t kh, int32_t kw, int32_t sd, int32_t sh, int32_t sw, int32_t pd, int32_t ph, int32_t pw);
int corrif_stem_fwd(const float* x, int64_t batch_pitch, const float* wk, float* y, int64_t ldy, int32_t B, int32_t D, int32_t H, int32_t W,
                    void* stream);
size_t corrif_stem_wgrad_workspace(void);
int corrif_stem_wgrad(const float* x, int64_t batch_pitch, const float* dy, int64_t lddy, float* dwk, float* ws, int32_t B, int32_t D, int32_t H,
                      int32_t W, void* stream);

/* W-type contraction over rows:  C[z][M,N] = sum_r A[r][M] * gather(B)[r][N]     N = taps*Cs
 * Replaces aten::convolution_backward's weight gradient (A = dY, B = layer input) for every
 * conv above, the weight gradients of nn.Linear, and the P^T*dO / dS^T*Q products of the
 * attention backward (mmvit4.py:309,312).  R rows are split over `splits` workgroup groups that
 * write fp32 slabs ws[split][M][N]; corrif_wgrad_reduce sums them deterministically. */
typedef struct CorrifWgrad {
    const float* A; int64_t lda;                  /* [R][M]                                  */
    const float* B; int64_t ldb; int32_t Cs;      /* gathered source, channels per tap       */
    float* C; int64_t ldc;                        /* [M][N] (used directly when splits == 1) */
    float* ws;                                    /* splits*M*N floats when splits > 1       */
    int32_t R, M, N; int32_t splits;
    int32_t Z, Zi; int64_t sA_o, sA_i, sB_o, sB_i, sC_o, sC_i;   /* batch; splits > 1 with Z > 1 (ABI 6): ws holds Z*splits slabs */
    CorrifGeom g;                                 /* rows = R enumerate g.R*, source = B     */
    int32_t f32_mfma;                             /* ABI 7: 1 = the fp32-input MFMA loop instead of the split-bf16 one (see CorrifGemm.f32_mfma) */
} CorrifWgrad;
int corrif_wgrad(const CorrifWgrad* p, void* stream);
int corrif_wgrad_is_split(const CorrifWgrad* p);       /* same question for a weight-gradient launch */
size_t corrif_wgrad_workspace(const CorrifWgrad* p);   /* bytes; host-only */
int corrif_wgrad_plan(int32_t R, int32_t M, int32_t N, int32_t Z);  /* recommended `splits` for a launch of Z batches; host-only */

/* out[i] = sum_j in[j*n + i], j < count (also used for bias gradients through col_sum) */
int corrif_slab_reduce(const float* ws, float* out, int64_t n, int32_t count, void* stream);
/* out[c] = sum_r x[r*ld + c]  (bias gradient; aten::sum in convolution_backward / addmm backward) */
int corrif_col_sum(const float* x, int64_t ld, int64_t rows, int32_t C, float* out, double* ws, void* stream);
size_t corrif_col_sum_workspace(int64_t rows, int32_t C);

/* weight re-layouts (private caches of the host layer; reference layout is O,I,kd,kh,kw):
 *   mode 0: out[o][t][i] = w[o][i][t]   (forward / weight-gradient layout, [N][K])
 *   mode 1: out[t][o][i] = w[o][i][t]   (data-gradient layout, [K][N], taps NOT flipped: dir=-1)
 *   mode 2: w[o][i][t] = in[o][t][i]    (weight-gradient back to the reference layout)
 *   mode 3: out[ch][o][t][c] = w[o][ch*cc + c][t]           (patch kernel, forward;  ldo carries cc)
 *   mode 4: out[ch][i][t'][c] = w[ch*cc + c][i][T-1-t']     (patch kernel, data gradient; ldo carries cc) */
int corrif_weight_repack(const float* in, float* out, int32_t O, int32_t I, int32_t T, int32_t mode, int64_t ldo, void* stream);
/* ldo: row pitch (floats, >= T*I) of the [o][t][i] side in modes 0 and 2 (the stem pads 147 -> 148) */

/* Patch-staged direct 3x3x3 convolution (stride 1) for narrow layers, Cout <= 32, Cin % 8 == 0: the decoder's
 * d*_c1 / d*_c2 and RFM 3x3x3 convs (mmvit4.py:52,225-235) and their data gradients.  The workgroup stages the halo'd input
 * patch of a 256-voxel output tile in LDS once per channel chunk; taps are LDS reads; v_mfma_f32_4x4x1_16b_f32 with A-operand
 * broadcast.  out[o] = bias + sum_t W[t] . X[clamp|zero(o + t - pad)],  o in the Od x Oh x Ow grid.
 *   forward      : pad = 1, (Od,Oh,Ow) = (Sd,Sh,Sw), clamp = replicate?   weights from corrif_weight_repack mode 3
 *   data gradient: X = dY, weights mode 4 (flipped taps, transposed channels); zero padding: pad = 1 same grid;
 *                  replicate padding: pad = 2 on the (S+2)^3 grid, then corrif_pad_fold (or fold = 1: fused).
 * cc must equal corrif_conv3_patch_cc(Ci, Co) (channel chunk, 16 or 8; 0 = shape not supported). */
typedef struct CorrifConv3Patch {
    const float* X; int64_t ldx;
    const float* Wp;                 /* [Ci/cc][Co][27][cc] */
    float* Y; int64_t ldy;
    const float* bias;               /* [Co] or NULL */
    int32_t B, Sd, Sh, Sw, Od, Oh, Ow, Ci, Co, pad, clamp, cc;
    int32_t fold;                    /* 1: data gradient of a replicate-padded conv with the padding adjoint fused: O = (n+2)^3 padded grid,
                                      * pad = 2, Y = the n^3 input gradient (rows of the n grid); needs n_d % 4 == n_h % 4 == n_w % 16 == 0 */
    /* forward epilogue extras (ABI 6; general_conv3d_prenorm, mmvit4.py:41-45: conv -> ReLU -> InstanceNorm3d):
     *   stats_part != NULL: per (sample, channel) sum and sum of squares of the stored values (after max(., 0) when stats_relu) into
     *     stats_part[((b*Co + c)*stats_chunks + slot)*2 + {0,1}] (doubles, PRE-ZEROED by the caller, stats_chunks =
     *     corrif_conv3_patch_stats_chunks(...)); corrif_norm_stats_finalize(stats_part, stats_chunks, B, Co, Od*Oh*Ow, ...) turns them
     *     into mean / rstd - the separate statistics pass over the conv output disappears;
     *   add_src != NULL: Y[b,d,h,w,:] += add_src[b, cls(d), h, w, :] with the depth classes of corrif_depth_bcast_add (add_Ds source
     *     slices, add_src on the 3*add_Ds grid, row pitch ld_add) before the statistics - the compact skip branch's share of d*_c2
     *     (mmvit4.py:271-287) without its own pass over the full-depth tensor. */
    double* stats_part; int32_t stats_chunks, stats_relu;
    const float* add_src; int64_t ld_add; int32_t add_Ds;
} CorrifConv3Patch;
int corrif_conv3_patch(const CorrifConv3Patch* p, void* stream);
int corrif_conv3_patch_stats_chunks(int32_t B, int32_t Od, int32_t Oh, int32_t Ow);   /* host-only */
int corrif_conv3_patch_stats_supported(int32_t Ci, int32_t Co);                       /* host-only: Co = 8, or Co = 16 with 8-channel chunks */
int corrif_conv3_patch_cc(int32_t Ci, int32_t Co);     /* host-only */
/* weight gradient of the same layers (Cout <= 16, Cin % 16 == 0): dW[Co][27][Ci] (then corrif_weight_repack mode 2).
 * X is the layer input (S grid), DY the output gradient (O grid = S grid, pad 1), ws = corrif_conv3_patch_wgrad_workspace bytes. */
typedef struct CorrifConv3PatchWgrad {
    const float* X; int64_t ldx;
    const float* DY; int64_t lddy;
    float* dW; float* ws;
    int32_t B, Sd, Sh, Sw, Od, Oh, Ow, Ci, Co, pad, clamp;
} CorrifConv3PatchWgrad;
int corrif_conv3_patch_wgrad(const CorrifConv3PatchWgrad* p, void* stream);
size_t corrif_conv3_patch_wgrad_workspace(int32_t Ci, int32_t Co);   /* host-only; 0 = shape not supported */
int corrif_conv3_patch_wgrad_slots(int32_t Ci, int32_t Co);          /* host-only */

/* Tiny-channel 1x1x1 convolution, Ci = Co in {8, 16}: d1_out / d2_out (mmvit4.py:233,236), 2-4 FLOP/B -> pure HBM stream, one
 * voxel per thread, weights broadcast from LDS.  w_transposed = 1 makes it the data gradient (dX = dY . W).  The weight-gradient
 * entry also produces the bias gradient. */
int corrif_conv1x1_small_supported(int32_t Ci, int32_t Co);   /* host-only */
int corrif_conv1x1_small_fwd(const float* x, int64_t ldx, const float* w, int32_t w_transposed, const float* bias, float* y, int64_t ldy,
                             int64_t rows, int32_t Ci, int32_t Co, void* stream);
int corrif_conv1x1_small_wgrad(const float* x, int64_t ldx, const float* dy, int64_t lddy, float* dw, float* db, double* ws,
                               int64_t rows, int32_t Ci, int32_t Co, void* stream);
size_t corrif_conv1x1_small_workspace(int64_t rows, int32_t Ci, int32_t Co);

/* Stem: Conv3d(1->64,(3,7,7),stride (1,2,2),pad (1,3,3), no bias) on x[:, m] of the NCDHW input
 * (mmvit4.py:120,172; aten::convolution with Cin = 1) runs through corrif_gemm_fwd / corrif_wgrad
 * with Cs = 1 (scalar gather): lda = 1, g.src_batch_pitch = 3*D*H*W, K padded 147 -> 148 with
 * g.ntaps = 147 and the packed weight [64][148] zero in its last column. */

/* ------------------------------------------------------------------------------------------
 * Normalisation over rows, channels-last.  groups G = 1 (BatchNorm3d: statistics over all rows,
 * mmvit4.py:121,132-136,143) or G = B (InstanceNorm3d, affine=False: statistics per sample,
 * mmvit4.py:24,34,74).  Replaces aten::native_batch_norm(+_backward) and the ReLU / residual add
 * around it.   flags:
 *   CORRIF_NORM_RELU_IN   x' = max(x,0) before the statistics       (conv->ReLU->IN, stem ReLU->BN)
 *   CORRIF_NORM_RELU_OUT  y = max(y,0)                               (Bottleneck3D bn1/bn2/bn3)
 *   CORRIF_NORM_EVAL      use mean/rstd as given (running statistics), no statistics pass
 * ------------------------------------------------------------------------------------------ */
#define CORRIF_NORM_RELU_IN 1
#define CORRIF_NORM_RELU_OUT 2
/* statistics: mean[G*C], rstd[G*C] (biased variance, eps); optional running-stat update (momentum,
 * unbiased variance) when running_mean != NULL (G must be 1).  ws: corrif_norm_workspace bytes. */
int corrif_norm_stats(const float* x, int64_t ldx, int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, float eps,
                      float* mean, float* rstd, float* running_mean, float* running_var, float momentum,
                      double* ws, void* stream);
/* second half of corrif_norm_stats for partials produced by a GEMM epilogue (CorrifGemm.stats_part) */
int corrif_norm_stats_finalize(const double* part, int32_t chunks, int32_t G, int32_t C, int64_t rows_per_group, float eps,
                               float* mean, float* rstd, float* running_mean, float* running_var, float momentum, void* stream);
/* rstd[c] = 1/sqrt(var[c] + eps) for eval mode */
int corrif_norm_eval_rstd(const float* running_var, float eps, float* rstd, int32_t C, void* stream);
/* y = act_out( gamma*(x' - mean)*rstd + beta + residual ) ; gamma/beta/residual may be NULL */
int corrif_norm_apply(const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma, const float* beta,
                      const float* residual, int64_t ldr, float* y, int64_t ldy,
                      int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, void* stream);
/* backward.  g = dy * (y > 0) if RELU_OUT.  Produces
 *   dx = [x>0 if RELU_IN] * gamma*rstd*( g - mean_g - xhat*mean_gxhat )      (train)
 *   dx = [..] * gamma*rstd*g                                                (frozen = 1, eval statistics)
 *   dres = g (optional), dgamma[c] = sum g*xhat, dbeta[c] = sum g (optional, G must be 1) */
int corrif_norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx,
                    const float* mean, const float* rstd, const float* gamma,
                    float* dx, int64_t lddx, float* dres, int64_t lddres, float* dgamma, float* dbeta,
                    int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, int32_t frozen, double* ws, void* stream);
/* the same with the reduction already done: part = per-(channel, 64-row block) partials (sum g', sum g' * xhat) written by the epilogue of
 * the data-gradient GEMM that produced dy (CorrifGemm.bstats_*), chunks blocks per channel; G = 1 (BatchNorm).  ws: corrif_norm_workspace. */
int corrif_norm_bwd_pre(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx, const float* mean,
                        const float* rstd, const float* gamma, float* dx, int64_t lddx, float* dres, int64_t lddres, float* dgamma,
                        float* dbeta, int64_t rows, int32_t C, int32_t flags, const double* part, int32_t chunks, double* ws, void* stream);
size_t corrif_norm_workspace(int64_t rows_per_group, int32_t G, int32_t C);
/* Grouped variants (ABI 6) for the three modality encoders run as one stacked pass (mmvit4.py:442-447: the same Encoder on x[:,0], x[:,1],
 * x[:,2]): group g = modality g, rows_per_group rows each, its own nn.BatchNorm3d.  affine_gstride = C: gamma / beta / dgamma / dbeta are
 * [G][C] (one parameter set per group); 0: shared [C] as in the plain entries.  running_means / running_vars: host arrays of G (<= 4)
 * device pointers, one per group's module buffers, or NULL. */
size_t corrif_norm_workspace_g(int64_t rows_per_group, int32_t G, int32_t C);   /* ws of corrif_norm_stats_g / _bwd_g / corrif_col_sum_g: every group is
                                                                                  * chunked like a standalone G = 1 launch (bit-identical results) */
int corrif_norm_stats_g(const float* x, int64_t ldx, int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, float eps, float* mean,
                        float* rstd, float* const* running_means, float* const* running_vars, float momentum, double* ws, void* stream);
int corrif_norm_stats_finalize_g(const double* part, int32_t chunks, int32_t G, int32_t C, int64_t rows_per_group, float eps, float* mean,
                                 float* rstd, float* const* running_means, float* const* running_vars, float momentum, void* stream);
int corrif_norm_apply_g(const float* x, int64_t ldx, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        const float* residual, int64_t ldr, float* y, int64_t ldy, int64_t rows_per_group, int32_t G, int32_t C,
                        int32_t flags, int64_t affine_gstride, void* stream);
int corrif_norm_bwd_g(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx, const float* mean,
                      const float* rstd, const float* gamma, float* dx, int64_t lddx, float* dres, int64_t lddres, float* dgamma, float* dbeta,
                      int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, int32_t frozen, int64_t affine_gstride, double* ws,
                      void* stream);
int corrif_norm_bwd_pre_g(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* x, int64_t ldx, const float* mean,
                          const float* rstd, const float* gamma, float* dx, int64_t lddx, float* dres, int64_t lddres, float* dgamma,
                          float* dbeta, int64_t rows_per_group, int32_t G, int32_t C, int32_t flags, int64_t affine_gstride,
                          const double* part, int32_t chunks, double* ws, void* stream);
/* out[g][c] = sum over the `rows` consecutive rows of group g (bias gradients of a grouped convolution) */
int corrif_col_sum_g(const float* x, int64_t ld, int64_t rows, int32_t G, int32_t C, float* out, double* ws, void* stream);
/* dst[g][0..n) = srcs[g][0..n): the weights / affine parameters of G (<= 4) same-shaped modules (the three modality encoders' twin
 * layers, mmvit4.py:394-396) gathered into one stacked operand of a grouped launch; srcs = host array of G device pointers */
int corrif_stack_groups(const float* const* srcs, int32_t G, float* dst, int64_t n, void* stream);
/* dst[entry.dst_off + i] = entry.src[i] (0 when src is NULL) for every entry of `table` ({const float* src; int64_t dst_off; int64_t n}
 * records in device memory) in ONE launch: block b handles elements [blk_off[b], blk_off[b] + 1024) of tensor blk_tensor[b].  Assembles
 * a gradient all-reduce bucket from the parameter gradients the backward kernels produced (data-parallel layer). */
int corrif_gather_multi(const void* table, const int32_t* blk_tensor, const int64_t* blk_off, int32_t nblocks, float* dst, void* stream);

/* LayerNorm over the last dim (C = 512), eps 1e-5 (mmvit4.py:327,335; aten::native_layer_norm).
 * Optional fused pre-add: xin = x + pos[row % pos_rows] (Transformer.forward `x = x + pos`,
 * mmvit4.py:385), written to xsum when non-NULL. */
int corrif_layernorm_fwd(const float* x, const float* pos, int64_t pos_rows, float* xsum, const float* gamma, const float* beta,
                         float* y, float* mean, float* rstd, int64_t rows, int32_t C, float eps, void* stream);
int corrif_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                         float* dx, double* ws, float* dgamma, float* dbeta, int64_t rows, int32_t C, void* stream);
size_t corrif_layernorm_workspace(int64_t rows, int32_t C);

/* MaxPool3d((1,3,3),(1,2,2),(0,1,1)) channels-last (mmvit4.py:123,174; aten::max_pool3d_with_indices).
 * idx: int8 window position (first maximum in (kh,kw) scan order, as ATen). */
int corrif_maxpool133_fwd(const float* x, float* y, int8_t* idx, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, void* stream);
int corrif_maxpool133_bwd(const float* dy, const int8_t* idx, float* dx, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, void* stream);

/* F.interpolate(mode='trilinear', align_corners=True) / nn.Upsample (mmvit4.py:187-191,243,263;
 * aten::upsample_trilinear3d[_backward]); channels-last, float index arithmetic as ATen.
 * The backward is a deterministic gather (no atomics). */
int corrif_trilinear_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t B, int32_t C,
                         int32_t Di, int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, void* stream);
int corrif_trilinear_bwd(const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t B, int32_t C,
                         int32_t Di, int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, void* stream);
/* The same adjoint for a pure up-sampling (Do>=Di, Ho>=Hi, Wo>=Wi; nn.Upsample(scale_factor=2), mmvit4.py:243) of contiguous tensors
 * (ld == C), one axis per pass (D, H, W) through `ws`: every incoming gradient is read once instead of once per input voxel that
 * touches it.  Deterministic; per input voxel the outputs are summed in ascending order per axis.
 * corrif_trilinear_bwd_sep_workspace: bytes of `ws`, or -1 when the geometry is not a pure up-sampling / too large. */
int64_t corrif_trilinear_bwd_sep_workspace(int32_t B, int32_t C, int32_t Di, int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo);
int corrif_trilinear_bwd_sep(const float* dy, float* dx, float* ws, int32_t B, int32_t C,
                             int32_t Di, int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, void* stream);
/* F.interpolate(size) default mode 'nearest' (mmvit4.py:271,276,281,286; aten::upsample_nearest3d) */
int corrif_nearest_fwd(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t B, int32_t C,
                       int32_t Di, int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, void* stream);
int corrif_nearest_bwd(const float* dy, int64_t lddy, float* dx, int64_t lddx, int32_t B, int32_t C,
                       int32_t Di, int32_t Hi, int32_t Wi, int32_t Do, int32_t Ho, int32_t Wo, void* stream);
/* adjoint of replicate padding (aten::replication_pad3d_backward, padding_mode='replicate' at
 * mmvit4.py:225-235): dx[n,d,h,w] = sum of dxp over padded voxels that clamp to (d,h,w); dxp is
 * the data gradient computed on the (D+2)(H+2)(W+2) padded grid. */
int corrif_pad_fold(const float* dxp, float* dx, int64_t lddx, int32_t B, int32_t D, int32_t H, int32_t W, int32_t C, void* stream);

/* Row softmax with scale, in place: p = softmax(s*scale) (mmvit4.py:309-310; aten::_softmax),
 * and its backward ds = scale * p * (dp - sum(dp*p)) (aten::_softmax_backward_data). */
int corrif_softmax_rows(float* s, int64_t rows, int32_t n, float scale, void* stream);
int corrif_softmax_rows_bwd(const float* p, float* dp_to_ds, int64_t rows, int32_t n, float scale, void* stream);

/* softmax + attention dropout fused (mmvit4.py:310-311): s <- P = softmax(s*scale) in place (kept for the backward), pd <- P * keep/(1-p)
 * with the same Philox stream and element indexing as corrif_dropout(offset); backward: dS = scale*P*(g - sum(P*g)), g = dP'*keep/(1-p). */
int corrif_softmax_dropout_rows(float* s, float* pd, int64_t rows, int32_t n, float scale, float p, uint64_t seed, uint64_t offset, void* stream);
int corrif_softmax_dropout_rows_bwd(const float* pr, float* dpd_to_ds, int64_t rows, int32_t n, float scale, float p, uint64_t seed,
                                    uint64_t offset, void* stream);

/* Flash-style multi-head self-attention, head dimension 64 (replaces the whole of mmvit4.py:305-312: q@k^T * scale, softmax,
 * attn_drop, attn@v) without writing the [B, heads, N, N] score tensor.  qkv: [B][N][3*heads*64] = the qkv Linear's output, laid
 * out as reshape(B, N, 3, heads, 64); out: [B][N][heads*64] (= the transpose(1,2).reshape of mmvit4.py:312); lse: [B*heads][N]
 * row log-sum-exp of the scaled scores, kept for the backward.  Dropout mask = the Philox stream of corrif_dropout over the flat
 * [B, heads, N, N] index at `offset` (p = 0: no dropout, mask may be NULL), so it equals corrif_softmax_dropout_rows on a
 * materialised score tensor; the forward leaves it as keep bits in mask ([B*heads][N][N/32] words = corrif_flash_attn_mask_bytes,
 * 1/32 of a score tensor) for the backward.  Backward recomputes the probabilities from qkv and lse: dvec [B*heads][N] is scratch
 * (rowsum(dO * O)), dqkv [B][N][3*heads*64] is fully written.  Deterministic (no atomics): dq and (dk, dv) come from two kernels
 * that own their output rows.  N must be a multiple of 128; fp32 MFMA (v_mfma_f32_32x32x2_f32) throughout. */
int corrif_flash_attn_supported(int32_t N, int32_t head_dim);
size_t corrif_flash_attn_mask_bytes(int32_t B, int32_t N, int32_t heads, float p);
int corrif_flash_attn_fwd(const float* qkv, float* out, float* lse, uint32_t* mask, int32_t B, int32_t N, int32_t heads, float scale,
                          float p, uint64_t seed, uint64_t offset, void* stream);
int corrif_flash_attn_bwd(const float* qkv, const float* out, const float* lse, const uint32_t* mask, const float* dout, float* dvec,
                          float* dqkv, int32_t B, int32_t N, int32_t heads, float scale, float p, void* stream);

/* Dropout with a counter-based Philox4x32-10 stream (replaces aten::bernoulli_/native_dropout at
 * mmvit4.py:302,304,336,353,355): y = x * keep/(1-p), keep = u(seed, offset + i) >= p.
 * The same call is its own backward (dx = dy * keep/(1-p)). x may alias y. */
int corrif_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream);

/* Depth-class broadcast for the decoder's skip branch (mmvit4.py:271-287: F.interpolate(nearest) -> cat -> d*_c2).  The nearest
 * up-sampling of Ds depth slices to D >= 2*Ds slices (source of slice d = min(floor(d*Ds/D), Ds-1), ATen's float arithmetic; D need
 * not be a multiple of Ds since ABI 6: the reference-native 3 bands, 12 bands) is constant along depth inside each block of slices that
 * share a source, so a replicate-padded 3x3x3 convolution of it takes three distinct values per block (first slice / interior / last
 * slice).  The host evaluates that share of the convolution on a compact grid of 3*Ds slices (class 3k + {0,1,2} of block k) and these
 * two calls move it in and out of the full-depth tensor (channels-last, S = H*W voxels per slice, row pitches ld*):
 *   corrif_depth_bcast_add:    y[b,d,s,:] += ys[b, cls(d), s, :]                     (in place on the convolution of the other channels)
 *   corrif_depth_class_reduce: out[b,c,s,:] = sum over d with cls(d) = c of g[b,d,s,:]   (its adjoint; fixed summation order) */
int corrif_depth_bcast_add(float* y, int64_t ldy, const float* ys, int64_t lds, int32_t B, int32_t D, int32_t S, int32_t C, int32_t Ds, void* stream);
int corrif_depth_class_reduce(const float* g, int64_t ldg, float* out, int64_t ldo, int32_t B, int32_t D, int32_t S, int32_t C, int32_t Ds, void* stream);

/* element-wise helpers (aten::add / gelu_backward / mul): y = a + b ; dx = dy * gelu'(x) */
int corrif_add(const float* a, const float* b, float* y, int64_t n, void* stream);
int corrif_add_bcast_rows(const float* a, const float* b, int64_t b_n, float* y, int64_t n, void* stream); /* b index = i % b_n */
int corrif_gelu_fwd(const float* x, float* y, int64_t n, void* stream);   /* exact erf GELU, mmvit4.py:341-345 */
int corrif_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);
int corrif_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* y = x * scalar[0], scalar on the device (upstream gradient of the scalar loss) */
int corrif_scale_dev(const float* x, const float* scalar, float* y, int64_t n, void* stream);
/* y = x * alpha (host scalar; the 1/world averaging of the data-parallel gradient buckets, x may alias y) ; y[:] = value
 * (zeroing the gradient buckets): replace aten::mul_ / aten::zero_ on the training step's path */
int corrif_scale(const float* x, float* y, int64_t n, float alpha, void* stream);
int corrif_fill(float* y, int64_t n, float value, void* stream);
/* strided 2-D copy / accumulate: dst[r*ldd + c] (+)= src[r*lds + c], c < C (torch.cat / chunk grads) */
int corrif_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t C, int32_t accumulate, void* stream);
/* out[c] = sum over rows r of x[r*ld + c] for r in a group of `rows` (pos gradient: sum over batch) */
int corrif_sum_groups(const float* x, float* out, int64_t group_elems, int32_t groups, void* stream);

/* Inter-modal correlation, mmvit4.py:481-491 (+ the [3,B*C*S] -> [B,3C,S] re-view of :485).
 * qkv[m]: channels-last [B*S, 3*C] rows (q | k | v column blocks) of modality m, S = 512, C = 512.
 * out[m][b'][s][c] = sum_i' w * v_i'[b'][s][c],   w = softmax_i( q_m[b]*k_i[b] / sqrt(3) )[i]
 * with (i, b) = divmod(3*b' + i', B) evaluated at the same (c, s)            
 * out: 3 tensors [B*S, C] (pitch ldo).  Backward returns dqkv[m] in the same layout. */
int corrif_intercorr_fwd(const float* qkv0, const float* qkv1, const float* qkv2, int64_t ldq,
                         float* out0, float* out1, float* out2, int64_t ldo, int32_t B, int32_t S, int32_t C, void* stream);
int corrif_intercorr_bwd(const float* qkv0, const float* qkv1, const float* qkv2, int64_t ldq,
                         const float* do0, const float* do1, const float* do2, int64_t ldo,
                         float* dqkv0, float* dqkv1, float* dqkv2, int32_t B, int32_t S, int32_t C, void* stream);

/* Head: final_conv (Conv3d 8->3, 1x1x1, bias) + sigmoid, channels-last [rows,8] in, NCDHW
 * [B,3,1,224,224] out (mmvit4.py:264,290-291).  Backward: d(logit) = dpred * p*(1-p), then
 * dx[rows,8], and slabs for dw[3][8], db[3]. */
int corrif_head_fwd(const float* x, const float* w, const float* b, float* pred, int32_t B, int32_t HW, void* stream);
int corrif_head_bwd(const float* dpred, const float* pred, const float* x, const float* w,
                    float* dx, float* dw, float* db, double* ws, int32_t B, int32_t HW, void* stream);
size_t corrif_head_workspace(int32_t B, int32_t HW);

/* Loss of the training step, F4_TRAIN.py:58-60: mean BCE-with-logits applied to the (already
 * sigmoided) prediction; loss: 1 float; dpred = d loss / d pred (optional). */
int corrif_bce_logits_mean(const float* pred, const float* target, int64_t n, float* loss, float* dpred, double* ws, void* stream);
size_t corrif_bce_workspace(int64_t n);

/* Jaccard2 / Jaccard / JaccardAndF1 of F5_JACCARD2.py:4-37 on y[n], y_pred[n]:
 * out[0] = Jaccard2, out[1] = Jaccard (no complement branch), out[2] = F1 (JaccardAndF1).
 * Partial sums are fp32 per lane-chunk (exact for 0/1 inputs) combined in a fixed order. */
int corrif_jaccard(const float* y, const float* y_pred, int64_t n, float eps, float* out, float* ws, void* stream);
size_t corrif_jaccard_workspace(int64_t n);

/* Adam step, torch.optim.Adam defaults as F2_MAIN.py:168-169 (N1): p, m, v updated in place. */
int corrif_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int32_t step, void* stream);

/* Input pipeline, F8_IMAGES4.py:36-88: raw pixel-interleaved patches rgb [N,HW,3] and all20 [N,HW,20] -> planar network
 * input images [N,3,3,HW] = (R,G,B | bands 9,10,11 | bands 12,13,14) minus the per-band mean over the TRAINING samples `trind`
 * (means[9], F8_IMAGES4.py:60-79), and targets [N,3,1,HW] = the mask repeated over three channels (F8_IMAGES4.py:88). */
int corrif_prep_means(const float* rgb, const float* all20, const int32_t* trind, int32_t ntr, int32_t HW, float* means, double* ws, void* stream);
int corrif_prep_stack(const float* rgb, const float* all20, const float* masks, const float* means, float* images, float* targets,
                      int32_t N, int32_t HW, void* stream);
size_t corrif_prep_workspace(int32_t ntr, int32_t HW);

/* the same update for ALL parameters in one launch.  table: device array of {float* p; const float* g; float* m; float* v;
 * int64_t n;} (40 bytes each); block b updates elements [blk_off[b], blk_off[b]+1024) of tensor blk_tensor[b]. */
int corrif_adam_multi(const void* table, const int32_t* blk_tensor, const int64_t* blk_off, int32_t nblocks, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int32_t step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CORRIF_H */
