// Host side of corrif_gemm_fwd (validation, tile choice, launch) and the plain-GEMM half of the kernel family; the kernels and the
// per-variant launcher live in igemm_fwd.h, the implicit-convolution half is instantiated in igemm_conv.hip, the weight gradients in
// igemm_wgrad.hip.
#include "igemm_fwd.h"

thread_local int g_plan_split = 0;

#define CORRIF_CONV_VARIANT(BM, BN, WM, WN) \
    extern template int launch_variant<BM, BN, WM, WN, 4, false, 0>(GemmArgs&, int, hipStream_t, bool, size_t*); \
    extern template int launch_variant<BM, BN, WM, WN, 4, false, 1>(GemmArgs&, int, hipStream_t, bool, size_t*);
CORRIF_CONV_VARIANT(128, 128, 2, 2)
CORRIF_CONV_VARIANT(128, 64, 2, 2)
CORRIF_CONV_VARIANT(64, 64, 2, 2)
CORRIF_CONV_VARIANT(256, 32, 4, 1)
#undef CORRIF_CONV_VARIANT

template <int BM, int BN, int WM, int WN, int VEC = 4>
static int launch_fwd(GemmArgs& a, int Z, hipStream_t s, bool plan_only, size_t* ws_bytes) {
    if constexpr (VEC == 1) {
        return launch_variant<BM, BN, WM, WN, 1, false, 0>(a, Z, s, plan_only, ws_bytes);
    } else if (a.g.is_gemm) {
        if (a.b_layout == 0) return launch_variant<BM, BN, WM, WN, VEC, true, 0>(a, Z, s, plan_only, ws_bytes);
        return launch_variant<BM, BN, WM, WN, VEC, true, 1>(a, Z, s, plan_only, ws_bytes);
    } else {
        if (a.b_layout == 0) return launch_variant<BM, BN, WM, WN, VEC, false, 0>(a, Z, s, plan_only, ws_bytes);
        return launch_variant<BM, BN, WM, WN, VEC, false, 1>(a, Z, s, plan_only, ws_bytes);
    }
}

// validation + tile choice + launch (plan_only: only report the stream-K workspace the launch would need)
static int gemm_fwd_impl(const CorrifGemm* p, void* stream, bool plan_only, size_t* ws_bytes) {
    if (ws_bytes) *ws_bytes = 0;
    if (!p || !p->A || !p->B || !p->C) return CORRIF_EINVAL;
    if (p->M <= 0 || p->N <= 0 || p->K <= 0 || p->Z < 1 || p->Zi < 1) return CORRIF_EINVAL;
    const bool scalar = p->Cs == 1;
    if (p->Cs <= 0 || (p->K & 3) || (p->ldb & 3)) return CORRIF_EUNSUPPORTED;
    if (!scalar && ((p->Cs & 3) || (p->lda & 3) || ((uintptr_t)p->A & 15) || (p->sA_o & 3) || (p->sA_i & 3) ||
                    (p->g.src_batch_pitch & 3)))
        return CORRIF_EUNSUPPORTED;
    if (scalar && (p->g.is_gemm || p->b_layout != 0)) return CORRIF_EUNSUPPORTED;
    if (((uintptr_t)p->B & 15)) return CORRIF_EUNSUPPORTED;
    if ((p->sB_o & 3) || (p->sB_i & 3)) return CORRIF_EUNSUPPORTED;
    if (p->b_layout == 1 && (p->N & 3)) return CORRIF_EUNSUPPORTED;
    if (p->b_layout != 0 && p->b_layout != 1) return CORRIF_EINVAL;
    if (!geom_ok(p->g)) return CORRIF_EINVAL;
    if (!p->g.is_gemm && (int64_t)p->g.Sd * p->g.Sh * p->g.Sw * p->lda >= (int64_t)1 << 31) return CORRIF_EUNSUPPORTED;   // 32-bit in-sample offsets
    if (p->ntap_sel < 0 || p->ntap_sel > 28 || (p->ntap_sel && (p->g.is_gemm || scalar))) return CORRIF_EINVAL;
    if (p->out_map && (p->g.is_gemm || p->addend || p->addend2 || p->N > 4096 * 1024)) return CORRIF_EINVAL;
    if (p->addend2 && !p->addend) return CORRIF_EINVAL;
    if (!p->g.is_gemm && !scalar && p->K != (p->ntap_sel ? p->ntap_sel : p->g.kd * p->g.kh * p->g.kw) * p->Cs) return CORRIF_EINVAL;
    if (scalar && (p->g.ntaps <= 0 || p->g.ntaps > p->g.kd * p->g.kh * p->g.kw || p->K < p->g.ntaps)) return CORRIF_EINVAL;
    if (p->Z > 65535) return CORRIF_EUNSUPPORTED;
    if (p->ws && ((uintptr_t)p->ws & 15)) return CORRIF_EUNSUPPORTED;
    GemmArgs a;
    a.A = p->A; a.B = p->B; a.C = p->C; a.bias = p->bias; a.addend = p->addend; a.addend2 = p->addend2; a.ld_add2 = p->ld_add2;
    a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc; a.ld_add = p->ld_add;
    a.M = p->M; a.N = p->N; a.K = p->K; a.Cs4 = p->Cs / 4; a.act = p->act; a.b_layout = p->b_layout; a.Zi = p->Zi;
    a.sA_o = p->sA_o; a.sA_i = p->sA_i; a.sB_o = p->sB_o; a.sB_i = p->sB_i; a.sC_o = p->sC_o; a.sC_i = p->sC_i;
    a.g = make_devgeom(p->g, p->lda);
    a.ntap_sel = p->ntap_sel;
    for (int i = 0; i < 28; ++i) a.tap_sel[i] = p->tap_sel[i];
    a.stats_part = p->stats_part; a.stats_relu = p->stats_relu; a.stats_rpg = 1; a.stats_chunks = 1;
    a.bs_x = p->bstats_x; a.bs_ldx = p->bstats_ldx; a.bs_y = p->bstats_y; a.bs_ldy = p->bstats_ldy; a.bs_mean = p->bstats_mean; a.bs_rstd = p->bstats_rstd;
    if (p->bstats_x) {
        if (!p->stats_part || !p->bstats_mean || !p->bstats_rstd || p->bstats_ldx < p->N || (p->bstats_y && p->bstats_ldy < p->N)) return CORRIF_EINVAL;
        if ((p->bstats_ldx & 3) || (p->bstats_ldy & 3) || ((uintptr_t)p->bstats_x & 15) || ((uintptr_t)p->bstats_y & 15)) return CORRIF_EUNSUPPORTED;
    }
    a.zs_bias = p->zs_bias; a.zs_add = p->zs_add; a.zs_add2 = p->zs_add2; a.zs_stats = p->zs_stats; a.zs_bsx = p->zs_bsx;
    a.zs_bsy = p->zs_bsy; a.zs_bsstat = p->zs_bsstat;
    if ((p->zs_add & 3) || (p->zs_add2 & 3) || (p->zs_bsx & 3) || (p->zs_bsy & 3)) return CORRIF_EUNSUPPORTED;
    if (p->stats_part) {
        if ((p->Z != 1 && (p->Zi != 1 || p->stats_rows_per_group != p->M)) || p->stats_rows_per_group <= 0 || p->stats_rows_per_group > p->M || (p->M % p->stats_rows_per_group)) return CORRIF_EINVAL;
        if (p->stats_rows_per_group != p->M && (p->stats_rows_per_group & 63)) return CORRIF_EUNSUPPORTED;
        if (scalar || p->out_map || p->N <= 16) return CORRIF_EUNSUPPORTED;     // only the TM = 2 tiles of gemm_fwd_kernel produce them
        a.stats_rpg = (int)p->stats_rows_per_group;
        a.stats_chunks = (int)((p->stats_rows_per_group + 63) / 64);
    }
    a.out_map = p->out_map; a.OD = p->OD; a.OH = p->OH; a.OW = p->OW;
    a.om_d = p->om_d; a.om_h = p->om_h; a.om_w = p->om_w; a.oo_d = p->oo_d; a.oo_h = p->oo_h; a.oo_w = p->oo_w;
    a.sk_ws = p->ws; a.sk_G = 0; a.sk_nk = 0; a.sk_tiles = 0; a.sk_tiles_mn = 0;
    a.sk_allowed = p->no_split ? 0 : 1;
    a.f32_mfma = p->f32_mfma ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const int Z = p->Z;
    if (scalar) return launch_fwd<128, 64, 2, 2, 1>(a, Z, s, plan_only, ws_bytes);
    if (p->ntap_sel || p->out_map) {       // parity-class data gradient: only the 32x32-MFMA tiles implement these options
        if (p->N <= 32) return launch_fwd<256, 32, 4, 1>(a, Z, s, plan_only, ws_bytes);
        if (p->N <= 64) return launch_fwd<128, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
        return launch_fwd<128, 128, 2, 2>(a, Z, s, plan_only, ws_bytes);
    }
    if (p->N <= 16 && !p->addend2 && !(p->N & 3) && !(p->ldc & 3) && !((uintptr_t)p->C & 15) && !(p->sC_o & 3) && !(p->sC_i & 3) &&
        (!p->addend || (!(p->ld_add & 3) && !((uintptr_t)p->addend & 15)))) {
        if (plan_only) return CORRIF_OK;
        return launch_smalln_fwd(a, Z, s);
    }
    if (p->N <= 32) return launch_fwd<256, 32, 4, 1>(a, Z, s, plan_only, ws_bytes);
    // 128-wide tiles measured faster than 64x64 even at ~1.5 workgroups per CU (e4: M=25088,N=256,K=2304: 70 vs 56 TF/s);
    // shrink only when the grid could not even cover the 256 CUs once.  (With the stream-K split the tile count no longer has to
    // cover the chip: the K loop is what is shared out.)
    auto tiles = [&](int bm, int bn) { return (int64_t)((p->M + bm - 1) / bm) * ((p->N + bn - 1) / bn) * p->Z; };
    const int nk = (p->K + BK - 1) / BK;
    const bool st = p->stats_part != nullptr;       // fused statistics need 64-row wave blocks (TM = 2)
    const bool splittable = a.sk_allowed != 0;
    if (p->N <= 64) {
        if (st || tiles(128, 64) >= 192 || (splittable && tiles(128, 64) * nk >= 2048)) return launch_fwd<128, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
        return launch_fwd<64, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
    }
    if (tiles(128, 128) >= 192 || (splittable && tiles(128, 128) * nk >= 2048)) return launch_fwd<128, 128, 2, 2>(a, Z, s, plan_only, ws_bytes);
    if (st || tiles(128, 64) >= 192) return launch_fwd<128, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
    return launch_fwd<64, 64, 2, 2>(a, Z, s, plan_only, ws_bytes);
}

extern "C" int corrif_gemm_fwd(const CorrifGemm* p, void* stream) { return gemm_fwd_impl(p, stream, false, nullptr); }

extern "C" int corrif_gemm_fwd_is_split(const CorrifGemm* p) {
    size_t n = 0;
    g_plan_split = 0;
    if (gemm_fwd_impl(p, nullptr, true, &n) != CORRIF_OK) return 0;
    return g_plan_split;
}

extern "C" size_t corrif_gemm_fwd_workspace(const CorrifGemm* p) {
    size_t n = 0;
    if (gemm_fwd_impl(p, nullptr, true, &n) != CORRIF_OK) return 0;
    return n;
}

