// TORCH_LIBRARY shim over the C-ABI of libcorrif_gfx950.so (include/corrif.h): the binding SURVEY section 8(b) / BASELINE north_star name.
//
// Scope: the NO-GRAD forward of the launch-heavy operator families - every nn.Conv3d of the path (mmvit4.py:32,72,120,131-135,161-168,
// 237-264,398-426; single and grouped = the three modality encoders' twins in one launch), nn.BatchNorm3d with running statistics
// (mmvit4.py:121,132-143; single and grouped), ReLU + nn.InstanceNorm3d (mmvit4.py:24,41-45) and nn.Linear (mmvit4.py:301,303,351,354).
// These are ~430 of the ~700 launches of an eval forward (F4_TRAIN.py:181-208, allJaccardResults_irem_f1_jcrd.py:201-222: batch 1, where
// the step is launch-bound).  An op validates its tensors, allocates outputs / re-laid weights with the caching allocator, fills the
// C-ABI descriptor IN C++ and enqueues on the current HIP stream - no Python between the nn.Module and the kernel.  The training path
// (autograd.Function objects with their cross-layer gradient links, ops.py) stays on the ctypes binding of the same C-ABI.
//
// Ops (namespace corrif): conv3d_fwd, conv3d_grouped_fwd, batch_norm_eval, batch_norm_grouped_eval, relu_instnorm_fwd, linear_fwd.
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <tuple>
#include <vector>

#include "../../include/corrif.h"

namespace {

void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }
void chk(int st, const char* what) { TORCH_CHECK(st == 0, "corrif: ", what, " failed with status ", st); }
const float* fp(const at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }
float* fpm(at::Tensor& t) { return t.defined() ? t.data_ptr<float>() : nullptr; }

// (t', rows, ld): last dim contiguous, every leading dim dense over a uniform row pitch ld, 16-byte aligned; a dense copy otherwise
// (ops.rows_view)
struct Rows { at::Tensor t; int64_t rows, ld; };
Rows rows_view(const at::Tensor& t0) {
    at::Tensor t = t0;
    const int64_t C = t.size(-1);
    bool ok = t.stride(-1) == 1 || C == 1;
    int64_t ld = -1;
    if (ok) {
        int64_t expect = -1;
        for (int64_t i = t.dim() - 2; i >= 0; --i) {
            if (t.size(i) == 1) continue;
            if (expect < 0) { ld = t.stride(i); expect = ld * t.size(i); }
            else { if (t.stride(i) != expect) { ok = false; break; } expect *= t.size(i); }
        }
        if (ld < 0) ld = C;
        if (ld < C || (ld & 3) || ((uintptr_t)t.data_ptr() & 15)) ok = false;
    }
    if (!ok) { t = t.contiguous(); ld = C; }
    return {t, t.numel() / C, ld};
}

at::Tensor empty_f32(at::IntArrayRef shape, const at::Tensor& like) { return at::empty(shape, like.options().dtype(at::kFloat)); }

CorrifGeom gemm_geom() {
    CorrifGeom g = {};
    g.is_gemm = 1; g.dir = 1; g.div_d = g.div_h = g.div_w = 1; g.kd = g.kh = g.kw = 1; g.ntaps = 1;
    return g;
}
CorrifGeom conv_geom(const int64_t (&R)[3], const int64_t (&S)[3], const int64_t (&k)[3], const int64_t (&stride)[3], const int64_t (&pad)[3],
                     bool clamp, int ntaps, int64_t src_batch_pitch) {
    CorrifGeom g = {};
    g.is_gemm = 0;
    g.Rd = (int)R[0]; g.Rh = (int)R[1]; g.Rw = (int)R[2];
    g.Sd = (int)S[0]; g.Sh = (int)S[1]; g.Sw = (int)S[2];
    g.kd = (int)k[0]; g.kh = (int)k[1]; g.kw = (int)k[2];
    g.mul_d = (int)stride[0]; g.mul_h = (int)stride[1]; g.mul_w = (int)stride[2];
    g.off_d = -(int)pad[0]; g.off_h = -(int)pad[1]; g.off_w = -(int)pad[2];
    g.div_d = g.div_h = g.div_w = 1;
    g.dir = 1;
    g.clamp = clamp ? 1 : 0;
    g.ntaps = ntaps;
    g.src_batch_pitch = src_batch_pitch;
    return g;
}
int64_t out_size(int64_t i, int64_t k, int64_t s, int64_t p) { return (i + 2 * p - k) / s + 1; }

at::Tensor repack(const at::Tensor& src, at::IntArrayRef shape_out, int O, int I, int T, int mode, int64_t ldo, bool zero) {
    at::Tensor out = empty_f32(shape_out, src);
    if (zero) chk(corrif_fill(out.data_ptr<float>(), out.numel(), 0.f, cur_stream()), "corrif_fill");
    chk(corrif_weight_repack(src.data_ptr<float>(), out.data_ptr<float>(), O, I, T, mode, ldo, cur_stream()), "corrif_weight_repack");
    return out;
}

void launch_gemm(CorrifGemm& g) {
    g.no_split = 1;
    g.f32_mfma = 0;
    chk(corrif_gemm_fwd(&g, cur_stream()), "corrif_gemm_fwd");
}

// ---------------------------------------------------------------------------------------------------------------- conv3d (one module)
// ConvFn.forward of ops.py without the autograd state.  stats_groups > 0: also produce the sum / sum-of-squares partials of the norm
// that follows (G groups of rows; relu: of max(y, 0)) when the kernel that runs the layer can; returns (out, part, chunks, rows per group)
// with an empty `part` when it cannot (the norm then takes its own statistics pass).
std::tuple<at::Tensor, at::Tensor, int64_t, int64_t> conv3d_fwd(const at::Tensor& x_in, const at::Tensor& weight, const c10::optional<at::Tensor>& bias_o,
                                                                at::IntArrayRef stride_, at::IntArrayRef pad_, bool replicate,
                                                                const c10::optional<at::Tensor>& out_o, int64_t stats_groups, bool stats_relu) {
    TORCH_CHECK(weight.dim() == 5 && weight.scalar_type() == at::kFloat && weight.is_contiguous(), "conv weight must be dense fp32 (O,I,kd,kh,kw)");
    TORCH_CHECK(x_in.is_cuda() && x_in.scalar_type() == at::kFloat, "fp32 device activations expected");
    const int64_t Co = weight.size(0), Ci = weight.size(1);
    const int64_t k[3] = {weight.size(2), weight.size(3), weight.size(4)};
    const int64_t stride[3] = {stride_[0], stride_[1], stride_[2]}, pad[3] = {pad_[0], pad_[1], pad_[2]};
    const int T = (int)(k[0] * k[1] * k[2]);
    const bool stem = Ci == 1;
    at::Tensor bias = bias_o.has_value() ? *bias_o : at::Tensor();
    at::Tensor x = x_in;
    int64_t B, S[3], lda, batch_pitch = 0;
    if (stem) {                                     // [B, D, H, W] strided view of the NCDHW input, one modality
        TORCH_CHECK(x.dim() == 4, "stem input must be [B, D, H, W]");
        B = x.size(0); S[0] = x.size(1); S[1] = x.size(2); S[2] = x.size(3);
        if (x.stride(3) != 1 || x.stride(2) != S[2] || x.stride(1) != S[1] * S[2]) x = x.contiguous();
        lda = 1; batch_pitch = x.stride(0);
    } else {
        TORCH_CHECK(x.dim() == 5 && x.size(4) == Ci, "activations must be channels-last [B, D, H, W, Ci]");
        Rows r = rows_view(x);
        x = r.t; lda = r.ld;
        B = x.size(0); S[0] = x.size(1); S[1] = x.size(2); S[2] = x.size(3);
    }
    const int64_t O[3] = {out_size(S[0], k[0], stride[0], pad[0]), out_size(S[1], k[1], stride[1], pad[1]), out_size(S[2], k[2], stride[2], pad[2])};
    const int64_t M = B * O[0] * O[1] * O[2];
    at::Tensor out = out_o.has_value() ? *out_o : empty_f32({B, O[0], O[1], O[2], Co}, x);
    Rows ro = rows_view(out);
    TORCH_CHECK(ro.t.is_same(out), "conv output slice must be row-addressable");
    const int64_t ldc = ro.ld;
    const bool unit = stride[0] == 1 && stride[1] == 1 && stride[2] == 1;
    const bool is_gemm = T == 1 && unit && !stem;
    at::Tensor part;
    int64_t chunks = 0, rpg = 0;
    void* s = cur_stream();

    if (stem && !bias.defined() && !replicate &&
        corrif_stem_supported((int)Co, (int)k[0], (int)k[1], (int)k[2], (int)stride[0], (int)stride[1], (int)stride[2], (int)pad[0], (int)pad[1], (int)pad[2])) {
        at::Tensor wp = repack(weight, {Co, 148}, (int)Co, 1, T, 0, 148, true);
        chk(corrif_stem_fwd(x.data_ptr<float>(), batch_pitch, wp.data_ptr<float>(), out.data_ptr<float>(), ldc, (int)B, (int)S[0], (int)S[1], (int)S[2], s),
            "corrif_stem_fwd");
    } else if (stem) {
        const int Kp = (T + 3) / 4 * 4;
        at::Tensor wp = repack(weight, {Co, Kp}, (int)Co, 1, T, 0, Kp, true);
        CorrifGemm g = {};
        g.A = x.data_ptr<float>(); g.lda = 1; g.Cs = 1; g.B = wp.data_ptr<float>(); g.ldb = Kp; g.b_layout = 0; g.C = out.data_ptr<float>(); g.ldc = ldc;
        g.bias = fp(bias); g.M = (int)M; g.N = (int)Co; g.K = Kp; g.Z = 1; g.Zi = 1;
        g.g = conv_geom(O, S, k, stride, pad, replicate, T, batch_pitch);
        launch_gemm(g);
    } else if (is_gemm && corrif_conv1x1_small_supported((int)Ci, (int)Co)) {
        chk(corrif_conv1x1_small_fwd(x.data_ptr<float>(), lda, weight.data_ptr<float>(), 0, fp(bias), out.data_ptr<float>(), ldc, M, (int)Ci, (int)Co, s),
            "corrif_conv1x1_small_fwd");
    } else if (T == 27 && unit && pad[0] == 1 && pad[1] == 1 && pad[2] == 1 && corrif_conv3_patch_cc((int)Ci, (int)Co)) {
        const int cc = corrif_conv3_patch_cc((int)Ci, (int)Co);
        at::Tensor wp = repack(weight, {Ci / cc, Co, T, cc}, (int)Co, (int)Ci, T, 3, cc, false);
        CorrifConv3Patch q = {};
        q.X = x.data_ptr<float>(); q.ldx = lda; q.Wp = wp.data_ptr<float>(); q.Y = out.data_ptr<float>(); q.ldy = ldc; q.bias = fp(bias);
        q.B = (int)B; q.Sd = (int)S[0]; q.Sh = (int)S[1]; q.Sw = (int)S[2]; q.Od = (int)O[0]; q.Oh = (int)O[1]; q.Ow = (int)O[2];
        q.Ci = (int)Ci; q.Co = (int)Co; q.pad = 1; q.clamp = replicate ? 1 : 0; q.cc = cc;
        if (stats_groups == B && stats_relu && corrif_conv3_patch_stats_supported((int)Ci, (int)Co)) {
            chunks = corrif_conv3_patch_stats_chunks((int)B, (int)O[0], (int)O[1], (int)O[2]);
            part = at::empty({B * Co * chunks * 2}, x.options().dtype(at::kDouble));
            chk(corrif_fill(reinterpret_cast<float*>(part.data_ptr<double>()), 2 * part.numel(), 0.f, s), "corrif_fill");
            q.stats_part = part.data_ptr<double>(); q.stats_chunks = (int)chunks; q.stats_relu = 1;
            rpg = O[0] * O[1] * O[2];
        }
        chk(corrif_conv3_patch(&q, s), "corrif_conv3_patch");
    } else {
        at::Tensor wp = T == 1 ? weight : repack(weight, {Co, (int64_t)T * Ci}, (int)Co, (int)Ci, T, 0, (int64_t)T * Ci, false);
        CorrifGemm g = {};
        g.A = x.data_ptr<float>(); g.lda = lda; g.Cs = (int)Ci; g.B = wp.data_ptr<float>(); g.ldb = (int64_t)T * Ci; g.b_layout = 0;
        g.C = out.data_ptr<float>(); g.ldc = ldc; g.bias = fp(bias); g.M = (int)M; g.N = (int)Co; g.K = (int)(T * Ci); g.Z = 1; g.Zi = 1;
        g.g = is_gemm ? gemm_geom() : conv_geom(O, S, k, stride, pad, replicate, T, 0);
        if (stats_groups > 0 && Co > 16) {
            const int64_t G = stats_groups, per = M / G;
            if (M % G == 0 && (G == 1 || per % 64 == 0)) {
                chunks = (per + 63) / 64; rpg = per;
                part = at::empty({G * Co * chunks * 2}, x.options().dtype(at::kDouble));
                g.stats_part = part.data_ptr<double>(); g.stats_rows_per_group = per; g.stats_relu = stats_relu ? 1 : 0;
            }
        }
        launch_gemm(g);
    }
    if (!part.defined()) part = at::empty({0}, x.options().dtype(at::kDouble));
    return {out, part, chunks, rpg};
}

// ---------------------------------------------------------------------------------------------------------------- conv3d (G twins, one launch)
at::Tensor stack_groups(at::TensorList ts) {
    const int G = (int)ts.size();
    TORCH_CHECK(G >= 1 && G <= 4, "1..4 groups");
    std::vector<at::Tensor> keep;
    const float* ptrs[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < G; ++i) { keep.push_back(ts[i].contiguous()); ptrs[i] = keep[i].data_ptr<float>(); }
    std::vector<int64_t> shape = {G};
    for (auto d : keep[0].sizes()) shape.push_back(d);
    at::Tensor out = empty_f32(shape, keep[0]);
    chk(corrif_stack_groups(ptrs, G, out.data_ptr<float>(), keep[0].numel(), cur_stream()), "corrif_stack_groups");
    return out;
}

// GroupedConvFn.forward of ops.py: zin / zout 0 = groups stacked along the batch axis, 1 = groups side by side in a concat buffer
at::Tensor conv3d_grouped_fwd(const at::Tensor& x_in, at::TensorList weights, at::TensorList biases, at::IntArrayRef stride_, at::IntArrayRef pad_,
                              int64_t zin, int64_t zout, const c10::optional<at::Tensor>& out_o) {
    const int64_t G = (int64_t)weights.size();
    at::Tensor w = stack_groups(weights);
    at::Tensor b = biases.size() ? stack_groups(biases) : at::Tensor();
    const int64_t Co = w.size(1), Ci = w.size(2);
    const int64_t k[3] = {w.size(3), w.size(4), w.size(5)};
    const int64_t stride[3] = {stride_[0], stride_[1], stride_[2]}, pad[3] = {pad_[0], pad_[1], pad_[2]};
    const int T = (int)(k[0] * k[1] * k[2]);
    Rows r = rows_view(x_in);
    at::Tensor x = r.t;
    const int64_t lda = r.ld;
    int64_t B, zA;
    const int64_t S[3] = {x.size(1), x.size(2), x.size(3)};
    const bool unit = stride[0] == 1 && stride[1] == 1 && stride[2] == 1;
    if (zin == 0) {
        B = x.size(0) / G;
        TORCH_CHECK(x.size(0) == G * B && x.size(4) == Ci && lda == Ci, "stacked input [G*B, D, H, W, Ci] expected");
        zA = B * S[0] * S[1] * S[2] * lda;
    } else {
        B = x.size(0);
        TORCH_CHECK(x.size(4) == G * Ci && T == 1 && unit, "concat-layout input needs a 1x1x1 stride-1 convolution");
        zA = Ci;
    }
    const int64_t O[3] = {out_size(S[0], k[0], stride[0], pad[0]), out_size(S[1], k[1], stride[1], pad[1]), out_size(S[2], k[2], stride[2], pad[2])};
    const int64_t M = B * O[0] * O[1] * O[2];
    at::Tensor out = out_o.has_value() ? *out_o
                                       : (zout == 0 ? empty_f32({G * B, O[0], O[1], O[2], Co}, x) : empty_f32({B, O[0], O[1], O[2], G * Co}, x));
    Rows ro = rows_view(out);
    TORCH_CHECK(ro.t.is_same(out), "conv output must be row-addressable");
    const int64_t ldc = ro.ld, zC = zout == 0 ? M * ldc : Co;
    const bool is_gemm = T == 1 && unit;
    at::Tensor wp = T == 1 ? w : repack(w, {G, Co, (int64_t)T * Ci}, (int)(G * Co), (int)Ci, T, 0, (int64_t)T * Ci, false);
    CorrifGemm g = {};
    g.A = x.data_ptr<float>(); g.lda = lda; g.Cs = (int)Ci; g.B = wp.data_ptr<float>(); g.ldb = (int64_t)T * Ci; g.b_layout = 0;
    g.C = out.data_ptr<float>(); g.ldc = ldc; g.bias = fp(b); g.M = (int)M; g.N = (int)Co; g.K = (int)(T * Ci);
    g.Z = (int)G; g.Zi = 1; g.sA_o = zA; g.sB_o = Co * T * Ci; g.sC_o = zC; g.zs_bias = Co;
    g.g = is_gemm ? gemm_geom() : conv_geom(O, S, k, stride, pad, false, T, 0);
    launch_gemm(g);
    return out;
}

// ---------------------------------------------------------------------------------------------------------------- BatchNorm3d, running statistics
at::Tensor batch_norm_eval(const at::Tensor& x_in, const at::Tensor& gamma, const at::Tensor& beta, const at::Tensor& running_mean,
                           const at::Tensor& running_var, const c10::optional<at::Tensor>& residual_o, int64_t flags, double eps,
                           const c10::optional<at::Tensor>& out_o) {
    Rows r = rows_view(x_in);
    const int64_t C = r.t.size(-1);
    at::Tensor rstd = empty_f32({C}, r.t);
    void* s = cur_stream();
    chk(corrif_norm_eval_rstd(running_var.data_ptr<float>(), (float)eps, rstd.data_ptr<float>(), (int)C, s), "corrif_norm_eval_rstd");
    at::Tensor res;
    int64_t ldr = 0;
    if (residual_o.has_value()) { Rows rr = rows_view(*residual_o); res = rr.t; ldr = rr.ld; }
    at::Tensor out = out_o.has_value() ? *out_o : empty_f32(r.t.sizes(), r.t);
    Rows ro = rows_view(out);
    TORCH_CHECK(ro.t.is_same(out), "norm output must be row-addressable");
    chk(corrif_norm_apply(r.t.data_ptr<float>(), r.ld, running_mean.data_ptr<float>(), rstd.data_ptr<float>(), gamma.data_ptr<float>(), beta.data_ptr<float>(),
                          fp(res), ldr, out.data_ptr<float>(), ro.ld, r.rows, 1, (int)C, (int)flags, s), "corrif_norm_apply");
    return out;
}

at::Tensor batch_norm_grouped_eval(const at::Tensor& x_in, at::TensorList gammas, at::TensorList betas, at::TensorList running_means,
                                   at::TensorList running_vars, const c10::optional<at::Tensor>& residual_o, int64_t flags, double eps) {
    const int64_t G = (int64_t)gammas.size();
    Rows r = rows_view(x_in);
    const int64_t C = r.t.size(-1), rpg = r.rows / G;
    at::Tensor gamma = stack_groups(gammas), beta = stack_groups(betas), mean = stack_groups(running_means), var = stack_groups(running_vars);
    at::Tensor rstd = empty_f32({G, C}, r.t);
    void* s = cur_stream();
    chk(corrif_norm_eval_rstd(var.data_ptr<float>(), (float)eps, rstd.data_ptr<float>(), (int)(G * C), s), "corrif_norm_eval_rstd");
    at::Tensor res;
    int64_t ldr = 0;
    if (residual_o.has_value()) { Rows rr = rows_view(*residual_o); res = rr.t; ldr = rr.ld; }
    at::Tensor out = empty_f32(r.t.sizes(), r.t);
    chk(corrif_norm_apply_g(r.t.data_ptr<float>(), r.ld, mean.data_ptr<float>(), rstd.data_ptr<float>(), gamma.data_ptr<float>(), beta.data_ptr<float>(),
                            fp(res), ldr, out.data_ptr<float>(), C, rpg, (int)G, (int)C, (int)flags, C, s), "corrif_norm_apply_g");
    return out;
}

// ---------------------------------------------------------------------------------------------------------------- ReLU + InstanceNorm3d
at::Tensor relu_instnorm_fwd(const at::Tensor& x_in, double eps, const c10::optional<at::Tensor>& out_o, const at::Tensor& part, int64_t chunks,
                             int64_t part_rpg) {
    Rows r = rows_view(x_in);
    const int64_t B = r.t.size(0), C = r.t.size(-1), rpg = r.rows / B;
    at::Tensor mean = empty_f32({B * C}, r.t), rstd = empty_f32({B * C}, r.t);
    void* s = cur_stream();
    if (part.numel() > 0 && part_rpg == rpg) {
        chk(corrif_norm_stats_finalize(part.data_ptr<double>(), (int)chunks, (int)B, (int)C, rpg, (float)eps, mean.data_ptr<float>(), rstd.data_ptr<float>(),
                                       nullptr, nullptr, 0.f, s), "corrif_norm_stats_finalize");
    } else {
        at::Tensor ws = at::empty({(int64_t)corrif_norm_workspace(rpg, (int)B, (int)C)}, r.t.options().dtype(at::kByte));
        chk(corrif_norm_stats(r.t.data_ptr<float>(), r.ld, rpg, (int)B, (int)C, CORRIF_NORM_RELU_IN, (float)eps, mean.data_ptr<float>(), rstd.data_ptr<float>(),
                              nullptr, nullptr, 0.f, reinterpret_cast<double*>(ws.data_ptr()), s), "corrif_norm_stats");
    }
    at::Tensor out = out_o.has_value() ? *out_o : empty_f32(r.t.sizes(), r.t);
    Rows ro = rows_view(out);
    TORCH_CHECK(ro.t.is_same(out), "norm output must be row-addressable");
    chk(corrif_norm_apply(r.t.data_ptr<float>(), r.ld, mean.data_ptr<float>(), rstd.data_ptr<float>(), nullptr, nullptr, nullptr, 0, out.data_ptr<float>(), ro.ld,
                          rpg, (int)B, (int)C, CORRIF_NORM_RELU_IN, s), "corrif_norm_apply");
    return out;
}

// ---------------------------------------------------------------------------------------------------------------- nn.Linear
at::Tensor linear_fwd(const at::Tensor& x_in, const at::Tensor& weight, const c10::optional<at::Tensor>& bias_o) {
    Rows r = rows_view(x_in);
    const int64_t N = weight.size(0), K = weight.size(1);
    std::vector<int64_t> shape(r.t.sizes().begin(), r.t.sizes().end());
    shape.back() = N;
    at::Tensor y = empty_f32(shape, r.t);
    CorrifGemm g = {};
    g.A = r.t.data_ptr<float>(); g.lda = r.ld; g.Cs = (int)K; g.B = weight.data_ptr<float>(); g.ldb = K; g.b_layout = 0; g.C = y.data_ptr<float>(); g.ldc = N;
    g.bias = bias_o.has_value() ? bias_o->data_ptr<float>() : nullptr;
    g.M = (int)r.rows; g.N = (int)N; g.K = (int)K; g.Z = 1; g.Zi = 1; g.g = gemm_geom();
    launch_gemm(g);
    return y;
}

}  // namespace

TORCH_LIBRARY(corrif, m) {
    m.def("conv3d_fwd(Tensor x, Tensor weight, Tensor? bias, int[] stride, int[] pad, bool replicate, Tensor? out, int stats_groups, bool stats_relu)"
          " -> (Tensor, Tensor, int, int)");
    m.def("conv3d_grouped_fwd(Tensor x, Tensor[] weights, Tensor[] biases, int[] stride, int[] pad, int zin, int zout, Tensor? out) -> Tensor");
    m.def("batch_norm_eval(Tensor x, Tensor gamma, Tensor beta, Tensor running_mean, Tensor running_var, Tensor? residual, int flags, float eps,"
          " Tensor? out) -> Tensor");
    m.def("batch_norm_grouped_eval(Tensor x, Tensor[] gammas, Tensor[] betas, Tensor[] running_means, Tensor[] running_vars, Tensor? residual,"
          " int flags, float eps) -> Tensor");
    m.def("relu_instnorm_fwd(Tensor x, float eps, Tensor? out, Tensor part, int chunks, int part_rpg) -> Tensor");
    m.def("linear_fwd(Tensor x, Tensor weight, Tensor? bias) -> Tensor");
}

TORCH_LIBRARY_IMPL(corrif, CUDA, m) {
    m.impl("conv3d_fwd", conv3d_fwd);
    m.impl("conv3d_grouped_fwd", conv3d_grouped_fwd);
    m.impl("batch_norm_eval", batch_norm_eval);
    m.impl("batch_norm_grouped_eval", batch_norm_grouped_eval);
    m.impl("relu_instnorm_fwd", relu_instnorm_fwd);
    m.impl("linear_fwd", linear_fwd);
}
