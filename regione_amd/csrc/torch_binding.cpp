// torch_binding.cpp - C++ registration of the hot-path op surface: TORCH_LIBRARY(regione_mi, m) + TORCH_LIBRARY_IMPL(regione_mi,
// CUDA, m) over the C ABI of include/regione_hip.h ("C++ TORCH_LIBRARY(regione_mi, m) ops, HIP
// implementations, hipStream_t = PyTorch current stream, tensors borrowed, caches mutated in place (Tensor(a!)), errors via
// TORCH_CHECK").  Built into regione_amd/lib/libregione_torch.so (regione_amd/build.py), linked against libregione_hip.so:
//
//     torch.ops.load_library(".../regione_amd/lib/libregione_torch.so")      # no Python shim: C++ / AOT callers
//     e, u, mask = torch.ops.regione_mi.arp_partition(sample, v, cond, dt_final, 0.88, 64, 64, True)
//
// The schemas are the ones regione_amd/torch_ops.py defines through torch.library (that registration stays as the A/B:
// RGN_TORCH_OPS=py | cpp); the two cannot be loaded into one process (same operator names).  Nothing here computes: every op
// validates, allocates its outputs with torch (device memory is torch's), and forwards device pointers + the current stream.
//
// | op                       | replaces (reference file:line)                                                        |
// | arp_partition            | token_selector + one-step estimate, FluxKontext/utils.py:282-354, inplace.py:650-651   |
// | gather_rows / scatter_rows_ | ids_gather / ids_scatter, utils.py:260-279 / :240-257                               |
// | split_euler_step         | scheduler update incl. the split update, inplace.py:648-680                            |
// | avd_apply                | cache = ids_gather(cache, ids); noise_pred = cache * ratio, inplace.py:315-318          |
// | cfg_combine              | inplace.py:364; Step1XEdit/inplace.py:401-410; QwenImageEdit/inplace.py:401-405         |
// | kv_partial_update_[pair_|group_] | _partially_linear x2 + norm_k + RoPE into the caches, inplace.py:734-794, fused_kernels.py:81-101 |
// | region_attention         | flash_attn_func / SDPA of the edited-token queries vs the full cache, inplace.py:796-806 |
#include <ATen/ATen.h>
#include <c10/core/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cmath>
#include <list>
#include <mutex>
#include <optional>
#include <tuple>
#include <vector>

#include "../../include/regione_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == 0, what, " failed (rc ", rc, "): ", rgn_last_error());
}

// every op runs on the device of its first tensor argument (a process that drives several GPUs may call with another device current)
// (c10::DeviceGuard, not c10::hip::HIPGuard: PyTorch-ROCm tensors carry DeviceType "cuda" and the plain HIP guard refuses them)
#define RGN_DEVICE_GUARD(t) const c10::DeviceGuard rgn_device_guard((t).device())

void* stream_of(const Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

int dt(const Tensor& t) {
    if (t.scalar_type() == at::kFloat) return RGN_F32;
    if (t.scalar_type() == at::kBFloat16) return RGN_BF16;
    TORCH_CHECK(false, "regione_mi: unsupported dtype ", t.scalar_type(), " (fp32 / bf16 only)");
}

const void* ptr(const Tensor& t) {
    TORCH_CHECK(t.is_cuda(), "regione_mi ops need CUDA/HIP tensors: there is no CPU fallback");
    return t.data_ptr();
}
const void* ptr(const OptTensor& t) { return (t.has_value() && t->defined()) ? ptr(*t) : nullptr; }

// [1, L, D] or [L, D] -> contiguous [L, D] view (the reference is batch-1, quirk A-5)
Tensor rows(const Tensor& t) {
    Tensor r = t;
    if (t.dim() == 3) {
        TORCH_CHECK(t.size(0) == 1, "region ops are per image (batch 1)");
        r = t[0];
    }
    TORCH_CHECK(r.dim() == 2 && r.is_contiguous(), "region ops take contiguous [L, D] rows");
    return r;
}

Tensor like_input(const Tensor& out2d, const Tensor& in) { return in.dim() == 3 ? out2d.unsqueeze(0) : out2d; }

// fp32 scratch for the round-aware schedules: one buffer per (kind, device, stream), bounded LRU (two forwards on two streams
// must not share split-K / KV-split partials) - the same policy as regione_amd/ops.py:_ws_lane
struct Lane { int kind; int dev; void* st; Tensor buf; };
std::mutex g_ws_mu;
std::list<Lane> g_ws;
Tensor workspace(int kind, const Tensor& like, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ws_mu);
    const int dev = like.get_device();
    void* st = stream_of(like);
    int same_kind = 0;
    for (auto it = g_ws.begin(); it != g_ws.end(); ++it) {
        if (it->kind == kind && it->dev == dev && it->st == st) {
            g_ws.splice(g_ws.end(), g_ws, it);                 // most recently used last
            return g_ws.back().buf;
        }
        same_kind += (it->kind == kind && it->dev == dev);
    }
    if (same_kind >= 4)          // four lanes per kind AND device (a process driving several GPUs keeps each device's lanes)
        for (auto it = g_ws.begin(); it != g_ws.end(); ++it)
            if (it->kind == kind && it->dev == dev) { g_ws.erase(it); break; }
    g_ws.push_back({kind, dev, st, at::empty({(int64_t)(bytes / 4)}, like.options().dtype(at::kFloat))});
    return g_ws.back().buf;
}
Tensor gemm_ws(const Tensor& like) { return workspace(0, like, rgn_gemm_workspace_bytes()); }
Tensor attn_ws(const Tensor& like) { return workspace(1, like, rgn_attention_workspace_bytes(0, 0)); }

// the lane of (kind, device of `like`, its current stream) as a tensor: regione_amd.ops borrows it for the block-body GEMMs it launches
// through ctypes, so that ONE 256 MiB split-K scratch serves a stream whichever way a launch reaches the library (advisor, round 4)
Tensor workspace_op(const Tensor& like, int64_t kind) {
    TORCH_CHECK(kind == 0 || kind == 1, "workspace: kind 0 (GEMM split-K) or 1 (attention KV split)");
    RGN_DEVICE_GUARD(like);
    return kind == 0 ? gemm_ws(like) : attn_ws(like);
}

// ---- region ops -------------------------------------------------------------------------------------------------------------
std::tuple<Tensor, Tensor, Tensor> arp_partition(const Tensor& sample, const OptTensor& model_output, const Tensor& cond,
                                                 double dt_final, double threshold, int64_t h_tok, int64_t w_tok,
                                                 bool erosion_dilation) {
    RGN_DEVICE_GUARD(sample);
    Tensor s = rows(sample), c = rows(cond);
    OptTensor mo;
    if (model_output.has_value() && model_output->defined()) mo = rows(*model_output);
    const int64_t L = s.size(0), D = s.size(1);
    TORCH_CHECK(c.size(0) == L && c.size(1) == D && h_tok * w_tok == L && c.device() == s.device(), "arp_partition: sample / cond [L, D] with L = h_tok * w_tok, one device");
    TORCH_CHECK(!mo || (mo->size(0) == L && mo->size(1) == D && mo->device() == s.device()), "arp_partition: model_output must be [L, D] like the sample");
    auto i64 = s.options().dtype(at::kLong), u8 = s.options().dtype(at::kByte);
    Tensor e = at::empty({L}, i64), u = at::empty({L}, i64), raw = at::empty({L}, u8), mask = at::empty({L}, u8);
    Tensor cnt = at::empty({1}, s.options().dtype(at::kInt));
    check_rc(rgn_arp_partition(ptr(s), dt(s), ptr(mo), mo ? dt(*mo) : RGN_F32, ptr(c), dt(c), (float)dt_final, (float)threshold,
                               (int)L, (int)D, (int)h_tok, (int)w_tok, (int)erosion_dilation, (int64_t*)e.data_ptr(),
                               (int64_t*)u.data_ptr(), (uint8_t*)raw.data_ptr(), (uint8_t*)mask.data_ptr(), nullptr,
                               (int32_t*)cnt.data_ptr(), stream_of(s)),
             "rgn_arp_partition");
    const int64_t k = cnt.item<int32_t>();                    // the ONE host sync of an edit: K_e (4 bytes)
    return {e.narrow(0, 0, k).unsqueeze(0), u.narrow(0, 0, L - k).unsqueeze(0), mask};
}

Tensor gather_rows(const Tensor& x, const Tensor& ids) {
    RGN_DEVICE_GUARD(x);
    Tensor src = rows(x), idv = ids.reshape({-1}).contiguous();
    TORCH_CHECK(idv.scalar_type() == at::kLong && idv.device() == src.device(), "gather_rows: ids must be int64 on the source's device");
    Tensor out = at::empty({idv.numel(), src.size(1)}, src.options());
    check_rc(rgn_gather_rows(ptr(src), (const int64_t*)ptr(idv), out.data_ptr(), (int)idv.numel(),
                             (int)(src.size(1) * src.element_size()), stream_of(src)), "rgn_gather_rows");
    return like_input(out, x);
}

void scatter_rows_(const Tensor& src, const Tensor& ids, Tensor dst) {
    RGN_DEVICE_GUARD(src);
    Tensor s = rows(src), d = rows(dst), idv = ids.reshape({-1}).contiguous();
    TORCH_CHECK(idv.scalar_type() == at::kLong && s.size(1) == d.size(1) && s.scalar_type() == d.scalar_type() && idv.device() == s.device() &&
                    d.device() == s.device(), "scatter_rows_: int64 ids, equal row width / dtype, one device");
    TORCH_CHECK(idv.numel() <= s.size(0), "scatter_rows_: ", idv.numel(), " ids but the source has ", s.size(0), " rows");
    check_rc(rgn_scatter_rows(ptr(s), (const int64_t*)ptr(idv), d.data_ptr(), (int)idv.numel(), (int)(s.size(1) * s.element_size()),
                              stream_of(s)), "rgn_scatter_rows");
}

Tensor split_euler_step(const Tensor& sample, const Tensor& v, double dt_, const OptTensor& mask, double dt_direct) {
    RGN_DEVICE_GUARD(sample);
    Tensor s = rows(sample), vv = rows(v);
    TORCH_CHECK(s.sizes() == vv.sizes(), "split_euler_step: shapes");
    TORCH_CHECK(!(mask.has_value() && mask->defined()) ||
                    (mask->scalar_type() == at::kByte && mask->is_contiguous() && mask->numel() == s.size(0)),
                "split_euler_step: mask must be uint8 [L] (the third result of arp_partition)");
    Tensor out = at::empty_like(vv);
    check_rc(rgn_euler_step(ptr(s), dt(s), ptr(vv), dt(vv), out.data_ptr(), (const uint8_t*)ptr(mask), (float)dt_, (float)dt_direct,
                            (int)s.size(0), (int)s.size(1), stream_of(s)), "rgn_euler_step");
    return like_input(out, v);
}

Tensor avd_apply(const Tensor& cache, double ratio, const OptTensor& ids, bool round_ratio) {
    RGN_DEVICE_GUARD(cache);
    Tensor c = rows(cache);
    OptTensor idv;
    if (ids.has_value() && ids->defined()) {
        TORCH_CHECK(ids->scalar_type() == at::kLong, "avd_apply: ids must be int64");   // read as 8-byte indices
        idv = ids->reshape({-1}).contiguous();
    }
    const int64_t K = idv ? idv->numel() : c.size(0);
    Tensor out = at::empty({K, c.size(1)}, c.options());
    check_rc(rgn_avd_apply(ptr(c), dt(c), (const int64_t*)ptr(idv), (float)ratio, (int)round_ratio, out.data_ptr(), (int)K,
                           (int)c.size(1), stream_of(c)), "rgn_avd_apply");
    return like_input(out, cache);
}

Tensor cfg_combine(const Tensor& pos, const Tensor& neg, double scale, int64_t mode, double power) {
    RGN_DEVICE_GUARD(pos);
    Tensor p = rows(pos), n = rows(neg);
    TORCH_CHECK(p.sizes() == n.sizes() && p.scalar_type() == n.scalar_type(), "cfg_combine: shapes");
    Tensor out = at::empty_like(p);
    check_rc(rgn_cfg_combine(ptr(p), ptr(n), out.data_ptr(), dt(p), (float)scale, (int)mode, (float)power, (int)p.size(0),
                             (int)p.size(1), stream_of(p)), "rgn_cfg_combine");
    return like_input(out, pos);
}

// ---- Region-Instruction KV cache -------------------------------------------------------------------------------------------
// `x` = the activations of the problem this epilogue belongs to: its rows are sequence rows [row_base, row_base + M), every tensor of
// the descriptor must live on its device and cover those rows (a C++ / AOT caller gets a TORCH_CHECK failure, never an
// out-of-bounds device read or K / V scattered into arbitrary cache rows - advisor finding, round 4)
rgn_qkv_epilogue epi(const Tensor& x, const Tensor& norm_q, const Tensor& norm_k, const Tensor& cos_q, const Tensor& sin_q, const Tensor& cos_k,
                     const Tensor& sin_k, const OptTensor& kv_rows, const Tensor& k_cache, const Tensor& vt_cache, int64_t heads,
                     int64_t row_base, double eps, bool fp16_roundtrip) {
    TORCH_CHECK(heads > 0 && k_cache.dim() == 2, "K slab [skv_pad, H*128]");
    const int64_t d = heads * 128, skv_pad = k_cache.size(0), M = x.size(0);
    const bool have_rows = kv_rows.has_value() && kv_rows->defined();
    TORCH_CHECK(row_base >= 0, "row_base must be >= 0");
    for (const Tensor* t : {&norm_q, &norm_k, &cos_q, &sin_q, &cos_k, &sin_k, &k_cache, &vt_cache})
        TORCH_CHECK(t->device() == x.device(), "fused Q/K/V epilogue: every tensor must live on the activations' device ", x.device());
    TORCH_CHECK(!have_rows || kv_rows->device() == x.device(), "kv_rows must live on the activations' device");
    TORCH_CHECK(k_cache.dim() == 2 && k_cache.size(1) == d && vt_cache.dim() == 2 && vt_cache.size(0) == d && vt_cache.size(1) == skv_pad &&
                    k_cache.is_contiguous() && vt_cache.is_contiguous(), "K slab [skv_pad, H*128] / V^T slab [H*128, skv_pad], contiguous");
    for (const Tensor* t : {&norm_q, &norm_k})
        TORCH_CHECK(t->scalar_type() == at::kBFloat16 && t->numel() == 128 && t->is_contiguous(), "per-head RMSNorm weights: bf16 [128]");
    TORCH_CHECK(cos_q.sizes() == sin_q.sizes() && cos_k.sizes() == sin_k.sizes(), "rotary tables: cos / sin of one table differ in shape");
    for (const Tensor* t : {&cos_q, &sin_q, &cos_k, &sin_k})
        TORCH_CHECK(t->scalar_type() == at::kFloat && t->dim() == 2 && t->size(1) == 128 && t->is_contiguous(), "rotary tables: fp32 [rows, 128]");
    // the kernel reads 8-byte indices: an int32 tensor would be read past its end and scatter K / V to garbage cache rows
    TORCH_CHECK(!(kv_rows.has_value() && kv_rows->defined()) ||
                    (kv_rows->scalar_type() == at::kLong && kv_rows->dim() == 1 && kv_rows->is_contiguous()), "kv_rows: int64 [rows]");
    // extents: the kernel reads cos_q / sin_q at sequence rows [row_base, row_base + M), kv_rows[row_base + m] for every row, and
    // cos_k / sin_k at the CACHE row of each sequence row (identity rows: the same range; gathered rows: see below)
    TORCH_CHECK(cos_q.size(0) >= row_base + M, "rotary table of the queries has ", cos_q.size(0), " rows, the problem needs ", row_base + M);
    if (have_rows) {
        TORCH_CHECK(kv_rows->numel() >= row_base + M, "kv_rows has ", kv_rows->numel(), " entries, the problem needs ", row_base + M);
        // gathered cache rows: the kernel reads cos_k / sin_k at row kv_rows[row_base + m] and writes slab row kv_rows[row_base + m].  The
        // VALUES are device data and are NOT checked here (that would be a host sync per launch): the caller guarantees
        // max(kv_rows) < min(cos_k rows, skv_pad).  What can be checked is the necessary condition: the rows are distinct scatter
        // targets, so the table and the slab must hold at least as many rows as the problem names.
        TORCH_CHECK(cos_k.size(0) >= row_base + M && skv_pad >= row_base + M, "rotary table of the keys (", cos_k.size(0), " rows) / the slab (",
                    skv_pad, " rows) cannot hold the ", row_base + M, " distinct cache rows kv_rows names");
    } else {
        TORCH_CHECK(cos_k.size(0) >= row_base + M, "rotary table of the keys has ", cos_k.size(0), " rows, the problem needs ", row_base + M);
        TORCH_CHECK(row_base + M <= skv_pad, "identity cache rows [", row_base, ", ", row_base + M, ") exceed the slab's ", skv_pad, " rows");
    }
    rgn_qkv_epilogue e;
    e.wq = ptr(norm_q); e.wk = ptr(norm_k);
    e.cos_q = (const float*)ptr(cos_q); e.sin_q = (const float*)ptr(sin_q);
    e.cos_k = (const float*)ptr(cos_k); e.sin_k = (const float*)ptr(sin_k);
    e.kv_rows = (const int64_t*)ptr(kv_rows);
    e.k_slab = k_cache.data_ptr(); e.vt_slab = vt_cache.data_ptr();
    e.row_base = (int)row_base; e.skv_pad = (int)skv_pad; e.k_col = 0; e.v_col = (int)d; e.q_col = (int)(2 * d); e.heads = (int)heads;
    e.eps = (float)eps; e.fp16_roundtrip = fp16_roundtrip ? 1 : 0;
    return e;
}

bool is_fp8(const Tensor& w) { return w.scalar_type() == at::kFloat8_e4m3fn; }

void check_act(const Tensor& x, const Tensor& w, const Tensor& out, const OptTensor& wscale, const OptTensor& bias = OptTensor()) {
    TORCH_CHECK(w.device() == x.device() && out.device() == x.device(), "projection: activations, weights and output on one device");
    if (bias.has_value() && bias->defined())
        TORCH_CHECK(bias->scalar_type() == at::kBFloat16 && bias->dim() == 1 && bias->numel() == w.size(0) && bias->is_contiguous() &&
                        bias->device() == x.device(), "projection: bias must be bf16 [N], contiguous, on the activations' device");
    if (wscale.has_value() && wscale->defined())
        TORCH_CHECK(wscale->is_contiguous() && wscale->device() == x.device(), "w_scale must be contiguous and on the activations' device");
    TORCH_CHECK(x.scalar_type() == at::kBFloat16 && out.scalar_type() == at::kBFloat16 && (w.scalar_type() == at::kBFloat16 || is_fp8(w)),
                "projection: bf16 activations, bf16 or fp8 (e4m3fn) weights");
    TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && out.dim() == 2 && x.stride(1) == 1 && w.stride(1) == 1 && out.stride(1) == 1 &&
                    w.size(1) == x.size(1) && out.size(0) == x.size(0) && out.size(1) == w.size(0), "projection: shapes / strides");
    TORCH_CHECK(!is_fp8(w) || (wscale.has_value() && wscale->defined() && wscale->scalar_type() == at::kFloat && wscale->numel() == w.size(0)),
                "fp8 weight without its per-output-channel fp32 scale (w_scale)");
}

void kv_partial_update_(const Tensor& x, const Tensor& w_kvq, const OptTensor& b_kvq, Tensor q_out, const Tensor& norm_q,
                        const Tensor& norm_k, const Tensor& cos_q, const Tensor& sin_q, const Tensor& cos_k, const Tensor& sin_k,
                        const OptTensor& kv_rows, Tensor k_cache, Tensor vt_cache, int64_t heads, int64_t row_base, double eps,
                        bool fp16_roundtrip, int64_t gelu_from_col, const OptTensor& w_scale) {
    RGN_DEVICE_GUARD(x);
    check_act(x, w_kvq, q_out, w_scale, b_kvq);
    rgn_qkv_epilogue e = epi(x, norm_q, norm_k, cos_q, sin_q, cos_k, sin_k, kv_rows, k_cache, vt_cache, heads, row_base, eps, fp16_roundtrip);
    const int gelu = (int)(gelu_from_col < 0 ? 3 * heads * 128 : gelu_from_col);
    Tensor ws = gemm_ws(x);
    const int M = (int)x.size(0), N = (int)w_kvq.size(0), K = (int)x.size(1);
    if (is_fp8(w_kvq))
        check_rc(rgn_gemm_w8_qkv(ptr(x), (int)x.stride(0), ptr(w_kvq), (int)w_kvq.stride(0), (const float*)ptr(w_scale), ptr(b_kvq),
                                 q_out.data_ptr(), (int)q_out.stride(0), M, N, K, gelu, &e, ws.data_ptr(), (size_t)ws.numel() * 4,
                                 stream_of(x)), "rgn_gemm_w8_qkv");
    else
        check_rc(rgn_gemm_bf16_qkv(ptr(x), (int)x.stride(0), ptr(w_kvq), (int)w_kvq.stride(0), ptr(b_kvq), q_out.data_ptr(),
                                   (int)q_out.stride(0), M, N, K, gelu, &e, ws.data_ptr(), (size_t)ws.numel() * 4, stream_of(x)),
                 "rgn_gemm_bf16_qkv");
}

// both streams of a double-stream block in one launch: image rows sit behind the `txt_len` text rows of the shared [text ; image]
// sequence (cache rows, rotary rows); only the image rows are ever partial (fp16 round trip, quirk A-3)
void kv_partial_update_pair_(const Tensor& x_img, const Tensor& w_img, const OptTensor& b_img, Tensor out_img, const Tensor& norm_q_img,
                             const Tensor& norm_k_img, const Tensor& x_txt, const Tensor& w_txt, const OptTensor& b_txt, Tensor out_txt,
                             const Tensor& norm_q_txt, const Tensor& norm_k_txt, const Tensor& cos_q, const Tensor& sin_q,
                             const Tensor& cos_k, const Tensor& sin_k, const OptTensor& kv_rows, Tensor k_cache, Tensor vt_cache,
                             int64_t heads, int64_t txt_len, double eps, bool fp16_roundtrip, const OptTensor& w_scale_img,
                             const OptTensor& w_scale_txt) {
    RGN_DEVICE_GUARD(x_img);
    check_act(x_img, w_img, out_img, w_scale_img, b_img);
    check_act(x_txt, w_txt, out_txt, w_scale_txt, b_txt);
    TORCH_CHECK(w_img.sizes() == w_txt.sizes() && w_img.is_contiguous() && w_txt.is_contiguous() && is_fp8(w_img) == is_fp8(w_txt),
                "pair: equal [N, K], contiguous weights of one format");
    rgn_qkv_epilogue e0 = epi(x_img, norm_q_img, norm_k_img, cos_q, sin_q, cos_k, sin_k, kv_rows, k_cache, vt_cache, heads, txt_len, eps, fp16_roundtrip);
    rgn_qkv_epilogue e1 = epi(x_txt, norm_q_txt, norm_k_txt, cos_q, sin_q, cos_k, sin_k, kv_rows, k_cache, vt_cache, heads, 0, eps, false);
    Tensor ws = gemm_ws(x_img);
    const int N = (int)w_img.size(0), K = (int)w_img.size(1);
    if (is_fp8(w_img))
        check_rc(rgn_gemm_w8_qkv_pair(ptr(x_img), (int)x_img.stride(0), ptr(w_img), (const float*)ptr(w_scale_img), ptr(b_img),
                                      out_img.data_ptr(), (int)out_img.stride(0), (int)x_img.size(0), &e0, ptr(x_txt), (int)x_txt.stride(0),
                                      ptr(w_txt), (const float*)ptr(w_scale_txt), ptr(b_txt), out_txt.data_ptr(), (int)out_txt.stride(0),
                                      (int)x_txt.size(0), &e1, N, K, ws.data_ptr(), (size_t)ws.numel() * 4, stream_of(x_img)),
                 "rgn_gemm_w8_qkv_pair");
    else
        check_rc(rgn_gemm_bf16_qkv_pair(ptr(x_img), (int)x_img.stride(0), ptr(w_img), ptr(b_img), out_img.data_ptr(), (int)out_img.stride(0),
                                        (int)x_img.size(0), &e0, ptr(x_txt), (int)x_txt.stride(0), ptr(w_txt), ptr(b_txt), out_txt.data_ptr(),
                                        (int)out_txt.stride(0), (int)x_txt.size(0), &e1, N, K, ws.data_ptr(), (size_t)ws.numel() * 4,
                                        stream_of(x_img)), "rgn_gemm_bf16_qkv_pair");
}

// the projections of up to four (stream, CFG branch) problems in ONE launch: per problem its activations, weights (shared between
// the branches of a stream), RMSNorm weights, rotary tables, cache-row list and K / V^T cache (one per branch)
void kv_partial_update_group_(at::TensorList x, at::TensorList w_kvq, const c10::List<OptTensor>& w_scale,
                              const c10::List<OptTensor>& b_kvq, at::TensorList q_out,
                              at::TensorList norm_q, at::TensorList norm_k, at::TensorList cos_q, at::TensorList sin_q,
                              at::TensorList cos_k, at::TensorList sin_k, const c10::List<OptTensor>& kv_rows, at::TensorList k_cache,
                              at::TensorList vt_cache, int64_t heads, at::IntArrayRef row_base, double eps, at::IntArrayRef fp16_roundtrip,
                              int64_t gelu_from_col) {
    const size_t n = x.size();
    TORCH_CHECK(n >= 1, "kv_partial_update_group_: no problems");
    RGN_DEVICE_GUARD(x[0]);
    TORCH_CHECK(n >= 1 && n <= 4 && w_kvq.size() == n && b_kvq.size() == n && q_out.size() == n && norm_q.size() == n && norm_k.size() == n &&
                    cos_q.size() == n && sin_q.size() == n && cos_k.size() == n && sin_k.size() == n && kv_rows.size() == n &&
                    k_cache.size() == n && vt_cache.size() == n && row_base.size() == n && (fp16_roundtrip.empty() || fp16_roundtrip.size() == n) &&
                    (w_scale.size() == 0 || w_scale.size() == n), "kv_partial_update_group_: one entry per problem (1..4)");
    std::vector<rgn_qkv_epilogue> es(n);
    std::vector<rgn_gemm_problem> ps;
    for (size_t i = 0; i < n; ++i) {
        OptTensor sc = w_scale.size() ? OptTensor(w_scale.get(i)) : OptTensor();
        check_act(x[i], w_kvq[i], q_out[i], sc, OptTensor(b_kvq.get(i)));
        TORCH_CHECK(w_kvq[i].sizes() == w_kvq[0].sizes() && w_kvq[i].is_contiguous() && is_fp8(w_kvq[i]) == is_fp8(w_kvq[0]),
                    "group: equal [N, K], contiguous weights of one format");
        es[i] = epi(x[i], norm_q[i], norm_k[i], cos_q[i], sin_q[i], cos_k[i], sin_k[i], kv_rows.get(i), k_cache[i], vt_cache[i], heads, row_base[i],
                    eps, !fp16_roundtrip.empty() && fp16_roundtrip[i] != 0);
        if (x[i].size(0) == 0) continue;
        rgn_gemm_problem p;
        p.A = ptr(x[i]); p.W = ptr(w_kvq[i]); p.wscale = (const float*)ptr(sc); p.bias = ptr(OptTensor(b_kvq.get(i)));
        p.C = q_out[i].data_ptr(); p.gate = nullptr; p.resid = nullptr; p.qkv = &es[i];
        p.lda = (int)x[i].stride(0); p.ldc = (int)q_out[i].stride(0); p.M = (int)x[i].size(0);
        ps.push_back(p);
    }
    if (ps.empty()) return;
    Tensor ws = gemm_ws(x[0]);
    check_rc(rgn_gemm_group(ps.data(), (int)ps.size(), (int)w_kvq[0].size(0), (int)w_kvq[0].size(1), RGN_EPI_QKV,
                            (int)(gelu_from_col < 0 ? 0 : gelu_from_col), ws.data_ptr(), (size_t)ws.numel() * 4, stream_of(x[0])),
             "rgn_gemm_group");
}

void region_attention(const Tensor& q, const Tensor& k_cache, const Tensor& vt_cache, Tensor out, int64_t skv, int64_t heads, double scale,
                      double score_bound) {
    RGN_DEVICE_GUARD(q);
    TORCH_CHECK(q.dim() == 2 && out.dim() == 2 && q.scalar_type() == at::kBFloat16 && out.scalar_type() == at::kBFloat16 &&
                    q.stride(1) == 1 && out.stride(1) == 1 && q.size(1) == heads * 128 && out.sizes() == q.sizes(), "region_attention: q / out [Sq, H*128] bf16");
    TORCH_CHECK(k_cache.dim() == 2 && k_cache.size(1) == heads * 128 && k_cache.is_contiguous() && vt_cache.is_contiguous() &&
                    vt_cache.size(0) == heads * 128 && vt_cache.size(1) == k_cache.size(0), "region_attention: cache slabs");
    TORCH_CHECK(skv >= 1 && skv <= k_cache.size(0), "region_attention: skv = ", skv, " outside the cache slab's ", k_cache.size(0), " rows");
    Tensor ws = attn_ws(q);
    check_rc(rgn_attention_bounded(ptr(q), (int)q.stride(0), ptr(k_cache), ptr(vt_cache), (int)k_cache.size(0), out.data_ptr(), (int)out.stride(0),
                                   (int)q.size(0), (int)skv, (int)heads, (float)(scale > 0 ? scale : 1.0 / std::sqrt(128.0)), (float)score_bound,
                                   ws.data_ptr(), (size_t)ws.numel() * 4, stream_of(q)), "rgn_attention_bounded");
}

}  // namespace

// the header this binding was compiled against: regione_amd/torch_ops.py compares both with libregione_hip.so's rgn_version() /
// rgn_abi_struct_bytes() before the first op runs (rgn_qkv_epilogue / rgn_gemm_problem travel by pointer)
extern "C" int rgn_torch_binding_abi_version(void) { return RGN_ABI_VERSION; }
extern "C" size_t rgn_torch_binding_struct_bytes(void) { return sizeof(rgn_qkv_epilogue) * 1000 + sizeof(rgn_gemm_problem); }

// the schemas: identical to regione_amd/torch_ops.py (SCHEMAS there is the single source the tests compare both against)
TORCH_LIBRARY(regione_mi, m) {
    m.def("arp_partition(Tensor sample, Tensor? model_output, Tensor cond, float dt_final, float threshold, int h_tok, int w_tok, "
          "bool erosion_dilation=True) -> (Tensor, Tensor, Tensor)");
    m.def("gather_rows(Tensor x, Tensor ids) -> Tensor");
    m.def("scatter_rows_(Tensor src, Tensor ids, Tensor(a!) dst) -> ()");
    m.def("split_euler_step(Tensor sample, Tensor v, float dt, Tensor? mask=None, float dt_direct=0.0) -> Tensor");
    m.def("avd_apply(Tensor cache, float ratio, Tensor? ids=None, bool round_ratio=False) -> Tensor");
    m.def("cfg_combine(Tensor pos, Tensor neg, float scale, int mode=0, float power=0.4) -> Tensor");
    m.def("kv_partial_update_(Tensor x, Tensor w_kvq, Tensor? b_kvq, Tensor(a!) q_out, Tensor norm_q, Tensor norm_k, Tensor cos_q, "
          "Tensor sin_q, Tensor cos_k, Tensor sin_k, Tensor? kv_rows, Tensor(b!) k_cache, Tensor(c!) vt_cache, int heads, int row_base=0, "
          "float eps=1e-6, bool fp16_roundtrip=False, int gelu_from_col=-1, Tensor? w_scale=None) -> ()");
    m.def("kv_partial_update_pair_(Tensor x_img, Tensor w_img, Tensor? b_img, Tensor(a!) out_img, Tensor norm_q_img, Tensor norm_k_img, "
          "Tensor x_txt, Tensor w_txt, Tensor? b_txt, Tensor(b!) out_txt, Tensor norm_q_txt, Tensor norm_k_txt, "
          "Tensor cos_q, Tensor sin_q, Tensor cos_k, Tensor sin_k, Tensor? kv_rows, Tensor(c!) k_cache, Tensor(d!) vt_cache, "
          "int heads, int txt_len, float eps=1e-6, bool fp16_roundtrip=False, Tensor? w_scale_img=None, Tensor? w_scale_txt=None) -> ()");
    m.def("kv_partial_update_group_(Tensor[] x, Tensor[] w_kvq, Tensor?[] w_scale, Tensor?[] b_kvq, Tensor(a!)[] q_out, Tensor[] norm_q, "
          "Tensor[] norm_k, Tensor[] cos_q, Tensor[] sin_q, Tensor[] cos_k, Tensor[] sin_k, Tensor?[] kv_rows, Tensor(b!)[] k_cache, "
          "Tensor(c!)[] vt_cache, int heads, int[] row_base, float eps=1e-6, int[] fp16_roundtrip=[], int gelu_from_col=-1) -> ()");
    m.def("region_attention(Tensor q, Tensor k_cache, Tensor vt_cache, Tensor(a!) out, int skv, int heads, float scale=-1.0, "
          "float score_bound=0.0) -> ()");
    m.def("workspace(Tensor like, int kind) -> Tensor");
}

// CUDA is the dispatch key of HIP tensors in PyTorch-ROCm; no CPU kernels are registered (a CPU tensor fails loudly)
TORCH_LIBRARY_IMPL(regione_mi, CUDA, m) {
    m.impl("arp_partition", &arp_partition);
    m.impl("gather_rows", &gather_rows);
    m.impl("scatter_rows_", &scatter_rows_);
    m.impl("split_euler_step", &split_euler_step);
    m.impl("avd_apply", &avd_apply);
    m.impl("cfg_combine", &cfg_combine);
    m.impl("kv_partial_update_", &kv_partial_update_);
    m.impl("kv_partial_update_pair_", &kv_partial_update_pair_);
    m.impl("kv_partial_update_group_", &kv_partial_update_group_);
    m.impl("region_attention", &region_attention);
    m.impl("workspace", &workspace_op);
}
