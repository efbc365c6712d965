// C-ABI operator entry points (include/rawformer_hip.h) and error plumbing.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "rf_common.h"

namespace rf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return RF_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return RF_E_DEVICE;
}

int check_launch(const char* what) { return check_hip(hipGetLastError(), what); }

// ---- per-launch event profiler (single-threaded diagnostic; off by default)
struct ProfRec {
    std::string key;
    double flops, bytes;
    hipEvent_t e0, e1;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;

ProfScope::ProfScope(hipStream_t st, const char* key, double flops, double bytes) : st_(st), rec_(-1) {
    if (!g_prof_on) return;
    ProfRec r;
    r.key = key;
    r.flops = flops;
    r.bytes = bytes;
    if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, st);
    rec_ = (int)g_prof.size();
    g_prof.push_back(r);
}

bool profiling_active() { return g_prof_on; }

ProfScope::~ProfScope() {
    if (rec_ >= 0) (void)hipEventRecord(g_prof[rec_].e1, st_);
}

static const float kHaarInit[16] = {   // dwt_init / iwt_init, taps t = 2*row + col, bands LL,HL,LH,HH
    0.5f, 0.5f, 0.5f, 0.5f, -0.5f, 0.5f, -0.5f, 0.5f, -0.5f, -0.5f, 0.5f, 0.5f, 0.5f, -0.5f, -0.5f, 0.5f};
static const float kHaarOrtho[16] = {  // HaarDWT, bands LL,LH,HL,HH
    0.5f, 0.5f, 0.5f, 0.5f, 0.5f, -0.5f, 0.5f, -0.5f, 0.5f, 0.5f, -0.5f, -0.5f, 0.5f, -0.5f, -0.5f, 0.5f};

}  // namespace rf

using namespace rf;

#define RF_TRY(expr)            \
    do {                        \
        const int rc_ = (expr); \
        if (rc_) return rc_;    \
    } while (0)

extern "C" {

const char* rf_last_error(void) { return g_err; }
int rf_version(void) { return 1; }

int rf_profile_begin(void) {
    for (auto& r : g_prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof.clear();
    g_prof_on = true;
    return RF_OK;
}

int rf_profile_end(char* json, size_t len) {
    g_prof_on = false;
    struct Agg { long n = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    int rc = RF_OK;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) rc = RF_E_DEVICE;
        Agg& a = agg[r.key];
        a.n += 1; a.ms += ms; a.flops += r.flops; a.bytes += r.bytes;
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
    if (rc) { set_error("rf_profile_end: event timing failed"); return rc; }
    std::string out = "[";
    bool first = true;
    for (auto& kv : agg) {
        char buf[512];
        snprintf(buf, sizeof(buf), "%s{\"kernel\": \"%s\", \"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.n, kv.second.ms, kv.second.flops, kv.second.bytes);
        out += buf;
        first = false;
    }
    out += "]";
    RF_CHECK_ARG(json && out.size() + 1 <= len, "rf_profile_end: buffer of %zu bytes too small (need %zu)", len, out.size() + 1);
    memcpy(json, out.c_str(), out.size() + 1);
    return RF_OK;
}

int rf_pixel_unshuffle2(const float* in, float* out, int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && B > 0 && C > 0 && h > 0 && w > 0, "pixel_unshuffle2: bad arguments");
    return launch_pixel_unshuffle2(in, out, B, C, h, w, (hipStream_t)stream);
}

int rf_pixel_shuffle2(const float* in, float* out, int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && B > 0 && C > 0 && h > 0 && w > 0, "pixel_shuffle2: bad arguments");
    return launch_pixel_shuffle2(in, out, B, C, h, w, (hipStream_t)stream);
}

int rf_dwt_haar(const float* in, float* out, int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && B > 0 && C > 0 && h > 0 && w > 0, "dwt_haar: bad arguments");
    return launch_dwt2x2(in, out, kHaarInit, 0, 1, B, C, h, w, 2 * h, 2 * w, (hipStream_t)stream);
}

int rf_idwt_haar(const float* in, float* out, int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && B > 0 && C > 0 && h > 0 && w > 0, "idwt_haar: bad arguments");
    return launch_idwt2x2(in, out, kHaarInit, 0, 1, B, C, h, w, (hipStream_t)stream);
}

static void scaled_kernel(const float* k16, int norm, float out[16]) {
    for (int i = 0; i < 16; ++i) out[i] = norm ? k16[i] / 2.0f : k16[i];
}

int rf_dwt_custom(const float* in, float* out, const float* k16, int norm, int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && k16 && B > 0 && C > 0 && h > 0 && w > 0, "dwt_custom: bad arguments");
    float k[16];
    scaled_kernel(k16, norm, k);
    return launch_dwt2x2(in, out, k, 1, 0, B, C, h, w, 2 * h, 2 * w, (hipStream_t)stream);
}

int rf_idwt_custom(const float* in, float* out, const float* k16, int norm, int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && k16 && B > 0 && C > 0 && h > 0 && w > 0, "idwt_custom: bad arguments");
    float k[16];
    scaled_kernel(k16, norm, k);
    return launch_idwt2x2(in, out, k, 1, 0, B, C, h, w, (hipStream_t)stream);
}

int rf_haar_dwt(const float* in, float* out, int B, int C, int hin, int win, void* stream) {
    RF_CHECK_ARG(in && out && B > 0 && C > 0 && hin > 1 && win > 1, "haar_dwt: bad arguments");
    return launch_dwt2x2(in, out, kHaarOrtho, 0, 0, B, C, (hin + 1) / 2, (win + 1) / 2, hin, win, (hipStream_t)stream);
}

int rf_layernorm2d(const float* in, float* out, const float* weight, const float* bias, float eps,
                   int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && weight && B > 0 && C > 0 && h > 0 && w > 0, "layernorm2d: bad arguments");
    return launch_layernorm2d(in, out, weight, bias, eps, B, C, h * w, (hipStream_t)stream);
}

int rf_conv1x1_scratch_bytes(int Cin_total, int Cout, size_t* bytes) {
    RF_CHECK_ARG(bytes && Cin_total > 0 && Cout > 0, "conv1x1_scratch_bytes: bad arguments");
    *bytes = (align_up(packed1x1_floats(Cin_total, Cout), 64) + packed1x1_b3_floats(Cin_total, Cout)) * sizeof(float);
    return RF_OK;
}

int rf_conv1x1(const float* in, const float* in2, float* out, const float* weight, const float* bias,
               const float* ln_w, const float* ln_b, const float* res, void* scratch,
               int B, int C1, int C2, int Cout, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && weight && scratch && aligned16(scratch), "conv1x1: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int K = C1 + C2;
    RF_TRY(pack_1x1(weight, (float*)scratch, Cout, K, K, 1, st));
    float* w3 = (float*)scratch + align_up(packed1x1_floats(K, Cout), 64);
    RF_TRY(pack_1x1_b3(weight, w3, Cout, K, K, 1, st));
    Conv1x1Args a{};
    a.x1 = in; a.C1 = C1; a.x1_bstride = (int64_t)C1 * h * w;
    a.x2 = C2 ? in2 : nullptr; a.C2 = C2; a.x2_bstride = (int64_t)C2 * h * w;
    a.wp = (const float*)scratch; a.wp3 = w3; a.bias = bias; a.ln_w = ln_w; a.ln_b = ln_b; a.ln_eps = 1e-5f;
    a.res = res; a.res_bstride = (int64_t)Cout * h * w;
    a.out = out; a.out_bstride = (int64_t)Cout * h * w; a.Cout = Cout; a.B = B; a.P = h * w; a.w = w;
    return launch_conv1x1(a, st);
}

int rf_dwconv3x3(const float* in, float* out, const float* weight, const float* bias, int gelu,
                 int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && weight && B > 0 && C > 0 && h > 0 && w > 0, "dwconv3x3: bad arguments");
    DwConvArgs a{};
    a.x = in; a.x_bstride = (int64_t)C * h * w; a.out = out; a.out_bstride = (int64_t)C * h * w;
    a.w = weight; a.bias = bias; a.B = B; a.C = C; a.h = h; a.w_ = w; a.gelu = gelu;
    return launch_dwconv3x3(a, (hipStream_t)stream);
}

int rf_conv3x3_scratch_bytes(int Cin, int Cout, size_t* bytes) {
    RF_CHECK_ARG(bytes && Cin > 0 && Cout > 0, "conv3x3_scratch_bytes: bad arguments");
    *bytes = packed3x3_floats(Cin, Cout) * sizeof(float);
    return RF_OK;
}

int rf_conv3x3(const float* in, float* out, const float* weight, const float* bias, void* scratch,
               int act, int store, int B, int Cin, int Cout, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && weight && scratch && aligned16(scratch), "conv3x3: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    RF_TRY(pack_3x3(weight, (float*)scratch, Cout, Cin, st));
    Conv3x3Args a{};
    a.x = in; a.x_bstride = (int64_t)Cin * h * w; a.wp = (const float*)scratch; a.bias = bias;
    a.out = out; a.out_bstride = (int64_t)Cout * h * w; a.B = B; a.Cin = Cin; a.Cout = Cout; a.h = h; a.w = w;
    a.act = act; a.store = store;
    return launch_conv3x3(a, st);
}

int rf_convT2x2_scratch_bytes(int Cin, int Cout, size_t* bytes) {
    RF_CHECK_ARG(bytes && Cin > 0 && Cout > 0, "convT2x2_scratch_bytes: bad arguments");
    *bytes = packed1x1_floats(Cin, 4 * Cout) * sizeof(float);
    return RF_OK;
}

int rf_convT2x2(const float* in, float* out, const float* weight, const float* bias, void* scratch,
                int B, int Cin, int Cout, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && weight && scratch && aligned16(scratch), "convT2x2: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    RF_TRY(pack_convT(weight, (float*)scratch, Cin, Cout, st));
    Conv1x1Args a{};
    a.x1 = in; a.C1 = Cin; a.x1_bstride = (int64_t)Cin * h * w; a.wp = (const float*)scratch; a.bias = bias;
    a.out = out; a.out_bstride = (int64_t)Cout * 4 * h * w; a.Cout = 4 * Cout; a.B = B; a.P = h * w; a.w = w; a.mode = 1;
    return launch_conv1x1(a, st);
}

// scratch layout of rf_chan_attn: packed qkv weights | qkv_pre | qkv | gram partials | folded weights
struct AttnScratch {
    size_t wqkv, wqkv3, pre, qkv, partial, wfold, wfold3, total;
};
static int attn_scratch(int B, int C, int heads, int h, int w, AttnScratch* s) {
    const int P = h * w;
    int ns, sl;
    size_t pf;
    RF_TRY(gram_plan(B, C, heads, P, &ns, &sl, &pf));
    if (attn_mid_supported(C, heads, h, w)) {
        size_t pf2;
        attn_mid_plan(h, w, &ns, &pf2, B, C);
        if (pf2 > pf) pf = pf2;
    }
    size_t off = 0;
    auto take = [&](size_t f) { const size_t o = off; off += align_up(f, 64); return o; };
    s->wqkv = take(packed1x1_floats(C, 3 * C));
    s->wqkv3 = take(packed1x1_b3_floats(C, 3 * C));
    s->pre = take((size_t)B * 3 * C * P);
    s->qkv = take((size_t)B * 3 * C * P);
    s->partial = take(pf);
    s->wfold = take((size_t)B * packed1x1_floats(C, C));
    s->wfold3 = take((size_t)B * packed1x1_b3_floats(C, C));
    s->total = off;
    return RF_OK;
}

int rf_chan_attn_scratch_bytes(int B, int C, int heads, int h, int w, size_t* bytes) {
    RF_CHECK_ARG(bytes && B > 0 && C > 0 && heads > 0 && h > 0 && w > 0, "chan_attn_scratch_bytes: bad arguments");
    AttnScratch s;
    RF_TRY(attn_scratch(B, C, heads, h, w, &s));
    *bytes = s.total * sizeof(float);
    return RF_OK;
}

int rf_chan_attn(const float* in, float* out, const float* qkv_w, const float* qkv_b,
                 const float* dw_w, const float* dw_b, const float* temperature,
                 const float* proj_w, const float* proj_b, void* scratch,
                 int B, int C, int heads, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && qkv_w && dw_w && temperature && proj_w && scratch && aligned16(scratch), "chan_attn: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int P = h * w;
    AttnScratch s;
    RF_TRY(attn_scratch(B, C, heads, h, w, &s));
    float* ws = (float*)scratch;
    RF_TRY(pack_1x1(qkv_w, ws + s.wqkv, 3 * C, C, C, 1, st));
    RF_TRY(pack_1x1_b3(qkv_w, ws + s.wqkv3, 3 * C, C, C, 1, st));
    Conv1x1Args q{};
    q.x1 = in; q.C1 = C; q.x1_bstride = (int64_t)C * P; q.wp = ws + s.wqkv; q.wp3 = ws + s.wqkv3; q.bias = qkv_b;
    q.out = ws + s.pre; q.out_bstride = (int64_t)3 * C * P; q.Cout = 3 * C; q.B = B; q.P = P; q.w = w;
    RF_TRY(launch_conv1x1(q, st));
    Conv1x1Args av{};
    int nslab = 0;
    if (attn_mid_supported(C, heads, h, w)) {
        // the forward's levels 1-2 kernel: depthwise 3x3 + Gram partials + v in one pass (rf_fused.hip)
        size_t pf;
        RF_TRY(attn_mid_plan(h, w, &nslab, &pf, B, C));
        RF_TRY(launch_attn_mid(ws + s.pre, ws + s.qkv, ws + s.partial, nslab, dw_w, dw_b, B, C, h, w, st));
        av.x1 = ws + s.qkv; av.x1_bstride = (int64_t)C * P;
    } else {
        DwConvArgs d{};
        d.x = ws + s.pre; d.x_bstride = (int64_t)3 * C * P; d.out = ws + s.qkv; d.out_bstride = (int64_t)3 * C * P;
        d.w = dw_w; d.bias = dw_b; d.B = B; d.C = 3 * C; d.h = h; d.w_ = w;
        RF_TRY(launch_dwconv3x3(d, st));
        GramArgs g{};
        g.q = ws + s.qkv; g.k = ws + s.qkv + (size_t)C * P; g.bstride = (int64_t)3 * C * P;
        g.B = B; g.C = C; g.heads = heads; g.P = P; g.partial = ws + s.partial;
        size_t pf;
        RF_TRY(gram_plan(B, C, heads, P, &g.nslab, &g.slab, &pf));
        RF_TRY(launch_gram(g, st));
        nslab = g.nslab;
        av.x1 = ws + s.qkv + (size_t)2 * C * P; av.x1_bstride = (int64_t)3 * C * P;
    }
    RF_TRY(launch_attn_fold(ws + s.partial, nslab, temperature, proj_w, ws + s.wfold, ws + s.wfold3, B, C, heads, st));
    av.C1 = C;
    av.wp = ws + s.wfold; av.wp_bstride = (int64_t)packed1x1_floats(C, C); av.bias = proj_b;
    av.wp3 = ws + s.wfold3; av.wp3_bstride = (int64_t)packed1x1_b3_floats(C, C);
    av.out = out; av.out_bstride = (int64_t)C * P; av.Cout = C; av.B = B; av.P = P; av.w = w;
    return launch_conv1x1(av, st);
}

// scratch layout of rf_transformer_block: packed qkv | packed pw1 | packed pw2 | run_transformer buffers
struct TbScratch { size_t wqkv, w1, w2, wqkv3, w13, w23, bufs, total; TbBufOffsets o; };
static void tb_scratch(int B, int C, int heads, int hc, int h, int w, TbScratch* s) {
    size_t off = 0;
    auto take = [&](size_t f) { const size_t r = off; off += align_up(f, 64); return r; };
    s->wqkv = take(packed1x1_floats(C, 3 * C));
    s->w1 = take(packed1x1_floats(C, hc));
    s->w2 = take(packed1x1_floats(hc, C));
    s->wqkv3 = take(packed1x1_b3_floats(C, 3 * C));
    s->w13 = take(packed1x1_b3_floats(C, hc));
    s->w23 = take(packed1x1_b3_floats(hc, C));
    s->bufs = off;
    s->total = off + transformer_scratch_floats(B, C, heads, hc, h, w, &s->o);
}

int rf_transformer_block_scratch_bytes(int B, int C, int heads, int ffn_expansion, int h, int w, size_t* bytes) {
    RF_CHECK_ARG(bytes && B > 0 && C > 0 && heads > 0 && C % heads == 0 && ffn_expansion > 0 && h > 0 && w > 0, "transformer_block_scratch_bytes: bad arguments");
    { int ns, sl; size_t pf; RF_TRY(gram_plan(B, C, heads, h * w, &ns, &sl, &pf)); }
    TbScratch s;
    tb_scratch(B, C, heads, C * ffn_expansion, h, w, &s);
    *bytes = s.total * sizeof(float);
    return RF_OK;
}

int rf_transformer_block(const float* in, float* out, const float* const* prm, void* scratch,
                         int B, int C, int heads, int ffn_expansion, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && prm && scratch && aligned16(scratch), "transformer_block: bad arguments");
    RF_CHECK_ARG(B > 0 && C % 4 == 0 && heads > 0 && C % heads == 0 && C / heads <= 64, "transformer_block: unsupported C=%d heads=%d", C, heads);
    for (int i = 0; i < 17; ++i) RF_CHECK_ARG(prm[i] != nullptr, "transformer_block: parameter %d is null", i);
    hipStream_t st = (hipStream_t)stream;
    const int hc = C * ffn_expansion;
    { int ns, sl; size_t pf; RF_TRY(gram_plan(B, C, heads, h * w, &ns, &sl, &pf)); }
    TbScratch s;
    tb_scratch(B, C, heads, hc, h, w, &s);
    float* ws = (float*)scratch;
    // prm: norm1.w, norm1.b, temperature, qkv.w, qkv.b, qkv_dwconv.w, qkv_dwconv.b, project_out.w, project_out.b,
    //      norm2.w, norm2.b, pointwise1.w, pointwise1.b, depthwise.w, depthwise.b, pointwise2.w, pointwise2.b
    RF_TRY(pack_1x1(prm[3], ws + s.wqkv, 3 * C, C, C, 1, st));
    RF_TRY(pack_1x1(prm[11], ws + s.w1, hc, C, C, 1, st));
    RF_TRY(pack_1x1(prm[15], ws + s.w2, C, hc, hc, 1, st));
    RF_TRY(pack_1x1_b3(prm[3], ws + s.wqkv3, 3 * C, C, C, 1, st));
    RF_TRY(pack_1x1_b3(prm[11], ws + s.w13, hc, C, C, 1, st));
    RF_TRY(pack_1x1_b3(prm[15], ws + s.w23, C, hc, hc, 1, st));
    TbParams p{prm[0], prm[1], prm[2], ws + s.wqkv, prm[4], prm[5], prm[6], prm[7], prm[8],
               prm[9], prm[10], ws + s.w1, prm[12], prm[13], prm[14], ws + s.w2, prm[16],
               ws + s.wqkv3, ws + s.w13, ws + s.w23, 0};
    return run_transformer(p, in, out, ws + s.bufs, s.o, B, C, heads, hc, h, w, st);
}

int rf_flca_scratch_bytes(int B, int C, int h, int w, size_t* bytes) {
    RF_CHECK_ARG(bytes && B > 0 && C > 0 && h > 0 && w > 0, "flca_scratch_bytes: bad arguments");
    *bytes = (align_up((size_t)B * flca_nblk(h, w) * C, 64) + align_up((size_t)B * C, 64)) * sizeof(float);
    return RF_OK;
}

int rf_flca(const float* feat, const float* guide, float* out, const float* const* prm, void* scratch,
            int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(feat && guide && out && prm && scratch && aligned16(scratch), "flca: bad arguments");
    for (int i = 0; i < 10; ++i) RF_CHECK_ARG(prm[i] != nullptr, "flca: parameter %d is null", i);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)scratch;
    float* ch = partial + align_up((size_t)B * flca_nblk(h, w) * C, 64);
    // prm: alpha, beta, gamma, low_attn.0.w, high_attn.0.w, chroma_attn.0.w, se.1.w, se.1.b, se.3.w, se.3.b
    FlcaSpatialArgs a{};
    a.feat = feat; a.xs = out; a.guide = guide;
    a.alpha = prm[0]; a.beta = prm[1]; a.gamma = prm[2]; a.w_low = prm[3]; a.w_high = prm[4]; a.w_chr = prm[5];
    a.partial = partial; a.B = B; a.C = C; a.h = h; a.w = w; a.nblk = flca_nblk(h, w);
    RF_TRY(launch_flca_spatial(a, st));
    const int hid = C / 8 > 8 ? C / 8 : 8;
    RF_TRY(launch_flca_se(partial, a.nblk, h * w, prm[6], prm[7], prm[8], prm[9], hid, ch, B, C, st));
    return launch_scale_channels(out, ch, B, C, h * w, st);
}

int rf_guidance_scratch_bytes(int B, int H, int W, size_t* bytes) {
    RF_CHECK_ARG(bytes && B > 0 && H > 0 && W > 0, "guidance_scratch_bytes: bad arguments");
    *bytes = guidance_scratch_floats(B, H, W) * sizeof(float);
    return RF_OK;
}

int rf_flca_guidance(const float* packed, float* guide, void* scratch, int B, int H, int W, int hf, int wf, void* stream) {
    RF_CHECK_ARG(packed && guide && scratch && hf > 0 && wf > 0, "flca_guidance: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    RF_TRY(launch_guidance_base(packed, 0, 0, (float*)scratch, B, H, W, st));
    return launch_guidance_level((const float*)scratch, guide, B, H, W, hf, wf, st);
}

int rf_upcat_scratch_bytes(int C, size_t* bytes) {
    RF_CHECK_ARG(bytes && C > 0 && C % 4 == 0, "upcat_scratch_bytes: bad arguments");
    *bytes = upcat_packed_floats(C) * sizeof(float);
    return RF_OK;
}

int rf_upcat(const float* x, const float* skip, float* out, const float* up_w, const float* up_b, const float* cr_w, const float* cr_b,
             void* scratch, int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(x && skip && out && up_w && cr_w && scratch && aligned16(scratch), "upcat: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    RF_TRY(pack_upcat(up_w, up_b, cr_w, cr_b, (float*)scratch, C, st));
    return launch_upcat(x, skip, out, (const float*)scratch, B, C, h, w, st);
}
}  // extern "C"
