// Training step of RawFormer (SURVEY.md section 8 f3 / BASELINE configs[4]; reference: train.py:127-147 -- forward, loss,
// backward -- with torch.autograd replaced by an explicit adjoint schedule).  rf_train_step = forward with the activations
// the backward needs kept in the workspace, L1 / Charbonnier loss, backward; the gradient of every parameter lands in ONE
// flat buffer (rf_flat_offset) that the host all-reduces across ranks (RCCL) and hands to rf_adam_step together with the flat
// parameter buffer.  Host code only: kernels in rf_train.hip and the forward files.
//
// The forward here is the op-by-op schedule (the fused level-0 kernels keep their intermediates on chip, which is exactly what a
// backward pass cannot use).  Adjoints:
//   1x1 conv          dX = conv1x1(dY, W^T)                         dW = gram2(dY, X)          db = channel sums of dY
//   3x3 conv          dX = conv3x3(dY, W^T with flipped taps)       dW = gram2<9 taps>(dY, X)
//   depthwise 3x3     dX = dwconv(dY, flipped taps)                 dW, db = dw_wgrad
//   ConvTranspose2d   dX = conv1x1(unshuffle(dY), W as [Cin][4 Cout])   dW = gram2(X, unshuffle(dY))
//   PixelShuffle / downshuffle: each other's adjoint;  LayerNorm, GELU, LeakyReLU: rf_train.hip
//   channel attention (q^ k^T T -> softmax -> A v -> project_out), per image and head, with G = q k^T, rq = 1/|q|, rk = 1/|k|,
//   c = rq G rk (cosines), S = T c, A = softmax(S), o = A v:
//       do = W_out^T dOut;  dW_out = gram2(dOut, o);  dA = gram2(do, v) per image;  dv = A^T do;
//       dS = A (dA - rowsum(dA A));  dT = sum dS c;  dc = T dS;
//       dq = (rq dc rk) k - diag(rq^2 rowsum(dc c)) q;     dk = (rq dc rk)^T q - diag(rk^2 colsum(dc c)) k
//     i.e. d[q;k] = M2 [q;k] with a per-image 2C x 2C matrix and dv = blockdiag(A^T) do: two 1x1 GEMMs with per-image weights.
// Variants 'plain' (conv branch) and 'flca' (rf_train.hip: launch_flca_backward).
//
// Schedule-level choices (round 3): every packed / transposed / tap-flipped weight form of the step is written by three batched
// launches before the forward (pack cache, build_pack_list); bias gradients are row sums inside gram2; the depthwise 3x3 of the
// FFN writes its pre-activation and GELU(.) in one pass; the halves of a concatenated-input gradient are read in place through
// strides; both residual adds of a block ride on the LayerNorm adjoints; inside a stage every backward tensor has its own
// buffer so that the weight-gradient kernels run on a second stream beside the dX chain (fork before each, join at the end of
// the stage); finished ranges of the flat gradient buffer are announced to the caller (rf_set_grad_ready) from its end towards
// its start so that the gradient all-reduce overlaps the rest of the backward.
#include <cstring>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "rf_handle.h"

using namespace rf;

#define RF_TRY(expr)            \
    do {                        \
        const int rc_ = (expr); \
        if (rc_) return rc_;    \
    } while (0)

namespace {

struct Bump {
    float* base;
    size_t off = 0;
    float* take(size_t n) {
        float* p = base ? base + off : nullptr;
        off += align_up(n, 64);
        return p;
    }
};

struct Stash {      // one Conv_Transformer stage
    const float* in;
    float *qkvp, *qkv, *partial, *x1, *f1, *f2, *g, *trans, *xs, *cr, *out;      // g = GELU(f2), written by the same kernel as f2
    float *xraw, *ch, *pool;           // FLCA: xs before the squeeze-excite gate, the gate [B][C], the pooling partial sums
    int nslab, slab;
};

struct TrainPlan {
    Stash st[8];                       // 1..7
    float *x4, *e, *down[3], *up[3], *catr[3], *pred;
    float *gscratch, *guide[4], *flca_scr;
    float *tA, *tB, *tC, *tD, *tE;     // backward temporaries (3 * U0 each)
    float *dskip[3], *dpred, *ga, *gb;
    float *wt1, *wt2;                  // on-the-fly packed / flipped weights (shapes the pack cache does not hold)
    float *pack_cache;                 // every packed weight form of the step, written by a few batched launches at its start
    float *part;                       // reduction partials
    float *part_wg;                    // ... of the kernels on the weight-gradient stream
    float *sb[16];                     // stage_backward's temporaries, one buffer per tensor (see there)
    float *small;                      // attention: per-image C x C matrices and packed per-image weights
    float *loss_part;
    size_t total;
};

size_t max_sz(size_t a, size_t b) { return a > b ? a : b; }

// ---- pack cache ------------------------------------------------------------------------------------------------------
// The step needs most weights in two packed forms: as they multiply in the forward (PF_N) and transposed -- 3x3: also tap-flipped
// -- for the dX product of the backward (PF_T); ConvTranspose2d has a third (PF_CTB, its dX GEMM).  They were packed where they
// were used: ~125 launches of a few microseconds per step.  Now the list is derived from the parameter shapes and filled by
// launch_pack_batch (3 launches) before the forward; the helpers look a weight up by (pointer, form).
enum PackForm { PF_N = 0, PF_T = 1, PF_CTB = 2, PF_N3 = 3, PF_T3 = 4, PF_CTB3 = 5 };      // ..3: the same matrix in b3 form (K >= 128: bf16x3 GEMM)
typedef std::map<std::pair<const float*, int>, const float*> PackMap;

size_t build_pack_list(const rf_handle* h, float* base, std::vector<PackDesc>* list, PackMap* map) {
    size_t off = 0;
    auto add = [&](const Param& q, int form, int kind, int rows, int cols, int64_t rs, int64_t cs, int flip) {
        PackDesc d{q.ptr, base ? base + off : nullptr, kind, rows, cols, rs, cs, flip};
        if (list) list->push_back(d);
        if (map) (*map)[std::make_pair(q.ptr, form)] = d.dst;
        off += align_up(pack_desc_floats(d), 64);
    };
    for (const Param& q : h->params) {
        if (q.ndim != 4 || q.name.find("FLCA.") != std::string::npos) continue;      // the gate convolutions have their own kernels
        const int n0 = (int)q.shape[0], n1 = (int)q.shape[1], kh = (int)q.shape[2];
        if (kh == 1) {                                   // 1x1 conv [Cout][K]
            add(q, PF_N, 0, n0, n1, n1, 1, 0);
            add(q, PF_T, 0, n1, n0, 1, n1, 0);
            if (n1 >= 128) add(q, PF_N3, 3, n0, n1, n1, 1, 0);
            if (n0 >= 128) add(q, PF_T3, 3, n1, n0, 1, n1, 0);
        } else if (kh == 3 && n1 == 1) {                 // depthwise [C][1][3][3]: dX runs the forward kernel on flipped taps
            add(q, PF_T, 2, n0, 9, 9, 1, 0);
        } else if (kh == 3) {                            // 3x3 conv [Cout][Cin][3][3]
            add(q, PF_N, 1, n0, n1, (int64_t)n1 * 9, 9, 0);
            add(q, PF_T, 1, n1, n0, 9, (int64_t)n1 * 9, 1);
        } else if (kh == 2) {                            // ConvTranspose2d [Cin][Cout][2][2]: GEMM row 4 o + 2 i + j, column k (pack_convT)
            add(q, PF_N, 0, 4 * n1, n0, 1, (int64_t)4 * n1, 0);
            add(q, PF_CTB, 0, n0, 4 * n1, (int64_t)4 * n1, 1, 0);
            if (n0 >= 128) add(q, PF_N3, 3, 4 * n1, n0, 1, (int64_t)4 * n1, 0);
            if (4 * n1 >= 128) add(q, PF_CTB3, 3, n0, 4 * n1, (int64_t)4 * n1, 1, 0);
        }
    }
    return off;
}

int make_train_plan(const rf_handle* h, float* base, int B, int H, int W, TrainPlan& p) {
    const rf_config& c = h->cfg;
    Bump b{base};
    const size_t U0 = (size_t)B * c.dim * H * W;
    const int hcx = c.ffn_expansion;
    p.x4 = b.take((size_t)B * 4 * H * W);
    p.e = b.take(U0);
    const bool flca = c.variant == RF_VARIANT_FLCA;
    if (flca) {
        p.gscratch = b.take(guidance_scratch_floats(B, H, W));
        for (int l = 0; l < 4; ++l) p.guide[l] = b.take((size_t)B * 4 * (H >> l) * (W >> l));
    }
    size_t part = 0, small = 0, wt = 0, fscr = 0;
    auto stage = [&](int i, int lvl) -> int {
        const int C = c.dim << lvl, hh = H >> lvl, ww = W >> lvl;
        const size_t U = (size_t)B * C * hh * ww;
        Stash& s = p.st[i];
        s.qkvp = b.take(3 * U); s.qkv = b.take(3 * U); s.x1 = b.take(U); s.f1 = b.take(hcx * U); s.f2 = b.take(hcx * U); s.g = b.take(hcx * U);
        s.trans = b.take(U); s.xs = b.take(U); s.cr = b.take(U); s.out = b.take(U);
        if (flca) {
            s.xraw = b.take(U); s.ch = b.take((size_t)B * C); s.pool = b.take((size_t)B * flca_nblk(hh, ww) * C);
            fscr = max_sz(fscr, flca_bwd_scratch_floats(B, C, hh, ww));
        }
        size_t pf;
        RF_TRY(gram_plan(B, C, c.heads[lvl], hh * ww, &s.nslab, &s.slab, &pf));
        s.partial = b.take(pf);
        const int hc = C * hcx;
        part = max_sz(part, gram2_partial_floats(B, C, C, hh, ww, 9));
        part = max_sz(part, gram2_partial_floats(B, 3 * C, C, hh, ww, 1));
        part = max_sz(part, gram2_partial_floats(B, hc, hc, hh, ww, 1));
        part = max_sz(part, gram2_partial_floats(B, 2 * C, 4 * C, hh, ww, 1));
        part = max_sz(part, ln_bwd_partial_floats(B, C, hh * ww));
        part = max_sz(part, dw_wgrad_partial_floats(B, 3 * C > hc ? 3 * C : hc, hh * ww));
        small = max_sz(small, (size_t)B * (3 * (size_t)C * C + packed1x1_floats(2 * C, 2 * C) + 2 * packed1x1_floats(C, C) + 64) + 64);
        wt = max_sz(wt, max_sz(packed3x3_floats(2 * C, 2 * C), max_sz(packed1x1_floats(4 * C, 4 * C), (size_t)C * C * 9 * 4)));
        return RF_OK;
    };
    for (int i = 1; i <= 4; ++i) {
        RF_TRY(stage(i, i - 1));
        if (i <= 3) p.down[i - 1] = b.take(U0 >> i);
    }
    for (int i = 1; i <= 3; ++i) {
        const int lvl = 3 - i;
        p.up[i - 1] = b.take(U0 >> lvl);
        p.catr[i - 1] = b.take(U0 >> lvl);
        RF_TRY(stage(4 + i, lvl));
        p.dskip[lvl] = b.take(U0 >> lvl);
    }
    p.pred = b.take((size_t)B * c.out_channels * 4 * H * W);
    p.dpred = b.take((size_t)B * c.out_channels * 4 * H * W);
    const size_t T = max_sz(3, (size_t)hcx) * U0;
    p.tA = b.take(T); p.tB = b.take(T); p.tC = b.take(T); p.tD = b.take(T); p.tE = b.take(64);
    p.ga = b.take(U0); p.gb = b.take(U0);
    p.wt1 = b.take(wt); p.wt2 = b.take(wt);
    p.pack_cache = b.take(build_pack_list(h, nullptr, nullptr, nullptr));
    p.part = b.take(max_sz(part, (size_t)B * 64 * 512));
    p.part_wg = b.take(max_sz(part, (size_t)B * 64 * 512));
    {   // d_pre, d_cr, d_cat, d_f2, d_f1, ln2, d_ln2, d_x1, o, d_o, d_qkv, d_qkvp, ln1, d_ln1, d_xs, d_tr   (units of U0)
        const size_t hx = (size_t)hcx;
        const size_t units[16] = {1, 1, 2, hx, hx, 1, 1, 1, 1, 1, 3, 3, 1, 1, 1, 1};
        for (int k = 0; k < 16; ++k) p.sb[k] = b.take(units[k] * U0);
    }
    p.small = b.take(small);
    p.loss_part = b.take(4096);
    p.flca_scr = b.take(fscr);
    p.total = b.off;
    return RF_OK;
}

const float* P(const rf_handle* h, const std::string& n) { return rf_param_ptr(h, n); }

struct Ctx {
    const rf_handle* h;
    TrainPlan* p;
    float* grads;      // flat gradient buffer
    int B;
    hipStream_t st;
    const PackMap* packs = nullptr;
    // weight-gradient stream: inside stage_backward the dW kernels (which nothing downstream of the stage waits for) run beside the
    // dX chain; wg == st: one stream (profiling, or the second stream could not be created)
    hipStream_t wg = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    mutable bool in_stage = false;
    bool forked() const { return in_stage && wg != st; }
    hipStream_t dw_stream() const { return forked() ? wg : st; }
    float* dw_part() const { return forked() ? p->part_wg : p->part; }
    int fork() const {                 // everything enqueued on st so far precedes what is enqueued on wg from now on
        if (!forked()) return RF_OK;
        if (int rc = check_hip(hipEventRecord(ev_fork, st), "weight-gradient stream fork (record)")) return rc;
        return check_hip(hipStreamWaitEvent(wg, ev_fork, 0), "weight-gradient stream fork (wait)");
    }
    int join() const {                 // ... and the reverse
        if (!forked()) return RF_OK;
        if (int rc = check_hip(hipEventRecord(ev_join, wg), "weight-gradient stream join (record)")) return rc;
        return check_hip(hipStreamWaitEvent(st, ev_join, 0), "weight-gradient stream join (wait)");
    }
    float* G(const std::string& n) const { return grads + h->flat_offset[rf_param_index(h, n)]; }
    const float* pk(const float* w, int form) const {      // nullptr: not in the cache (the helper packs on the fly)
        if (!packs) return nullptr;
        auto it = packs->find(std::make_pair(w, form));
        return it == packs->end() ? nullptr : it->second;
    }
};

// ---- forward helpers (raw weights, packed on the fly) ---------------------------------------------------------------
int f_conv1x1(const Ctx& c, const float* x1, int C1, const float* x2, int C2, const float* w, const float* bias, const float* ln_w, const float* ln_b,
              const float* res, float* out, int Cout, int P_, const float* wp_pre = nullptr, int64_t wp_bstride = 0) {
    const int K = C1 + C2;
    const float* wp3 = nullptr;
    if (!wp_pre) {
        wp_pre = c.pk(w, PF_N);
        if (wp_pre) wp3 = c.pk(w, PF_N3);
    }
    if (!wp_pre) RF_TRY(pack_1x1(w, c.p->wt1, Cout, K, K, 1, c.st));
    Conv1x1Args a{};
    a.wp3 = wp3;
    a.x1 = x1; a.C1 = C1; a.x1_bstride = (int64_t)C1 * P_;
    a.x2 = x2; a.C2 = C2; a.x2_bstride = (int64_t)C2 * P_;
    a.wp = wp_pre ? wp_pre : c.p->wt1; a.wp_bstride = wp_bstride; a.bias = bias; a.ln_w = ln_w; a.ln_b = ln_b; a.ln_eps = 1e-5f;
    a.res = res; a.res_bstride = (int64_t)Cout * P_;
    a.out = out; a.out_bstride = (int64_t)Cout * P_; a.Cout = Cout; a.B = c.B; a.P = P_; a.w = P_;
    return launch_conv1x1(a, c.st);
}

int f_conv3x3(const Ctx& c, const float* x, int Cin, const float* w, const float* bias, float* out, int Cout, int hh, int ww, int act, int store,
              int unshuffle_in = 0, const float* wp_pre = nullptr) {
    if (!wp_pre && w) wp_pre = c.pk(w, PF_N);
    if (!wp_pre) RF_TRY(pack_3x3(w, c.p->wt1, Cout, Cin, c.st));
    Conv3x3Args a{};
    a.x = x; a.x_bstride = (int64_t)Cin * hh * ww; a.wp = wp_pre ? wp_pre : c.p->wt1; a.bias = bias; a.out = out;
    a.out_bstride = (int64_t)Cout * hh * ww; a.B = c.B; a.Cin = Cin; a.Cout = Cout; a.h = hh; a.w = ww; a.act = act; a.store = store;
    a.unshuffle_in = unshuffle_in;
    return launch_conv3x3(a, c.st);
}

int f_dw(const Ctx& c, const float* x, const float* w, const float* bias, float* out, int C, int hh, int ww, float* out_gelu = nullptr) {
    DwConvArgs d{};
    d.x = x; d.x_bstride = (int64_t)C * hh * ww; d.out = out; d.out_bstride = (int64_t)C * hh * ww; d.w = w; d.bias = bias;
    d.B = c.B; d.C = C; d.h = hh; d.w_ = ww; d.gelu = 0; d.out2 = out_gelu;
    return launch_dwconv3x3(d, c.st);
}

// ---- backward helpers ------------------------------------------------------------------------------------------------
// dX of a 1x1 conv with raw weight [Cout][K]: conv1x1 with W^T (out: K channels)
// (dy_bstride: floats between the images of dy when it is a channel slice of a wider tensor; 0 = contiguous)
int b_conv1x1_dx(const Ctx& c, const float* dy, int Cout, const float* w, int K, float* dx, int P_, const float* res = nullptr, int64_t dy_bstride = 0) {
    const float* wt = c.pk(w, PF_T);
    if (!wt) {
        RF_TRY(pack_1x1(w, c.p->wt1, K, Cout, 1, K, c.st));             // rows = k, cols = co : element W[co][k] at co * K + k
        wt = c.p->wt1;
    }
    Conv1x1Args a{};
    a.x1 = dy; a.C1 = Cout; a.x1_bstride = dy_bstride ? dy_bstride : (int64_t)Cout * P_; a.wp = wt;
    if (wt != c.p->wt1) a.wp3 = c.pk(w, PF_T3);
    a.res = res; a.res_bstride = (int64_t)K * P_;
    a.out = dx; a.out_bstride = (int64_t)K * P_; a.Cout = K; a.B = c.B; a.P = P_; a.w = P_;
    return launch_conv1x1(a, c.st);
}

// dW [Cout][ld] columns [col0, col0 + Cx) += / = gram2(dy, x);  db = channel sums of dy, taken in the same pass
// (x2 / Cx2: the layer's input is cat(x, x2) along channels, read in place)
int b_conv1x1_dw(const Ctx& c, const float* dy, int Cout, const float* x, int Cx, float* dW, int ld, int col0, float* db, int hh, int ww,
                 int64_t dy_bstride = 0, const float* x2 = nullptr, int Cx2 = 0) {
    const int64_t dys = dy_bstride ? dy_bstride : (int64_t)Cout * hh * ww;
    RF_TRY(c.fork());
    const hipStream_t ws = c.dw_stream();
    float* part = c.dw_part();
    if (Cx2 && Cx % 16 != 0) {       // the two-source contraction cuts the inputs at a tile boundary: otherwise one pass per input
        RF_TRY(launch_gram2(dy, dys, Cout, x, (int64_t)Cx * hh * ww, Cx, dW + col0, ld, part, c.B, hh, ww, 1, 0, 0, 0, 0, 1, ws, col0 == 0 ? db : nullptr));
        return launch_gram2(dy, dys, Cout, x2, (int64_t)Cx2 * hh * ww, Cx2, dW + col0 + Cx, ld, part, c.B, hh, ww, 1, 0, 0, 0, 0, 1, ws);
    }
    return launch_gram2(dy, dys, Cout, x, (int64_t)Cx * hh * ww, Cx, dW + col0, ld, part, c.B, hh, ww,
                        1, 0, 0, 0, 0, 1, ws, col0 == 0 ? db : nullptr, x2, (int64_t)Cx2 * hh * ww, Cx2);
}

int b_conv3x3_dx(const Ctx& c, const float* dy, int Cout, const float* w, int Cin, float* dx, int hh, int ww) {
    const float* wt = c.pk(w, PF_T);
    if (!wt) {
        RF_TRY(launch_flip3x3(w, c.p->wt2, Cout, Cin, 1, c.st));        // [Cin][Cout][flipped taps]
        RF_TRY(pack_3x3(c.p->wt2, c.p->wt1, Cin, Cout, c.st));
        wt = c.p->wt1;
    }
    return f_conv3x3(c, dy, Cout, nullptr, nullptr, dx, Cin, hh, ww, 0, 0, 0, wt);
}

int b_conv3x3_dw(const Ctx& c, const float* dy, int Cout, const float* x, int Cin, float* dW, float* db, int hh, int ww) {
    RF_TRY(c.fork());
    return launch_gram2(dy, (int64_t)Cout * hh * ww, Cout, x, (int64_t)Cin * hh * ww, Cin, dW, Cin, c.dw_part(), c.B, hh, ww, 9, 0, 0, 0, 0, 1, c.dw_stream(), db);
}

int b_dw(const Ctx& c, const float* dy, const float* x, const float* w, float* dx, float* dW, float* db, int C, int hh, int ww) {
    RF_TRY(c.fork());
    RF_TRY(launch_dw_wgrad(x, dy, dW, db, c.dw_part(), c.B, C, hh, ww, 1, c.dw_stream()));
    const float* wf = c.pk(w, PF_T);
    if (!wf) {
        RF_TRY(launch_flip3x3(w, c.p->wt2, C, 1, 0, c.st));
        wf = c.p->wt2;
    }
    return f_dw(c, dy, wf, nullptr, dx, C, hh, ww);
}

// ---- attention, small per-(head, image) kernel ----------------------------------------------------------------------
// layout of `small` per image: dA [C][C] (from gram2), then packed M2 (K = 2C, Cout = 2C), packed A^T (C, C), packed A (C, C)
struct AttnSmall {
    const float* partial; int nslab;       // Gram partials of the forward (rf_attn.hip layout, kRowW = 66)
    const float* temperature;
    const float* dA; size_t dA_istride;    // [B][C][C]
    float* m2; size_t m2_istride;          // packed [2C x 2C], zero-filled by the caller
    float* at; float* ap; size_t a_istride;   // packed A^T and A [C x C], zero-filled by the caller
    float* dT_part;                        // [B][heads]
    int C, heads, fwd_only;
};

__device__ __forceinline__ size_t pk(int NT, int co, int k) { return ((size_t)(k >> 2) * NT + (co >> 4)) * 64 + (co & 15) + 16 * (k & 3); }

__global__ void __launch_bounds__(256) attn_small_kernel(AttnSmall a) {
    const int hd = blockIdx.x, b = blockIdx.y;
    const int C = a.C, c = C / a.heads, NT = (C + 15) >> 4;
    constexpr int kRowW = 66;
    __shared__ float G[64][65], A[64][65], D[64][65];
    __shared__ float nq[64], nk[64], srow[64], tcol[64];
    const float* pb = a.partial + (size_t)b * a.nslab * NT * 16 * kRowW;
    // 1. reduce the slab partials in slab order (same values as attn_fold_kernel's first step)
    for (int v = threadIdx.x; v < c * c + 2 * c; v += 256) {
        int qch, col;
        if (v < c * c) {
            const int ii = v / c, jj = v % c;
            qch = hd * c + ii;
            const int kch = hd * c + jj;
            const int lo_ch = 16 * (qch >> 4), hi_ch = (lo_ch + 15 < C - 1) ? lo_ch + 15 : C - 1;
            const int tklo = ((lo_ch / c) * c) / 16;
            (void)hi_ch;
            col = ((kch >> 4) - tklo) * 16 + (kch & 15);
        } else if (v < c * c + c) { qch = hd * c + (v - c * c); col = 64; }
        else { qch = hd * c + (v - c * c - c); col = 65; }
        const float* src = pb + ((size_t)(qch >> 4) * 16 + (qch & 15)) * kRowW + col;
        float s = 0.f;
        for (int sl = 0; sl < a.nslab; ++sl) s += src[(size_t)sl * NT * 16 * kRowW];
        if (v < c * c) G[v / c][v % c] = s;
        else if (v < c * c + c) nq[v - c * c] = s;
        else nk[v - c * c - c] = s;
    }
    __syncthreads();
    const float T = a.temperature[hd];
    // 2. cosines, softmax
    if (threadIdx.x < c) {
        const int i = threadIdx.x;
        const float rq = 1.0f / fmaxf(sqrtf(nq[i]), 1e-12f);
        float m = -INFINITY;
        for (int j = 0; j < c; ++j) {
            const float cs = G[i][j] * rq * (1.0f / fmaxf(sqrtf(nk[j]), 1e-12f));
            G[i][j] = cs;                       // G now holds the cosines c_ij
            m = fmaxf(m, cs * T);
        }
        float sum = 0.f;
        for (int j = 0; j < c; ++j) { const float e = expf(G[i][j] * T - m); A[i][j] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int j = 0; j < c; ++j) A[i][j] *= inv;
    }
    __syncthreads();
    // packed A (o = blockdiag(A) v) and A^T (dv = blockdiag(A^T) do)
    float* ap = a.ap + (size_t)b * a.a_istride;
    float* at = a.at + (size_t)b * a.a_istride;
    for (int v = threadIdx.x; v < c * c; v += 256) {
        const int i = v / c, j = v % c;
        ap[pk(NT, hd * c + i, hd * c + j)] = A[i][j];
        at[pk(NT, hd * c + j, hd * c + i)] = A[i][j];
    }
    if (a.fwd_only) return;
    // 3. dS = A (dA - rowsum(dA A));  dc = T dS
    const float* dAb = a.dA + (size_t)b * a.dA_istride;
    for (int v = threadIdx.x; v < c * c; v += 256) D[v / c][v % c] = dAb[(size_t)(hd * c + v / c) * C + hd * c + v % c];
    __syncthreads();
    if (threadIdx.x < c) {
        const int i = threadIdx.x;
        float dot = 0.f;
        for (int j = 0; j < c; ++j) dot = fmaf(D[i][j], A[i][j], dot);
        float dts = 0.f, sr = 0.f;
        for (int j = 0; j < c; ++j) {
            const float dS = A[i][j] * (D[i][j] - dot);
            dts = fmaf(dS, G[i][j], dts);
            const float dc = T * dS;
            D[i][j] = dc;                       // D now holds dc_ij
            sr = fmaf(dc, G[i][j], sr);
        }
        srow[i] = sr;
        nq[i] = 1.0f / fmaxf(sqrtf(nq[i]), 1e-12f);      // nq, nk now hold rq, rk
        tcol[i] = dts;                                   // (reused below as the per-row dT contribution)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int i = 0; i < c; ++i) s += tcol[i];
        a.dT_part[(size_t)b * a.heads + hd] = s;
    }
    __syncthreads();
    if (threadIdx.x < c) {
        const int j = threadIdx.x;
        nk[j] = 1.0f / fmaxf(sqrtf(nk[j]), 1e-12f);
        float tc = 0.f;
        for (int i = 0; i < c; ++i) tc = fmaf(D[i][j], G[i][j], tc);
        tcol[j] = tc;
    }
    __syncthreads();
    // 4. M2: rows dq (0..C), dk (C..2C); columns q (0..C), k (C..2C)
    float* m2 = a.m2 + (size_t)b * a.m2_istride;
    const int NT2 = (2 * C + 15) >> 4;
    for (int v = threadIdx.x; v < c * c; v += 256) {
        const int i = v / c, j = v % c;
        const float mqk = nq[i] * D[i][j] * nk[j];
        m2[pk(NT2, hd * c + i, C + hd * c + j)] = mqk;            // dq_i += mqk k_j
        m2[pk(NT2, C + hd * c + j, hd * c + i)] = mqk;            // dk_j += mqk q_i
    }
    for (int i = threadIdx.x; i < c; i += 256) {
        m2[pk(NT2, hd * c + i, hd * c + i)] = -nq[i] * nq[i] * srow[i];
        m2[pk(NT2, C + hd * c + i, C + hd * c + i)] = -nk[i] * nk[i] * tcol[i];
    }
}

int attn_small(const Ctx& c, const Stash& s, const float* temperature, int C, int heads, int fwd_only, float* dT) {
    const size_t CC = (size_t)C * C;
    float* sm = c.p->small;
    const size_t per = 3 * CC + packed1x1_floats(2 * C, 2 * C) + 2 * packed1x1_floats(C, C);
    AttnSmall a{};
    a.partial = s.partial; a.nslab = s.nslab; a.temperature = temperature;
    a.dA = sm; a.dA_istride = per;
    a.m2 = sm + 3 * CC; a.m2_istride = per;
    a.at = a.m2 + packed1x1_floats(2 * C, 2 * C); a.ap = a.at + packed1x1_floats(C, C); a.a_istride = per;
    a.dT_part = sm + (size_t)c.B * per;
    a.C = C; a.heads = heads; a.fwd_only = fwd_only;
    attn_small_kernel<<<dim3((unsigned)heads, (unsigned)c.B), 256, 0, c.st>>>(a);
    RF_TRY(check_launch("attn_small"));
    if (!fwd_only && dT) RF_TRY(launch_reduce_rows(a.dT_part, dT, c.B, (size_t)heads, 1, c.st));   // dT[h] += sum over images, in order
    return RF_OK;
}

}  // namespace

// ---- the step ---------------------------------------------------------------------------------------------------------
namespace {

struct StageNames { std::string pre, t; };

int stage_forward(const Ctx& c, int i, int lvl, const float* in, int H, int W) {
    const rf_handle* h = c.h;
    const rf_config& cfg = h->cfg;
    const int C = cfg.dim << lvl, hh = H >> lvl, ww = W >> lvl, Pn = hh * ww, heads = cfg.heads[lvl], hc = C * cfg.ffn_expansion;
    const std::string pre = "conv_tran" + std::to_string(i) + ".", t = pre + "Transformer.";
    Stash& s = c.p->st[i];
    s.in = in;
    RF_TRY(f_conv1x1(c, in, C, nullptr, 0, P(h, t + "attn.qkv.weight"), P(h, t + "attn.qkv.bias"), P(h, t + "norm1.body.weight"),
                     P(h, t + "norm1.body.bias"), nullptr, s.qkvp, 3 * C, Pn));
    RF_TRY(f_dw(c, s.qkvp, P(h, t + "attn.qkv_dwconv.weight"), P(h, t + "attn.qkv_dwconv.bias"), s.qkv, 3 * C, hh, ww));
    GramArgs g{};
    g.q = s.qkv; g.k = s.qkv + (size_t)C * Pn; g.bstride = (int64_t)3 * C * Pn; g.B = c.B; g.C = C; g.heads = heads; g.P = Pn;
    g.partial = s.partial; g.nslab = s.nslab; g.slab = s.slab;
    RF_TRY(launch_gram(g, c.st));
    // attention map -> per-image folded projection (forward only), x1 = in + W_out A v + b
    float* wfold = c.p->small;
    RF_TRY(launch_attn_fold(s.partial, s.nslab, P(h, t + "attn.temperature"), P(h, t + "attn.project_out.weight"), wfold, nullptr, c.B, C, heads, c.st));
    {
        Conv1x1Args a{};
        a.x1 = s.qkv + (size_t)2 * C * Pn; a.C1 = C; a.x1_bstride = (int64_t)3 * C * Pn; a.wp = wfold; a.wp_bstride = (int64_t)packed1x1_floats(C, C);
        a.bias = P(h, t + "attn.project_out.bias"); a.res = in; a.res_bstride = (int64_t)C * Pn;
        a.out = s.x1; a.out_bstride = (int64_t)C * Pn; a.Cout = C; a.B = c.B; a.P = Pn; a.w = ww;
        RF_TRY(launch_conv1x1(a, c.st));
    }
    RF_TRY(f_conv1x1(c, s.x1, C, nullptr, 0, P(h, t + "ffn.pointwise1.weight"), P(h, t + "ffn.pointwise1.bias"), P(h, t + "norm2.body.weight"),
                     P(h, t + "norm2.body.bias"), nullptr, s.f1, hc, Pn));
    RF_TRY(f_dw(c, s.f1, P(h, t + "ffn.depthwise.weight"), P(h, t + "ffn.depthwise.bias"), s.f2, hc, hh, ww, s.g));   // f2 and g = gelu(f2)
    RF_TRY(f_conv1x1(c, s.g, hc, nullptr, 0, P(h, t + "ffn.pointwise2.weight"), P(h, t + "ffn.pointwise2.bias"), nullptr, nullptr, s.x1, s.trans, C, Pn));
    if (cfg.variant == RF_VARIANT_FLCA) {
        const std::string f = pre + "FLCA.";
        FlcaSpatialArgs sa{};
        sa.feat = in; sa.xs = s.xraw; sa.guide = c.p->guide[lvl];
        sa.w_low = P(h, f + "low_attn.0.weight"); sa.w_high = P(h, f + "high_attn.0.weight"); sa.w_chr = P(h, f + "chroma_attn.0.weight");
        sa.alpha = P(h, f + "alpha"); sa.beta = P(h, f + "beta"); sa.gamma = P(h, f + "gamma");
        sa.partial = s.pool; sa.B = c.B; sa.C = C; sa.h = hh; sa.w = ww; sa.nblk = flca_nblk(hh, ww);
        RF_TRY(launch_flca_spatial(sa, c.st));
        const int hid = C / 8 > 8 ? C / 8 : 8;
        RF_TRY(launch_flca_se(s.pool, sa.nblk, Pn, P(h, f + "se.1.weight"), P(h, f + "se.1.bias"), P(h, f + "se.3.weight"), P(h, f + "se.3.bias"),
                              hid, s.ch, c.B, C, c.st));
        RF_TRY(launch_scale_channels_to(s.xraw, s.xs, s.ch, c.B, C, Pn, c.st));                         // xs = branch output z
    } else {
        RF_TRY(f_conv3x3(c, in, C, P(h, pre + "conv.weight"), P(h, pre + "conv.bias"), s.xs, C, hh, ww, cfg.branch_lrelu ? 1 : 0, 0));
    }
    RF_TRY(f_conv1x1(c, s.xs, C, s.trans, C, P(h, pre + "channel_reduce.weight"), P(h, pre + "channel_reduce.bias"), nullptr, nullptr, nullptr, s.cr, C, Pn));
    RF_TRY(f_conv3x3(c, s.cr, C, P(h, pre + "Conv_out.weight"), P(h, pre + "Conv_out.bias"), s.out, C, hh, ww, 1, 0));
    return RF_OK;
}

// dout: gradient w.r.t. the stage output (consumed); din: receives the gradient w.r.t. the stage input
int stage_backward(const Ctx& c, int i, int lvl, float* dout, float* din, int H, int W) {
    const rf_handle* h = c.h;
    const rf_config& cfg = h->cfg;
    const int C = cfg.dim << lvl, hh = H >> lvl, ww = W >> lvl, Pn = hh * ww, heads = cfg.heads[lvl], hc = C * cfg.ffn_expansion;
    const std::string pre = "conv_tran" + std::to_string(i) + ".", t = pre + "Transformer.";
    const Stash& s = c.p->st[i];
    const size_t U = (size_t)c.B * C * Pn;
    // One buffer per tensor: the weight-gradient kernels read them from their own stream while the dX chain moves on, so nothing
    // is overwritten inside a stage; the stage ends with a join, after which the next stage reuses the buffers.
    float* const* sb = c.p->sb;
    float *d_pre = sb[0], *d_cr = sb[1], *d_cat = sb[2], *d_f2 = sb[3], *d_f1 = sb[4], *ln2 = sb[5], *d_ln2 = sb[6], *d_x1 = sb[7], *o = sb[8],
          *d_o = sb[9], *d_qkv = sb[10], *d_qkvp = sb[11], *ln1 = sb[12], *d_ln1 = sb[13], *d_xs = sb[14], *d_tr = sb[15];
    c.in_stage = true;
    // Conv_out + LeakyReLU
    RF_TRY(launch_ewise(dout, s.out, d_pre, U, 2, 0.2f, c.st));                                      // d(pre-activation)
    RF_TRY(b_conv3x3_dw(c, d_pre, C, s.cr, C, c.G(pre + "Conv_out.weight"), c.G(pre + "Conv_out.bias"), hh, ww));
    RF_TRY(b_conv3x3_dx(c, d_pre, C, P(h, pre + "Conv_out.weight"), C, d_cr, hh, ww));
    // channel_reduce over cat[xs, trans]
    RF_TRY(b_conv1x1_dw(c, d_cr, C, s.xs, C, c.G(pre + "channel_reduce.weight"), 2 * C, 0, c.G(pre + "channel_reduce.bias"), hh, ww, 0, s.trans, C));
    RF_TRY(b_conv1x1_dx(c, d_cr, C, P(h, pre + "channel_reduce.weight"), 2 * C, d_cat, Pn));           // d_cat = [dxs ; dtrans] per image
    // dxs and dtrans are read in place as channel slices of d_cat (image stride 2C Pn) by the kernels that take a stride; only the
    // plain variant's element-wise LeakyReLU adjoint (and a LayerNorm shape without the fused kernel) needs contiguous halves
    const float* dxs = d_cat;
    const float* dtr = d_cat + (size_t)C * Pn;
    int64_t half_bs = (int64_t)2 * C * Pn;
    if (cfg.variant != RF_VARIANT_FLCA || !ln_bwd_fused_shape(C, Pn)) {
        RF_TRY(launch_split_halves(d_cat, d_xs, d_tr, c.B, C, Pn, c.st));
        dxs = d_xs; dtr = d_tr; half_bs = (int64_t)C * Pn;
    }
    if (cfg.variant == RF_VARIANT_FLCA) {
        const std::string f = pre + "FLCA.";
        const char* names[10] = {"alpha", "beta", "gamma", "low_attn.0.weight", "high_attn.0.weight", "chroma_attn.0.weight",
                                 "se.1.weight", "se.1.bias", "se.3.weight", "se.3.bias"};
        const float* prm[10];
        float* grd[10];
        for (int k = 0; k < 10; ++k) { prm[k] = P(h, f + names[k]); grd[k] = c.G(f + names[k]); }
        RF_TRY(launch_flca_backward(s.in, c.p->guide[lvl], s.xraw, dxs, half_bs, s.ch, s.pool, flca_nblk(hh, ww), prm, grd, din, 0,
                                    c.p->flca_scr, c.B, C, hh, ww, c.st));                             // din = branch part
    } else {
        // conv branch
        if (cfg.branch_lrelu) RF_TRY(launch_ewise(d_xs, s.xs, d_xs, U, 2, 0.2f, c.st));
        RF_TRY(b_conv3x3_dw(c, d_xs, C, s.in, C, c.G(pre + "conv.weight"), c.G(pre + "conv.bias"), hh, ww));
        RF_TRY(b_conv3x3_dx(c, d_xs, C, P(h, pre + "conv.weight"), C, din, hh, ww));                  // din = branch part
    }
    // FFN:  trans = x1 + pw2(gelu(dw(pw1(LN2(x1)))))          dtr = dtrans (also the residual part of dx1)
    RF_TRY(b_conv1x1_dw(c, dtr, C, s.g, hc, c.G(t + "ffn.pointwise2.weight"), hc, 0, c.G(t + "ffn.pointwise2.bias"), hh, ww, half_bs));
    RF_TRY(b_conv1x1_dx(c, dtr, C, P(h, t + "ffn.pointwise2.weight"), hc, d_f2, Pn, nullptr, half_bs));   // dg ...
    RF_TRY(launch_ewise(d_f2, s.f2, d_f2, (size_t)c.B * hc * Pn, 1, 0.f, c.st));                      // ... -> df2, in place
    RF_TRY(b_dw(c, d_f2, s.f1, P(h, t + "ffn.depthwise.weight"), d_f1, c.G(t + "ffn.depthwise.weight"), c.G(t + "ffn.depthwise.bias"), hc, hh, ww));
    RF_TRY(launch_layernorm2d(s.x1, ln2, P(h, t + "norm2.body.weight"), P(h, t + "norm2.body.bias"), 1e-5f, c.B, C, Pn, c.st));   // LN2(x1) again
    RF_TRY(b_conv1x1_dw(c, d_f1, hc, ln2, C, c.G(t + "ffn.pointwise1.weight"), C, 0, c.G(t + "ffn.pointwise1.bias"), hh, ww));
    RF_TRY(b_conv1x1_dx(c, d_f1, hc, P(h, t + "ffn.pointwise1.weight"), C, d_ln2, Pn));               // d LN2 out
    // d_x1 = dtrans + (LayerNorm adjoint), dtrans read in place
    RF_TRY(launch_ln_bwd(s.x1, d_ln2, P(h, t + "norm2.body.weight"), d_x1, c.G(t + "norm2.body.weight"), c.p->part, c.B, C, Pn, 1e-5f, 0, 1, c.st, dtr, half_bs));
    // attention:  x1 = in + W_out (A v) + b          (the residual din += dx1 rides on the last kernel of the stage)
    const size_t CC = (size_t)C * C;
    const size_t per = 3 * CC + packed1x1_floats(2 * C, 2 * C) + 2 * packed1x1_floats(C, C);
    RF_TRY(check_hip(hipMemsetAsync(c.p->small, 0, (c.B * (per + 64)) * sizeof(float), c.st), "memset"));
    RF_TRY(attn_small(c, s, P(h, t + "attn.temperature"), C, heads, 1, nullptr));                     // packed A, A^T
    float* m2 = c.p->small + 3 * CC;
    float* at = m2 + packed1x1_floats(2 * C, 2 * C);
    float* ap = at + packed1x1_floats(C, C);
    const float* v = s.qkv + (size_t)2 * C * Pn;
    {   // o = blockdiag(A) v
        Conv1x1Args a{};
        a.x1 = v; a.C1 = C; a.x1_bstride = (int64_t)3 * C * Pn; a.wp = ap; a.wp_bstride = (int64_t)per;
        a.out = o; a.out_bstride = (int64_t)C * Pn; a.Cout = C; a.B = c.B; a.P = Pn; a.w = ww;
        RF_TRY(launch_conv1x1(a, c.st));
    }
    RF_TRY(b_conv1x1_dw(c, d_x1, C, o, C, c.G(t + "attn.project_out.weight"), C, 0, c.G(t + "attn.project_out.bias"), hh, ww));
    RF_TRY(b_conv1x1_dx(c, d_x1, C, P(h, t + "attn.project_out.weight"), C, d_o, Pn));
    // dA per image: on the dX chain (the softmax adjoint waits for it)
    RF_TRY(launch_gram2(d_o, (int64_t)C * Pn, C, v, (int64_t)3 * C * Pn, C, c.p->small, C, c.p->part, c.B, hh, ww, 1, 0, 0, 1, per, 0, c.st));
    RF_TRY(attn_small(c, s, P(h, t + "attn.temperature"), C, heads, 0, c.G(t + "attn.temperature")));
    {   // d(qkv): [dq ; dk] = M2 [q ; k],  dv = blockdiag(A^T) do
        Conv1x1Args a{};
        a.x1 = s.qkv; a.C1 = 2 * C; a.x1_bstride = (int64_t)3 * C * Pn; a.wp = m2; a.wp_bstride = (int64_t)per;
        a.out = d_qkv; a.out_bstride = (int64_t)3 * C * Pn; a.Cout = 2 * C; a.B = c.B; a.P = Pn; a.w = ww;
        RF_TRY(launch_conv1x1(a, c.st));
        Conv1x1Args d{};
        d.x1 = d_o; d.C1 = C; d.x1_bstride = (int64_t)C * Pn; d.wp = at; d.wp_bstride = (int64_t)per;
        d.out = d_qkv + (size_t)2 * C * Pn; d.out_bstride = (int64_t)3 * C * Pn; d.Cout = C; d.B = c.B; d.P = Pn; d.w = ww;
        RF_TRY(launch_conv1x1(d, c.st));
    }
    RF_TRY(b_dw(c, d_qkv, s.qkvp, P(h, t + "attn.qkv_dwconv.weight"), d_qkvp, c.G(t + "attn.qkv_dwconv.weight"), c.G(t + "attn.qkv_dwconv.bias"), 3 * C, hh, ww));
    RF_TRY(launch_layernorm2d(s.in, ln1, P(h, t + "norm1.body.weight"), P(h, t + "norm1.body.bias"), 1e-5f, c.B, C, Pn, c.st));   // LN1(in) again
    RF_TRY(b_conv1x1_dw(c, d_qkvp, 3 * C, ln1, C, c.G(t + "attn.qkv.weight"), C, 0, c.G(t + "attn.qkv.bias"), hh, ww));
    RF_TRY(b_conv1x1_dx(c, d_qkvp, 3 * C, P(h, t + "attn.qkv.weight"), C, d_ln1, Pn));                // d LN1 out
    // din (branch part) += dx1 (residual of x1 = in + attention) + (LayerNorm adjoint)
    RF_TRY(launch_ln_bwd(s.in, d_ln1, P(h, t + "norm1.body.weight"), din, c.G(t + "norm1.body.weight"), c.p->part, c.B, C, Pn, 1e-5f, 1, 1, c.st, d_x1, (int64_t)C * Pn));
    RF_TRY(c.join());
    c.in_stage = false;
    return RF_OK;
}

// second stream of the step (the handle's branch stream and events, rf_handle.h): none while profiling -- the per-kernel brackets
// assume one stream -- or when it cannot be created
hipStream_t train_side_stream(rf_handle* h, hipStream_t st) {
    if (h->side_failed || profiling_active()) return st;
    if (!h->side) {
        if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            h->side_failed = true;
            h->side = nullptr;
            return st;
        }
    }
    return h->side;
}

}  // namespace

// Modules in the order their gradients become final = reverse registry order (the registry is in forward order).
std::vector<std::string> grad_order(const rf_handle* h) {
    (void)h;
    return {"conv_out.", "conv_tran7.", "up3.", "conv_tran6.", "up2.", "conv_tran5.", "up1.", "conv_tran4.", "conv_tran3.", "conv_tran2.", "conv_tran1.", "embedding."};
}
// first flat float of the module with this prefix ("up<i>." is followed by "channel_reduce<i>." in the registry, "conv_tran<i>." by
// "down<i>.": a range runs from its module's first float to the previous watermark, so those ride along)
size_t module_start(const rf_handle* h, const std::string& prefix) {
    size_t best = h->flat_floats;
    for (size_t i = 0; i < h->params.size(); ++i)
        if (h->params[i].name.compare(0, prefix.size(), prefix) == 0 && h->flat_offset[i] < best) best = h->flat_offset[i];
    return best;
}
struct GradNotifier {
    const rf_handle* h; hipStream_t st; size_t watermark;
    void done(const std::string& prefix) {
        const size_t lo = module_start(h, prefix);
        if (lo >= watermark) return;
        if (h->grad_ready) h->grad_ready(h->grad_ready_user, lo, watermark - lo, (void*)st);
        watermark = lo;
    }
};

extern "C" {

int rf_set_grad_ready(rf_handle* h, rf_grad_ready_fn ready, void* user) {
    RF_CHECK_ARG(h, "rf_set_grad_ready: null handle");
    h->grad_ready = ready; h->grad_ready_user = ready ? user : nullptr;
    return RF_OK;
}

int rf_grad_range_count(const rf_handle* h, int* count) {
    RF_CHECK_ARG(h && count, "rf_grad_range_count: null argument");
    size_t wm = h->flat_floats; int n = 0;
    for (const std::string& m : grad_order(h)) { const size_t lo = module_start(h, m); if (lo < wm) { ++n; wm = lo; } }
    *count = n;
    return RF_OK;
}

int rf_grad_range(const rf_handle* h, int index, size_t* offset, size_t* count) {
    RF_CHECK_ARG(h && offset && count && index >= 0, "rf_grad_range: bad arguments");
    size_t wm = h->flat_floats; int n = 0;
    for (const std::string& m : grad_order(h)) {
        const size_t lo = module_start(h, m);
        if (lo >= wm) continue;
        if (n == index) { *offset = lo; *count = wm - lo; return RF_OK; }
        ++n; wm = lo;
    }
    set_error("rf_grad_range: index %d out of range (%d ranges)", index, n);
    return RF_E_INVALID;
}

int rf_flat_param_floats(const rf_handle* h, size_t* floats) {
    RF_CHECK_ARG(h && floats, "rf_flat_param_floats: null argument");
    *floats = h->flat_floats;
    return RF_OK;
}

int rf_flat_offset(const rf_handle* h, int index, size_t* offset) {
    RF_CHECK_ARG(h && offset && index >= 0 && index < (int)h->params.size(), "rf_flat_offset: index %d out of range", index);
    *offset = h->flat_offset[index];
    return RF_OK;
}

int rf_train_workspace_bytes(const rf_handle* h, int B, int H, int W, size_t* bytes) {
    RF_CHECK_ARG(h && bytes && B > 0 && H % 8 == 0 && W % 8 == 0 && H > 0 && W > 0, "rf_train_workspace_bytes: bad arguments");
    TrainPlan p;
    RF_TRY(make_train_plan(h, nullptr, B, H, W, p));
    *bytes = p.total * sizeof(float);
    return RF_OK;
}

int rf_train_step(rf_handle* h, const float* in, const float* gt, float* grads, float* loss_out, float* pred_out, void* workspace,
                  size_t workspace_bytes, int B, int H, int W, int loss_mode, float loss_eps, void* stream) {
    RF_CHECK_ARG(h && in && gt && grads && loss_out && workspace && aligned16(workspace) && aligned16(grads), "rf_train_step: bad arguments");
    RF_CHECK_ARG((h->cfg.variant == RF_VARIANT_PLAIN || h->cfg.variant == RF_VARIANT_FLCA) && !h->cfg.clamp_io,
                 "rf_train_step: variants 'plain' and 'flca' without clamp_io have their adjoint so far");
    RF_CHECK_ARG(B > 0 && B <= 65535 && H % 8 == 0 && W % 8 == 0 && W % 32 == 0, "rf_train_step: packed size %dx%d (H %% 8, W %% 32 == 0)", H, W);
    for (const Param& q : h->params) RF_CHECK_ARG(q.ptr, "rf_train_step: parameter '%s' not set", q.name.c_str());
    TrainPlan p;
    RF_TRY(make_train_plan(h, (float*)workspace, B, H, W, p));
    if (workspace_bytes < p.total * sizeof(float)) {
        set_error("rf_train_step: workspace of %zu bytes, need %zu", workspace_bytes, p.total * sizeof(float));
        return RF_E_NOMEM;
    }
    hipStream_t st = (hipStream_t)stream;
    const rf_config& cfg = h->cfg;
    const int d = cfg.dim, oc = cfg.out_channels;
    Ctx c{h, &p, grads, B, st};
    RF_TRY(check_hip(hipMemsetAsync(grads, 0, h->flat_floats * sizeof(float), st), "memset grads"));
    std::vector<PackDesc> pack_list;
    PackMap pack_map;
    build_pack_list(h, p.pack_cache, &pack_list, &pack_map);
    RF_TRY(launch_pack_batch(pack_list.data(), (int)pack_list.size(), st));
    c.packs = &pack_map;
    c.wg = train_side_stream(h, st);
    c.ev_fork = h->ev_fork; c.ev_join = h->ev_join;

    // ------------------------------------------------------------------ forward
    RF_TRY(launch_pixel_unshuffle2(in, p.x4, B, 1, H, W, st));
    if (cfg.variant == RF_VARIANT_FLCA) {
        RF_TRY(launch_guidance_base(p.x4, 0, 0, p.gscratch, B, H, W, st));
        for (int l = 0; l < 4; ++l) RF_TRY(launch_guidance_level(p.gscratch, p.guide[l], B, H, W, H >> l, W >> l, st));
    }
    RF_TRY(f_conv3x3(c, p.x4, 4, P(h, "embedding.weight"), P(h, "embedding.bias"), p.e, d, H, W, 0, 0));
    const float* cur = p.e;
    for (int i = 1; i <= 3; ++i) {
        const int lvl = i - 1, C = d << lvl, hh = H >> lvl, ww = W >> lvl;
        RF_TRY(stage_forward(c, i, lvl, cur, H, W));
        RF_TRY(f_conv3x3(c, p.st[i].out, C, P(h, "down" + std::to_string(i) + ".body.0.weight"), nullptr, p.down[i - 1], C / 2, hh, ww, 0, 1));
        cur = p.down[i - 1];
    }
    RF_TRY(stage_forward(c, 4, 3, cur, H, W));
    cur = p.st[4].out;
    for (int i = 1; i <= 3; ++i) {
        const int lvl = 3 - i, C = d << lvl, hh = H >> lvl, ww = W >> lvl, Pn = hh * ww;
        const std::string u = "up" + std::to_string(i), r = "channel_reduce" + std::to_string(i);
        const float* upw = c.pk(P(h, u + ".weight"), PF_N);
        if (!upw) {
            RF_TRY(pack_convT(P(h, u + ".weight"), p.wt1, 2 * C, C, st));
            upw = p.wt1;
        }
        Conv1x1Args up{};
        up.x1 = cur; up.C1 = 2 * C; up.x1_bstride = (int64_t)2 * C * (Pn / 4); up.wp = upw; up.bias = P(h, u + ".bias");
        if (upw != p.wt1) up.wp3 = c.pk(P(h, u + ".weight"), PF_N3);
        up.out = p.up[i - 1]; up.out_bstride = (int64_t)C * Pn; up.Cout = 4 * C; up.B = B; up.P = Pn / 4; up.w = ww / 2; up.mode = 1;
        RF_TRY(launch_conv1x1(up, st));
        RF_TRY(f_conv1x1(c, p.up[i - 1], C, p.st[lvl + 1].out, C, P(h, r + ".weight"), P(h, r + ".bias"), nullptr, nullptr, nullptr, p.catr[i - 1], C, Pn));
        RF_TRY(stage_forward(c, 4 + i, lvl, p.catr[i - 1], H, W));
        cur = p.st[4 + i].out;
    }
    RF_TRY(f_conv3x3(c, cur, d, P(h, "conv_out.weight"), P(h, "conv_out.bias"), p.pred, 4 * oc, H, W, 1, 2));
    const size_t npred = (size_t)B * oc * 4 * H * W;
    if (pred_out) RF_TRY(check_hip(hipMemcpyAsync(pred_out, p.pred, npred * 4, hipMemcpyDeviceToDevice, st), "copy pred"));

    // ------------------------------------------------------------------ loss
    RF_TRY(launch_loss(p.pred, gt, p.dpred, loss_out, p.loss_part, npred, loss_mode, loss_eps, st));

    // ------------------------------------------------------------------ backward
    float *tA = p.tA, *tB = p.tB;
    // conv_out + LeakyReLU + PixelShuffle
    RF_TRY(launch_pixel_unshuffle2(p.dpred, tA, B, oc, H, W, st));          // [B, 4 oc, H, W]
    RF_TRY(launch_pixel_unshuffle2(p.pred, tB, B, oc, H, W, st));
    RF_TRY(launch_ewise(tA, tB, tA, (size_t)B * 4 * oc * H * W, 2, 0.2f, st));
    RF_TRY(b_conv3x3_dw(c, tA, 4 * oc, p.st[7].out, d, c.G("conv_out.weight"), c.G("conv_out.bias"), H, W));
    GradNotifier note{h, st, h->flat_floats};
    note.done("conv_out.");
    float* tE_src = p.tC;
    RF_TRY(b_conv3x3_dx(c, tA, 4 * oc, P(h, "conv_out.weight"), d, tE_src, H, W));    // d(stage 7 out)
    for (int l = 0; l < 3; ++l) RF_TRY(check_hip(hipMemsetAsync(p.dskip[l], 0, ((size_t)B * (d << l) * (H >> l) * (W >> l)) * 4, st), "memset dskip"));
    // ga: gradient w.r.t. the output of the stage about to be processed; gb receives the gradient w.r.t. its input
    float *ga = p.ga, *gb = p.gb;
    RF_TRY(check_hip(hipMemcpyAsync(ga, tE_src, (size_t)B * d * H * W * 4, hipMemcpyDeviceToDevice, st), "copy"));
    // decoder, top-down
    for (int i = 3; i >= 1; --i) {
        const int lvl = 3 - i, C = d << lvl, hh = H >> lvl, ww = W >> lvl, Pn = hh * ww;
        const std::string u = "up" + std::to_string(i), r = "channel_reduce" + std::to_string(i);
        RF_TRY(stage_backward(c, 4 + i, lvl, ga, gb, H, W));                                   // gb = d(catr_i)
        note.done("conv_tran" + std::to_string(4 + i) + ".");
        // channel_reduce_i over cat[up, skip]
        RF_TRY(b_conv1x1_dw(c, gb, C, p.up[i - 1], C, c.G(r + ".weight"), 2 * C, 0, c.G(r + ".bias"), hh, ww, 0, p.st[lvl + 1].out, C));
        RF_TRY(b_conv1x1_dx(c, gb, C, P(h, r + ".weight"), 2 * C, p.tC, Pn));                 // tC = [dup ; dskip]
        RF_TRY(launch_split_halves(p.tC, p.tA, p.dskip[lvl], B, C, Pn, st));
        // ConvTranspose2d(2C -> C): dX = conv1x1(unshuffle(dup), W as [2C][4C]);  dW = gram2(x, unshuffle(dup));  db = channel sums of dup
        RF_TRY(launch_chan_sum(p.tA, (int64_t)C * Pn, c.G(u + ".bias"), p.part, B, C, Pn, 1, st));
        RF_TRY(launch_pixel_unshuffle2(p.tA, p.tB, B, C, hh / 2, ww / 2, st));                 // [B, 4C, hh/2, ww/2]
        const float* xin = (i == 1) ? p.st[4].out : p.st[4 + i - 1].out;
        RF_TRY(launch_gram2(xin, (int64_t)2 * C * (Pn / 4), 2 * C, p.tB, (int64_t)4 * C * (Pn / 4), 4 * C, c.G(u + ".weight"), 4 * C, p.part, B, hh / 2, ww / 2,
                            1, 0, 0, 0, 0, 1, st));
        const float* upb = c.pk(P(h, u + ".weight"), PF_CTB);
        if (!upb) {
            RF_TRY(pack_1x1(P(h, u + ".weight"), p.wt1, 2 * C, 4 * C, 4 * C, 1, st));
            upb = p.wt1;
        }
        Conv1x1Args a{};
        a.x1 = p.tB; a.C1 = 4 * C; a.x1_bstride = (int64_t)4 * C * (Pn / 4); a.wp = upb;
        if (upb != p.wt1) a.wp3 = c.pk(P(h, u + ".weight"), PF_CTB3);
        a.out = ga; a.out_bstride = (int64_t)2 * C * (Pn / 4); a.Cout = 2 * C; a.B = B; a.P = Pn / 4; a.w = ww / 2;
        RF_TRY(launch_conv1x1(a, st));                                                         // ga = d(previous stage out) [B, 2C, Pn/4]
        note.done(u + ".");                                                                    // up_i and channel_reduce_i
    }
    // bottleneck and encoder, bottom-up: ga = d(stage i out) on entry (for i <= 3 after the Downsample adjoint and the skip gradient)
    for (int i = 4; i >= 1; --i) {
        const int lvl = i - 1, C = d << lvl, hh = H >> lvl, ww = W >> lvl, Pn = hh * ww;
        if (i <= 3) {
            // gb = d(down_i output) [B, 2C, Pn/4] = d(stage i+1 input)
            RF_TRY(launch_pixel_shuffle2(gb, p.tA, B, C / 2, hh / 2, ww / 2, st));             // [B, C/2, hh, ww]
            RF_TRY(b_conv3x3_dw(c, p.tA, C / 2, p.st[i].out, C, c.G("down" + std::to_string(i) + ".body.0.weight"), nullptr, hh, ww));
            RF_TRY(b_conv3x3_dx(c, p.tA, C / 2, P(h, "down" + std::to_string(i) + ".body.0.weight"), C, ga, hh, ww));
            RF_TRY(launch_ewise(ga, p.dskip[lvl], ga, (size_t)B * C * Pn, 0, 0.f, st));
        }
        RF_TRY(stage_backward(c, i, lvl, ga, gb, H, W));                                       // gb = d(stage i input)
        note.done("conv_tran" + std::to_string(i) + ".");                                      // and down_i, whose gradient came first
    }
    float* dcur = gb;
    // embedding
    RF_TRY(b_conv3x3_dw(c, dcur, d, p.x4, 4, c.G("embedding.weight"), c.G("embedding.bias"), H, W));
    note.done("embedding.");
    if (note.watermark != 0 && h->grad_ready) h->grad_ready(h->grad_ready_user, 0, note.watermark, (void*)st);      // anything in front (never, by construction)
    return RF_OK;
}

int rf_adam_step(float* params, const float* grads, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int decoupled, int step, float grad_scale, void* stream) {
    RF_CHECK_ARG(params && grads && m && v && step >= 1, "rf_adam_step: bad arguments");
    return launch_adam(params, grads, m, v, n, lr, beta1, beta2, eps, weight_decay, decoupled, step, grad_scale, (hipStream_t)stream);
}

}  // extern "C"
