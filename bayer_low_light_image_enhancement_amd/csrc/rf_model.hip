// RawFormer handle: parameter registry, weight packing, workspace plan and the forward
// schedule (RawFomer_WFB_FFAB/model.py:473-508 == FrequencyawareLumaChromaAttentionRAWFormer.py:330-370).
// Host code only; every kernel lives in the rf_*.hip files next to this one.
#include <string>
#include <unordered_map>
#include <vector>
#include <cstring>
#include <cstdlib>
#include "rf_common.h"

using namespace rf;

#include "rf_handle.h"

namespace {

void add_param(rf_handle* h, const std::string& name, std::initializer_list<int64_t> shape) {
    Param p;
    p.name = name;
    p.ndim = (int)shape.size();
    int i = 0;
    for (auto s : shape) p.shape[i++] = s;
    for (; i < 4; ++i) p.shape[i] = 1;
    p.ptr = nullptr;
    h->index[name] = (int)h->params.size();
    h->params.push_back(p);
}

void add_pack(rf_handle* h, const std::string& name, PackKind kind) {
    const int pi = h->index.at(name);
    const Param& p = h->params[pi];
    PackItem it;
    it.param = pi;
    it.kind = kind;
    it.offset = h->packed_floats;
    if (kind == PK_1x1) it.floats = packed1x1_floats((int)p.shape[1], (int)p.shape[0]);
    else if (kind == PK_3x3) it.floats = packed3x3_floats((int)p.shape[1], (int)p.shape[0]);
    else if (kind == PK_1x1_B3) it.floats = packed1x1_b3_floats((int)p.shape[1], (int)p.shape[0]);
    else it.floats = packed1x1_floats((int)p.shape[0], 4 * (int)p.shape[1]);
    h->packed_floats += align_up(it.floats, 64);
    (kind == PK_1x1_B3 ? h->pack3_index : h->pack_index)[name] = (int)h->packs.size();
    h->packs.push_back(it);
}

// Stage tail  channel_reduce(cat(branch, x1 + pointwise2(g)))  as one bf16x3 GEMM over [branch ; x1 ; g] (K = 2C + hidden):
// K blocks of 32 channels must not straddle the sources.  Level 0 of RawFormer-S / -B runs the fused FFN kernel instead
// (run_stage decides per call: the fused kernel takes only some image sizes).
bool tail_composable(int C, int hc) { return C % 32 == 0 && hc % 32 == 0 && hc > 0; }

void add_stage(rf_handle* h, int i, int C, int heads) {
    const rf_config& cfg = h->cfg;
    const std::string pre = "conv_tran" + std::to_string(i) + ".";
    if (cfg.variant == RF_VARIANT_TRUECOLOR) {
        // EnhancedFLCA (BayerTORGBColorMultiLvl.py:192-231), state_dict order
        const std::string f = pre + "FLCA.";
        const int hid = C / 8 > 8 ? C / 8 : 8;
        add_param(h, f + "color_attention.0.weight", {C, 5, 3, 3});
        add_param(h, f + "color_attention.0.bias", {C});
        add_param(h, f + "low_attn.0.weight", {C, 1, 3, 3});
        add_param(h, f + "low_attn.0.bias", {C});
        add_param(h, f + "high_attn.0.weight", {C, 1, 3, 3});
        add_param(h, f + "high_attn.0.bias", {C});
        add_param(h, f + "se.1.weight", {hid, C, 1, 1});
        add_param(h, f + "se.1.bias", {hid});
        add_param(h, f + "se.3.weight", {C, hid, 1, 1});
        add_param(h, f + "se.3.bias", {C});
        add_param(h, f + "res_proj.0.weight", {C, C, 1, 1});
        add_param(h, f + "res_proj.0.bias", {C});
        add_param(h, f + "res_proj.2.weight", {C, C, 1, 1});
        add_param(h, f + "res_proj.2.bias", {C});
        for (const char* n : {"res_proj.0.weight", "res_proj.2.weight"}) {
            add_pack(h, f + n, PK_1x1);
            add_pack(h, f + n, PK_1x1_B3);
        }
    } else if (cfg.variant == RF_VARIANT_FLCA) {
        const std::string f = pre + "FLCA.";
        const int hid = C / 8 > 8 ? C / 8 : 8;
        add_param(h, f + "alpha", {});
        add_param(h, f + "beta", {});
        add_param(h, f + "gamma", {});
        add_param(h, f + "low_attn.0.weight", {C, 1, 3, 3});
        add_param(h, f + "high_attn.0.weight", {C, 1, 3, 3});
        add_param(h, f + "chroma_attn.0.weight", {C, 2, 3, 3});
        add_param(h, f + "se.1.weight", {hid, C, 1, 1});
        add_param(h, f + "se.1.bias", {hid});
        add_param(h, f + "se.3.weight", {C, hid, 1, 1});
        add_param(h, f + "se.3.bias", {C});
    } else {
        add_param(h, pre + "conv.weight", {C, C, 3, 3});
        add_param(h, pre + "conv.bias", {C});
        add_pack(h, pre + "conv.weight", PK_3x3);
    }
    const std::string t = pre + "Transformer.";
    const int hc = C * cfg.ffn_expansion;
    add_param(h, t + "norm1.body.weight", {C});
    add_param(h, t + "norm1.body.bias", {C});
    add_param(h, t + (cfg.variant == RF_VARIANT_TRUECOLOR ? "attn.log_temperature" : "attn.temperature"), {heads, 1, 1});
    add_param(h, t + "attn.qkv.weight", {3 * C, C, 1, 1});
    add_param(h, t + "attn.qkv.bias", {3 * C});
    add_param(h, t + "attn.qkv_dwconv.weight", {3 * C, 1, 3, 3});
    add_param(h, t + "attn.qkv_dwconv.bias", {3 * C});
    add_param(h, t + "attn.project_out.weight", {C, C, 1, 1});
    add_param(h, t + "attn.project_out.bias", {C});
    add_param(h, t + "norm2.body.weight", {C});
    add_param(h, t + "norm2.body.bias", {C});
    add_param(h, t + "ffn.pointwise1.weight", {hc, C, 1, 1});
    add_param(h, t + "ffn.pointwise1.bias", {hc});
    add_param(h, t + "ffn.depthwise.weight", {hc, 1, 3, 3});
    add_param(h, t + "ffn.depthwise.bias", {hc});
    add_param(h, t + "ffn.pointwise2.weight", {C, hc, 1, 1});
    add_param(h, t + "ffn.pointwise2.bias", {C});
    add_param(h, pre + "channel_reduce.weight", {C, 2 * C, 1, 1});
    add_param(h, pre + "channel_reduce.bias", {C});
    add_param(h, pre + "Conv_out.weight", {C, C, 3, 3});
    add_param(h, pre + "Conv_out.bias", {C});
    add_pack(h, t + "attn.qkv.weight", PK_1x1);
    add_pack(h, t + "ffn.pointwise1.weight", PK_1x1);
    add_pack(h, t + "ffn.pointwise2.weight", PK_1x1);
    add_pack(h, t + "attn.qkv.weight", PK_1x1_B3);          // b3 forms for the bf16x3 GEMM kernels (rf_common.h)
    add_pack(h, t + "ffn.pointwise1.weight", PK_1x1_B3);
    add_pack(h, t + "ffn.pointwise2.weight", PK_1x1_B3);
    if (cfg.variant == RF_VARIANT_PLAIN) {
        add_pack(h, pre + "channel_reduce.weight", PK_1x1);
        add_pack(h, pre + "channel_reduce.weight", PK_1x1_B3);
    }
    if (tail_composable(C, hc)) {      // pointwise2 composed into channel_reduce (run_stage)
        h->tail_offset[i] = h->packed_floats;
        h->packed_floats += align_up(tail_composed_floats(C, hc), 64);
        if (cfg.variant == RF_VARIANT_PLAIN) {
            h->tail3_offset[i] = h->packed_floats;
            h->packed_floats += align_up(packed1x1_b3_floats(2 * C + hc, C), 64);
        }
    }
    add_pack(h, pre + "Conv_out.weight", PK_3x3);
}

const float* P(const rf_handle* h, const std::string& name) { return h->params[h->index.at(name)].ptr; }
const float* PK(const rf_handle* h, const std::string& name) { return h->packed + h->packs[h->pack_index.at(name)].offset; }
const float* PK3(const rf_handle* h, const std::string& name) { return h->packed + h->packs[h->pack3_index.at(name)].offset; }

// ---- workspace plan ---------------------------------------------------------------------
struct Plan {
    size_t total = 0;
    size_t gscratch, guide[4], skip[3], tA, tB, tU, bufA, bufB, x1, trans, xs, cr;
    size_t gram_partial, wfold_attn, wfold_cr, wfold_attn3, wfold_cr3, flca_partial, ch;
    size_t ks, ks_floats;       // scratch of the 3x3 convs' input-channel split (small frames only: ks_floats = 0 otherwise)
    int guide_planes;
};

size_t take(Plan& p, size_t floats) {
    const size_t off = p.total;
    p.total += align_up(floats, 64);   // 256-byte granules keep every buffer 16-byte aligned
    return off;
}

int make_plan(const rf_handle* h, int B, int H, int W, Plan& p) {
    const rf_config& c = h->cfg;
    const size_t U0 = (size_t)B * c.dim * H * W;   // floats of a level-0 activation
    const bool tc = c.variant == RF_VARIANT_TRUECOLOR;
    const int levels = c.flca_levels > 0 ? c.flca_levels : 2;
    p.guide_planes = tc ? 7 : 4;
    p.gscratch = take(p, tc ? tc_front_scratch_floats(B, H, W, levels) : guidance_scratch_floats(B, H, W));
    for (int l = 0; l < 4; ++l) p.guide[l] = take(p, (size_t)B * p.guide_planes * (H >> l) * (W >> l));
    for (int l = 0; l < 3; ++l) p.skip[l] = take(p, U0 >> l);
    p.tA = take(p, U0);
    p.tB = take(p, U0);
    p.tU = take(p, U0);
    // widest TransformerBlock intermediate: qkv (3C) or the FFN hidden tensor (ffn_expansion * C) on the op-by-op path
    const size_t wide = (size_t)(c.ffn_expansion > 3 ? c.ffn_expansion : 3);
    p.bufA = take(p, wide * U0);
    p.bufB = take(p, wide * U0);
    p.x1 = take(p, U0);
    p.trans = take(p, U0);
    p.xs = take(p, U0);
    p.cr = take(p, U0);
    size_t gp = 0, wa = 0, wc = 0, fp = 0, wa3 = 0, wc3 = 0;
    for (int l = 0; l < 4; ++l) {
        const int C = c.dim << l, Pl = (H >> l) * (W >> l);
        int ns, sl;
        size_t pf;
        const int rc = gram_plan(B, C, c.heads[l], Pl, &ns, &sl, &pf);
        if (rc) return rc;
        if (pf > gp) gp = pf;
        if (fused_attn_supported(C, c.heads[l], H >> l, W >> l)) {
            size_t pf2;
            fused_attn_plan(H >> l, W >> l, &ns, &pf2, B, C);
            if (pf2 > gp) gp = pf2;
        }
        if (attn_mid_supported(C, c.heads[l], H >> l, W >> l)) {
            size_t pf2;
            attn_mid_plan(H >> l, W >> l, &ns, &pf2, B, C);
            if (pf2 > gp) gp = pf2;
        }
        const size_t a = (size_t)B * packed1x1_floats(C, C), cr = (size_t)B * packed1x1_floats(2 * C, C);
        if (a > wa) wa = a;
        if (cr > wc) wc = cr;
        const int hcl = C * c.ffn_expansion;
        const size_t a3 = (size_t)B * packed1x1_b3_floats(C, C),
                     cr3 = (size_t)B * packed1x1_b3_floats(tail_composable(C, hcl) ? 2 * C + hcl : 2 * C, C);
        if (a3 > wa3) wa3 = a3;
        if (cr3 > wc3) wc3 = cr3;
        const size_t f = (size_t)B * (tc ? tc_nblk(H >> l, W >> l) : flca_nblk(H >> l, W >> l)) * C;
        if (f > fp) fp = f;
    }
    p.gram_partial = take(p, gp);
    p.wfold_attn = take(p, wa);
    p.wfold_cr = take(p, wc);
    p.wfold_attn3 = take(p, wa3);
    p.wfold_cr3 = take(p, wc3);
    p.flca_partial = take(p, fp);
    p.ch = take(p, (size_t)B * (c.dim << 3));
    p.ks_floats = 0;
    for (int l = 0; l < 4; ++l) {
        const size_t f = conv3x3_ksplit_floats(B, c.dim << l, H >> l, W >> l);
        if (f > p.ks_floats) p.ks_floats = f;
    }
    p.ks = take(p, p.ks_floats);
    return RF_OK;
}

#define RF_TRY(expr)          \
    do {                      \
        const int rc_ = (expr); \
        if (rc_) return rc_;  \
    } while (0)

// ---- branch stream -------------------------------------------------------------------------
// A stage's branch (FLCA gates + squeeze-excite fold, or the plain variant's 3x3) depends on the stage input only, like the
// TransformerBlock beside it; so does the guidance pyramid at the head of the forward.  On a single frame every kernel of both
// chains is a few dozen microseconds of mostly latency, so the branch runs on a second stream: fork = an event on the caller's
// stream that the branch stream waits for, join = the reverse before channel_reduce.  Off while profiling (the per-kernel
// brackets assume one stream), for a spatial shard (its collectives stay on the caller's stream) and for TrueColor (its
// branch shares bufA with the block).
hipStream_t branch_stream(rf_handle* h, hipStream_t st) {
    if (h->side_failed || profiling_active() || h->shard_allreduce || h->cfg.variant == RF_VARIANT_TRUECOLOR) return st;
#ifdef RF_DIAG   // diagnostic build only: everything on the caller's stream
    if (getenv("RF_NO_SIDE")) return st;
#endif
    if (!h->side) {
        if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            h->side_failed = true;       // no second stream: the single-stream schedule is always valid
            h->side = nullptr;
            return st;
        }
    }
    return h->side;
}
int fork_branch(rf_handle* h, hipStream_t st, hipStream_t side) {
    if (side == st) return RF_OK;
    RF_TRY(check_hip(hipEventRecord(h->ev_fork, st), "branch fork (record)"));
    return check_hip(hipStreamWaitEvent(side, h->ev_fork, 0), "branch fork (wait)");
}
int join_branch(rf_handle* h, hipStream_t st, hipStream_t side) {
    if (side == st) return RF_OK;
    RF_TRY(check_hip(hipEventRecord(h->ev_join, side), "branch join (record)"));
    return check_hip(hipStreamWaitEvent(st, h->ev_join, 0), "branch join (wait)");
}

// one Conv_Transformer stage
int run_stage(rf_handle* h, int i, int lvl, const float* in, float* out, float* ws, const Plan& p,
              int B, int H, int W, hipStream_t st, hipStream_t side) {
    const rf_config& cfg = h->cfg;
    const int C = cfg.dim << lvl, hh = H >> lvl, ww = W >> lvl, Pn = hh * ww, heads = cfg.heads[lvl];
    const int hc = C * cfg.ffn_expansion;
    const std::string pre = "conv_tran" + std::to_string(i) + ".", t = pre + "Transformer.";
    float* trans = ws + p.trans;
    float* xs = ws + p.xs;
    float* crb = ws + p.cr;

    // TransformerBlock: x + attn(LN1(x)), then x + ffn(LN2(x))  (rf_block.hip)
    TbParams tp{};
    tp.ln1_w = P(h, t + "norm1.body.weight"); tp.ln1_b = P(h, t + "norm1.body.bias");
    if (cfg.variant != RF_VARIANT_TRUECOLOR) tp.temperature = P(h, t + "attn.temperature");
    tp.qkv_wp = PK(h, t + "attn.qkv.weight"); tp.qkv_b = P(h, t + "attn.qkv.bias");
    tp.qkv_dw_w = P(h, t + "attn.qkv_dwconv.weight"); tp.qkv_dw_b = P(h, t + "attn.qkv_dwconv.bias");
    tp.proj_w = P(h, t + "attn.project_out.weight"); tp.proj_b = P(h, t + "attn.project_out.bias");
    tp.ln2_w = P(h, t + "norm2.body.weight"); tp.ln2_b = P(h, t + "norm2.body.bias");
    tp.pw1_wp = PK(h, t + "ffn.pointwise1.weight"); tp.pw1_b = P(h, t + "ffn.pointwise1.bias");
    tp.dw_w = P(h, t + "ffn.depthwise.weight"); tp.dw_b = P(h, t + "ffn.depthwise.bias");
    tp.pw2_wp = PK(h, t + "ffn.pointwise2.weight"); tp.pw2_b = P(h, t + "ffn.pointwise2.bias");
    tp.qkv_wp3 = PK3(h, t + "attn.qkv.weight"); tp.pw1_wp3 = PK3(h, t + "ffn.pointwise1.weight"); tp.pw2_wp3 = PK3(h, t + "ffn.pointwise2.weight");
    if (cfg.variant == RF_VARIANT_TRUECOLOR) { tp.temperature = P(h, t + "attn.log_temperature"); tp.log_temperature = 1; }
    // spatial shard: this level's interior rows and the frame's pixel count for the pooled mean
    const bool sharded = h->shard_allreduce != nullptr;
    const int ylo = sharded ? h->shard_y_lo >> lvl : 0, yhi = sharded ? h->shard_y_hi >> lvl : 0;
    const int P_pool = sharded ? (h->shard_total_rows >> lvl) * ww : Pn;
    if (sharded) { tp.ylo = ylo; tp.yhi = yhi; tp.allreduce = h->shard_allreduce; tp.allreduce_user = h->shard_user; }
    TbBufOffsets to{p.bufA, p.bufB, p.x1, p.gram_partial, p.wfold_attn, p.wfold_attn3};
    // Composed tail: where the FFN runs op by op, its last GEMM (x1 + W2 g + b2 -> trans, K = hidden) and channel_reduce
    // ([Wa' | Wb] [xs ; trans], K = 2C) become ONE GEMM over [xs ; x1 ; g] with [Wa' | Wb | Wb W2] (same MFMA count; `trans` --
    // C floats per pixel written and read back -- never exists).  The bias and Wb W2 are composed at parameter load.
    bool compose = h->tail_offset[i] != 0 && Pn % 4 == 0 && !transformer_ffn_is_fused(tp, C, hc, hh, ww);
#ifdef RF_DIAG   // diagnostic build only: the two-GEMM form
    if (getenv("RF_NO_COMPOSE") || getenv("RF_NO_B3")) compose = false;
#endif
    tp.defer_pw2 = compose;
    const float* composed = compose ? h->packed + h->tail_offset[i] : nullptr;
    // the branch is launched first (on the branch stream when there is one), the block beside it; TrueColor's branch borrows
    // bufA and therefore follows the block on the same stream
    if (cfg.variant == RF_VARIANT_TRUECOLOR) RF_TRY(run_transformer(tp, in, trans, ws, to, B, C, heads, hc, hh, ww, st));
    else RF_TRY(fork_branch(h, st, side));

    // branch, cat, channel_reduce -------------------------------------------------------------
    Conv1x1Args r{};
    r.x1 = xs; r.C1 = C; r.x1_bstride = (int64_t)C * Pn;
    r.x2 = trans; r.C2 = C; r.x2_bstride = (int64_t)C * Pn;
    r.bias = P(h, pre + "channel_reduce.bias");
    r.out = crb; r.out_bstride = (int64_t)C * Pn; r.Cout = C; r.B = B; r.P = Pn; r.w = ww;
    if (cfg.variant == RF_VARIANT_TRUECOLOR) {
        // EnhancedFLCA (BayerTORGBColorMultiLvl.py:249-293): spatial gate -> x + 0.2 tanh(res_proj(x)) -> squeeze-excite (folded
        // into channel_reduce like the FLCA variant's)
        const std::string f = pre + "FLCA.";
        RF_TRY(launch_tc_spatial(in, xs, ws + p.guide[lvl], P(h, f + "color_attention.0.weight"), P(h, f + "color_attention.0.bias"),
                                 P(h, f + "low_attn.0.weight"), P(h, f + "low_attn.0.bias"), P(h, f + "high_attn.0.weight"),
                                 P(h, f + "high_attn.0.bias"), B, C, hh, ww, st));
        Conv1x1Args r0{};
        r0.x1 = xs; r0.C1 = C; r0.x1_bstride = (int64_t)C * Pn; r0.wp = PK(h, f + "res_proj.0.weight"); r0.wp3 = PK3(h, f + "res_proj.0.weight");
        r0.bias = P(h, f + "res_proj.0.bias"); r0.out = crb; r0.out_bstride = (int64_t)C * Pn; r0.Cout = C; r0.B = B; r0.P = Pn; r0.w = ww; r0.act = 2;
        RF_TRY(launch_conv1x1(r0, st));
        Conv1x1Args r2 = r0;
        r2.x1 = crb; r2.wp = PK(h, f + "res_proj.2.weight"); r2.wp3 = PK3(h, f + "res_proj.2.weight"); r2.bias = P(h, f + "res_proj.2.bias");
        r2.out = ws + p.bufA; r2.act = 0;
        RF_TRY(launch_conv1x1(r2, st));
        RF_TRY(launch_tc_residual(xs, ws + p.bufA, xs, ws + p.flca_partial, B, C, hh, ww, st));
        const int hid = C / 8 > 8 ? C / 8 : 8;
        RF_TRY(launch_flca_se_fold(ws + p.flca_partial, tc_nblk(hh, ww), Pn, P(h, f + "se.1.weight"), P(h, f + "se.1.bias"),
                                   P(h, f + "se.3.weight"), P(h, f + "se.3.bias"), hid, P(h, pre + "channel_reduce.weight"),
                                   ws + p.wfold_cr, ws + p.wfold_cr3, ws + p.ch, B, C, st, composed, hc));
        r.wp = ws + p.wfold_cr; r.wp_bstride = (int64_t)packed1x1_floats(2 * C, C);
        r.wp3 = ws + p.wfold_cr3; r.wp3_bstride = (int64_t)packed1x1_b3_floats(compose ? 2 * C + hc : 2 * C, C);
    } else if (cfg.variant == RF_VARIANT_FLCA) {
        const std::string f = pre + "FLCA.";
        FlcaSpatialArgs s{};
        s.feat = in; s.xs = xs; s.guide = ws + p.guide[lvl];
        s.w_low = P(h, f + "low_attn.0.weight"); s.w_high = P(h, f + "high_attn.0.weight"); s.w_chr = P(h, f + "chroma_attn.0.weight");
        s.alpha = P(h, f + "alpha"); s.beta = P(h, f + "beta"); s.gamma = P(h, f + "gamma");
        s.partial = ws + p.flca_partial; s.B = B; s.C = C; s.h = hh; s.w = ww; s.nblk = flca_nblk(hh, ww);
        s.ylo = ylo; s.yhi = yhi;
        RF_TRY(launch_flca_spatial(s, side));
        if (sharded) h->shard_allreduce(h->shard_user, s.partial, (size_t)B * s.nblk * C, 0, (void*)side);
        const int hid = C / 8 > 8 ? C / 8 : 8;
        RF_TRY(launch_flca_se_fold(s.partial, s.nblk, P_pool, P(h, f + "se.1.weight"), P(h, f + "se.1.bias"),
                                   P(h, f + "se.3.weight"), P(h, f + "se.3.bias"), hid, P(h, pre + "channel_reduce.weight"),
                                   ws + p.wfold_cr, ws + p.wfold_cr3, ws + p.ch, B, C, side, composed, hc));
        r.wp = ws + p.wfold_cr; r.wp_bstride = (int64_t)packed1x1_floats(2 * C, C);
        r.wp3 = ws + p.wfold_cr3; r.wp3_bstride = (int64_t)packed1x1_b3_floats(compose ? 2 * C + hc : 2 * C, C);
    } else {
        Conv3x3Args cb{};
        cb.x = in; cb.x_bstride = (int64_t)C * Pn; cb.wp = PK(h, pre + "conv.weight"); cb.bias = P(h, pre + "conv.bias");
        cb.out = xs; cb.out_bstride = (int64_t)C * Pn; cb.B = B; cb.Cin = C; cb.Cout = C; cb.h = hh; cb.w = ww;
        cb.act = cfg.branch_lrelu ? 1 : 0;
        RF_TRY(launch_conv3x3(cb, side));
        r.wp = PK(h, pre + "channel_reduce.weight");
        r.wp3 = compose ? h->packed + h->tail3_offset[i] : PK3(h, pre + "channel_reduce.weight");
    }
    if (cfg.variant != RF_VARIANT_TRUECOLOR) {
        RF_TRY(run_transformer(tp, in, trans, ws, to, B, C, heads, hc, hh, ww, st));
        RF_TRY(join_branch(h, st, side));
    }
    if (compose) {
        r.wp = nullptr;
        r.x2 = ws + p.x1;
        r.x3 = ws + p.bufB; r.C3 = hc; r.x3_bstride = (int64_t)hc * Pn;
        r.bias = composed + (size_t)C * hc;
    }
    RF_TRY(launch_conv1x1(r, st));

    Conv3x3Args co{};
    co.x = crb; co.x_bstride = (int64_t)C * Pn; co.wp = PK(h, pre + "Conv_out.weight"); co.bias = P(h, pre + "Conv_out.bias");
    co.out = out; co.out_bstride = (int64_t)C * Pn; co.B = B; co.Cin = C; co.Cout = C; co.h = hh; co.w = ww; co.act = 1;
    if (p.ks_floats) { co.ks_scratch = ws + p.ks; co.ks_floats = p.ks_floats; }
    RF_TRY(launch_conv3x3(co, st));
    return RF_OK;
}

}  // namespace

extern "C" {

int rf_create(const rf_config* cfg, rf_handle** out) {
    RF_CHECK_ARG(cfg && out, "rf_create: null argument");
    RF_CHECK_ARG(cfg->dim > 0 && cfg->dim % 8 == 0, "rf_create: dim=%d must be a positive multiple of 8", cfg->dim);
    RF_CHECK_ARG(cfg->inp_channels == 1, "rf_create: inp_channels=%d (only the 1-channel Bayer mosaic is supported)", cfg->inp_channels);
    RF_CHECK_ARG(cfg->out_channels > 0 && cfg->ffn_expansion > 0, "rf_create: bad out_channels / ffn_expansion");
    RF_CHECK_ARG(cfg->variant == RF_VARIANT_FLCA || cfg->variant == RF_VARIANT_PLAIN || cfg->variant == RF_VARIANT_TRUECOLOR,
                 "rf_create: unknown variant %d", cfg->variant);
    RF_CHECK_ARG(cfg->flca_levels >= 0 && cfg->flca_levels <= 3, "rf_create: flca_levels=%d (1..3, 0 = default 2)", cfg->flca_levels);
    RF_CHECK_ARG(cfg->variant != RF_VARIANT_TRUECOLOR || cfg->out_channels == 3, "rf_create: the TrueColor colour head is defined for 3 output channels");
    for (int l = 0; l < 4; ++l) {
        const int C = cfg->dim << l;
        RF_CHECK_ARG(cfg->heads[l] > 0 && C % cfg->heads[l] == 0 && C / cfg->heads[l] <= 64,
                     "rf_create: heads[%d]=%d incompatible with %d channels (head size must divide and be <= 64)", l, cfg->heads[l], C);
        int ns, sl;
        size_t pf;
        RF_TRY(gram_plan(1, C, cfg->heads[l], 256, &ns, &sl, &pf));   // rejects head sizes whose query tiles straddle too many key tiles
    }
    rf_handle* h = new rf_handle();
    h->cfg = *cfg;
    const int d = cfg->dim;
    if (cfg->variant == RF_VARIANT_TRUECOLOR) {   // EnhancedBayerProcessor (BayerTORGBColorMultiLvl.py:73-98), state_dict order
        add_param(h, "bayer_processor.wb_gains", {4});
        add_param(h, "bayer_processor.color_matrix", {3, 4});
        add_param(h, "bayer_processor.demosaic_refine.0.weight", {32, 3, 3, 3});
        add_param(h, "bayer_processor.demosaic_refine.0.bias", {32});
        add_param(h, "bayer_processor.demosaic_refine.2.weight", {3, 32, 3, 3});
        add_param(h, "bayer_processor.demosaic_refine.2.bias", {3});
        add_param(h, "bayer_processor.chroma_extractor.0.weight", {16, 4, 3, 3});
        add_param(h, "bayer_processor.chroma_extractor.0.bias", {16});
        add_param(h, "bayer_processor.chroma_extractor.2.weight", {2, 16, 3, 3});
        add_param(h, "bayer_processor.chroma_extractor.2.bias", {2});
        for (const char* n : {"demosaic_refine.0.weight", "demosaic_refine.2.weight", "chroma_extractor.0.weight", "chroma_extractor.2.weight"})
            add_pack(h, std::string("bayer_processor.") + n, PK_3x3);
    }
    add_param(h, "embedding.weight", {d, 4 * cfg->inp_channels, 3, 3});
    add_param(h, "embedding.bias", {d});
    add_pack(h, "embedding.weight", PK_3x3);
    for (int i = 1; i <= 3; ++i) {
        const int C = d << (i - 1);
        add_stage(h, i, C, cfg->heads[i - 1]);
        const std::string n = "down" + std::to_string(i) + ".body.0.weight";
        add_param(h, n, {C / 2, C, 3, 3});
        add_pack(h, n, PK_3x3);
    }
    add_stage(h, 4, d * 8, cfg->heads[3]);
    for (int i = 1; i <= 3; ++i) {
        const int lvl = 3 - i, C = d << lvl;
        const std::string u = "up" + std::to_string(i), r = "channel_reduce" + std::to_string(i);
        add_param(h, u + ".weight", {2 * C, C, 2, 2});
        add_param(h, u + ".bias", {C});
        add_param(h, r + ".weight", {C, 2 * C, 1, 1});
        add_param(h, r + ".bias", {C});
        add_pack(h, u + ".weight", PK_CONVT);      // the two-kernel form stays available for widths that are not
        add_pack(h, r + ".weight", PK_1x1);        // multiples of 4 (e.g. level 3 of a 1424 x 2128 frame)
        h->upcat_offset[i - 1] = h->packed_floats;
        h->packed_floats += align_up(upcat_packed_floats(C), 64);
        add_stage(h, 4 + i, C, cfg->heads[lvl]);
    }
    add_param(h, "conv_out.weight", {4 * cfg->out_channels, d, 3, 3});
    add_param(h, "conv_out.bias", {4 * cfg->out_channels});
    add_pack(h, "conv_out.weight", PK_3x3);
    if (cfg->variant == RF_VARIANT_TRUECOLOR) {   // CameraAwareColorCorrection (BayerTORGBColorMultiLvl.py:139-158)
        add_param(h, "color_correction.gamma_param", {});
        add_param(h, "color_correction.color_transform.0.weight", {64, 3, 1, 1});
        add_param(h, "color_correction.color_transform.0.bias", {64});
        add_param(h, "color_correction.color_transform.2.weight", {3, 64, 1, 1});
        add_param(h, "color_correction.color_transform.2.bias", {3});
        add_param(h, "color_correction.tone_curve.0.weight", {32, 1, 1, 1});
        add_param(h, "color_correction.tone_curve.0.bias", {32});
        add_param(h, "color_correction.tone_curve.2.weight", {1, 32, 1, 1});
        add_param(h, "color_correction.tone_curve.2.bias", {1});
    }
    // flat layout for training: registry order, every tensor on a 16-byte boundary (a LayerNorm's weight and bias stay adjacent)
    for (const Param& q : h->params) {
        h->flat_offset.push_back(h->flat_floats);
        h->flat_floats += align_up(q.numel(), 4);
    }
    *out = h;
    return RF_OK;
}

void rf_destroy(rf_handle* h) {
    if (!h) return;
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
}

int rf_param_count(const rf_handle* h) { return h ? (int)h->params.size() : RF_E_INVALID; }

int rf_param_info(const rf_handle* h, int index, const char** name, int64_t shape[4], int* ndim) {
    RF_CHECK_ARG(h && index >= 0 && index < (int)h->params.size(), "rf_param_info: index %d out of range", index);
    const Param& p = h->params[index];
    if (name) *name = p.name.c_str();
    if (shape) std::memcpy(shape, p.shape, sizeof(p.shape));
    if (ndim) *ndim = p.ndim;
    return RF_OK;
}

int rf_set_param(rf_handle* h, const char* name, const float* dev_ptr, const int64_t* shape, int ndim) {
    RF_CHECK_ARG(h && name && dev_ptr, "rf_set_param: null argument");
    auto it = h->index.find(name);
    if (it == h->index.end()) {
        set_error("rf_set_param: unexpected key '%s'", name);
        return RF_E_MISSING;
    }
    Param& p = h->params[it->second];
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    bool same = n == p.numel();
    // accept [heads,1,1] vs [1,heads,1,1] style differences, reject anything that changes sizes
    if (same && ndim == p.ndim)
        for (int i = 0; i < ndim; ++i) same = same && shape[i] == p.shape[i];
    else if (same)
        same = p.ndim <= 1 || p.name.find("temperature") != std::string::npos;
    RF_CHECK_ARG(same, "rf_set_param: size mismatch for %s: got %zu elements in %d dims, expected %zu", name, n, ndim, p.numel());
    RF_CHECK_ARG((reinterpret_cast<uintptr_t>(dev_ptr) & 3) == 0, "rf_set_param: %s is not 4-byte aligned", name);
    p.ptr = dev_ptr;
    h->packed = nullptr;
    return RF_OK;
}

int rf_packed_bytes(const rf_handle* h, size_t* bytes) {
    RF_CHECK_ARG(h && bytes, "rf_packed_bytes: null argument");
    *bytes = h->packed_floats * sizeof(float);
    return RF_OK;
}

int rf_pack_params(rf_handle* h, void* packed_dev, size_t bytes, void* stream) {
    RF_CHECK_ARG(h && packed_dev, "rf_pack_params: null argument");
    RF_CHECK_ARG(aligned16(packed_dev), "rf_pack_params: buffer must be 16-byte aligned");
    if (bytes < h->packed_floats * sizeof(float)) {
        set_error("rf_pack_params: buffer of %zu bytes, need %zu", bytes, h->packed_floats * sizeof(float));
        return RF_E_NOMEM;
    }
    for (const Param& p : h->params)
        if (!p.ptr) {
            set_error("rf_pack_params: missing key '%s'", p.name.c_str());
            return RF_E_MISSING;
        }
    hipStream_t st = (hipStream_t)stream;
    float* base = (float*)packed_dev;
    for (const PackItem& it : h->packs) {
        const Param& p = h->params[it.param];
        int rc;
        if (it.kind == PK_1x1) rc = pack_1x1(p.ptr, base + it.offset, (int)p.shape[0], (int)p.shape[1], p.shape[1], 1, st);
        else if (it.kind == PK_1x1_B3) rc = pack_1x1_b3(p.ptr, base + it.offset, (int)p.shape[0], (int)p.shape[1], p.shape[1], 1, st);
        else if (it.kind == PK_3x3) rc = pack_3x3(p.ptr, base + it.offset, (int)p.shape[0], (int)p.shape[1], st);
        else rc = pack_convT(p.ptr, base + it.offset, (int)p.shape[0], (int)p.shape[1], st);
        if (rc) return rc;
    }
    for (int i = 1; i <= 3; ++i) {
        const int C = h->cfg.dim << (3 - i);
        const std::string u = "up" + std::to_string(i), r = "channel_reduce" + std::to_string(i);
        RF_TRY(pack_upcat(P(h, u + ".weight"), P(h, u + ".bias"), P(h, r + ".weight"), P(h, r + ".bias"), base + h->upcat_offset[i - 1], C, st));
    }
    for (int i = 1; i <= 7; ++i) {
        if (!h->tail_offset[i]) continue;
        const int lvl = i <= 4 ? i - 1 : 7 - i, C = h->cfg.dim << lvl, hc = C * h->cfg.ffn_expansion;
        const std::string pre = "conv_tran" + std::to_string(i) + ".";
        float* composed = base + h->tail_offset[i];
        RF_TRY(pack_tail(P(h, pre + "channel_reduce.weight"), P(h, pre + "channel_reduce.bias"), P(h, pre + "Transformer.ffn.pointwise2.weight"),
                         P(h, pre + "Transformer.ffn.pointwise2.bias"), composed, C, hc, st));
        if (h->tail3_offset[i])
            RF_TRY(launch_tail_fold(P(h, pre + "channel_reduce.weight"), nullptr, composed, base + h->tail3_offset[i], 1, C, hc, st));
    }
    h->packed = base;
    return RF_OK;
}

int rf_workspace_bytes(const rf_handle* h, int B, int H, int W, size_t* bytes) {
    RF_CHECK_ARG(h && bytes, "rf_workspace_bytes: null argument");
    RF_CHECK_ARG(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0, "packed size %dx%d must be positive multiples of 8 (mosaic divisible by 16)", H, W);
    Plan p;
    RF_TRY(make_plan(h, B, H, W, p));
    *bytes = p.total * sizeof(float);
    return RF_OK;
}

int rf_set_shard(rf_handle* h, int y_lo, int y_hi, int total_rows, rf_allreduce_fn allreduce, void* user) {
    RF_CHECK_ARG(h, "rf_set_shard: null handle");
    if (!allreduce) {
        h->shard_y_lo = h->shard_y_hi = h->shard_total_rows = 0;
        h->shard_allreduce = nullptr; h->shard_user = nullptr;
        return RF_OK;
    }
    RF_CHECK_ARG(h->cfg.variant != RF_VARIANT_TRUECOLOR, "rf_set_shard: variants flca and plain only");
    RF_CHECK_ARG(y_lo >= 0 && y_hi > y_lo && y_lo % 8 == 0 && y_hi % 8 == 0 && total_rows >= y_hi - y_lo && total_rows % 8 == 0,
                 "rf_set_shard: interior rows [%d, %d) of %d must be multiples of 8", y_lo, y_hi, total_rows);
    h->shard_y_lo = y_lo; h->shard_y_hi = y_hi; h->shard_total_rows = total_rows;
    h->shard_allreduce = allreduce; h->shard_user = user;
    return RF_OK;
}

int rf_forward_stage(rf_handle* h, int stage, const float* in, const float* packed, float* out, void* workspace,
                     size_t workspace_bytes, int B, int H, int W, void* stream) {
    RF_CHECK_ARG(h && in && out && workspace && stage >= 1 && stage <= 7, "rf_forward_stage: bad arguments (stage 1..7)");
    RF_CHECK_ARG(B > 0 && B <= 65535 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0,
                 "rf_forward_stage: packed size %dx%d must be positive multiples of 8", H, W);
    RF_CHECK_ARG(h->cfg.variant != RF_VARIANT_FLCA || packed, "rf_forward_stage: the FLCA branch needs the packed frame for its guidance");
    RF_CHECK_ARG(h->cfg.variant != RF_VARIANT_TRUECOLOR, "rf_forward_stage: not available for the TrueColor variant");
    if (!h->packed) {
        set_error("rf_forward_stage: parameters not packed (call rf_pack_params after rf_set_param)");
        return RF_E_MISSING;
    }
    RF_CHECK_ARG(aligned16(workspace) && aligned16(in) && aligned16(out), "rf_forward_stage: buffers must be 16-byte aligned");
    Plan p;
    RF_TRY(make_plan(h, B, H, W, p));
    if (workspace_bytes < p.total * sizeof(float)) {
        set_error("rf_forward_stage: workspace of %zu bytes, need %zu", workspace_bytes, p.total * sizeof(float));
        return RF_E_NOMEM;
    }
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const int lvl = stage <= 4 ? stage - 1 : 7 - stage;
    if (h->cfg.variant == RF_VARIANT_FLCA) {
        RF_TRY(launch_guidance_base(packed, 0, h->cfg.clamp_io, ws + p.gscratch, B, H, W, st));
        RF_TRY(launch_guidance_level(ws + p.gscratch, ws + p.guide[lvl], B, H, W, H >> lvl, W >> lvl, st));
    }
    if (p.ks_floats) RF_TRY(check_hip(hipMemsetAsync(ws + p.ks, 0, conv3x3_ksplit_counter_bytes(), st), "rf_forward_stage: memset"));
    return run_stage(h, stage, lvl, in, out, ws, p, B, H, W, st, st);
}

int rf_forward(rf_handle* h, const float* in, float* out, void* workspace, size_t workspace_bytes,
               int B, int H, int W, int packed_input, void* stream) {
    RF_CHECK_ARG(h && in && out && workspace, "rf_forward: null argument");
    RF_CHECK_ARG(B > 0 && B <= 65535 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0,
                 "rf_forward: packed size %dx%d must be positive multiples of 8 (mosaic divisible by 16)", H, W);
    RF_CHECK_ARG((size_t)H * W < (1u << 30), "rf_forward: frame too large");
    if (!h->packed) {
        set_error("rf_forward: parameters not packed (call rf_pack_params after rf_set_param)");
        return RF_E_MISSING;
    }
    RF_CHECK_ARG(aligned16(workspace) && aligned16(in) && aligned16(out), "rf_forward: buffers must be 16-byte aligned");
    RF_CHECK_ARG(!h->shard_allreduce || h->shard_y_hi <= H, "rf_forward: shard interior [%d, %d) outside the %d-row window",
                 h->shard_y_lo, h->shard_y_hi, H);
    Plan p;
    RF_TRY(make_plan(h, B, H, W, p));
    if (workspace_bytes < p.total * sizeof(float)) {
        set_error("rf_forward: workspace of %zu bytes, need %zu", workspace_bytes, p.total * sizeof(float));
        return RF_E_NOMEM;
    }
    const rf_config& cfg = h->cfg;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const int d = cfg.dim;
    const int mosaic = packed_input ? 0 : 1;

    const int levels = cfg.flca_levels > 0 ? cfg.flca_levels : 2;
    const hipStream_t side = branch_stream(h, st);
    if (p.ks_floats)      // tickets of the 3x3 convs' input-channel split (the kernels leave them zero; the workspace is the caller's)
        RF_TRY(check_hip(hipMemsetAsync(ws + p.ks, 0, conv3x3_ksplit_counter_bytes(), st), "rf_forward: memset"));
    if (cfg.variant == RF_VARIANT_FLCA) {
        // the guidance pyramid feeds the FLCA branches only: it runs on their stream, beside the embedding
        RF_TRY(fork_branch(h, st, side));
        RF_TRY(launch_guidance_base(in, mosaic, cfg.clamp_io, ws + p.gscratch, B, H, W, side, h->shard_allreduce, h->shard_user));
        for (int l = 0; l < 4; ++l)
            RF_TRY(launch_guidance_level(ws + p.gscratch, ws + p.guide[l], B, H, W, H >> l, W >> l, side));
    } else if (cfg.variant == RF_VARIANT_TRUECOLOR) {
        const std::string bp = "bayer_processor.";
        RF_TRY(launch_tc_front(in, mosaic, P(h, bp + "wb_gains"), P(h, bp + "color_matrix"),
                               PK(h, bp + "chroma_extractor.0.weight"), P(h, bp + "chroma_extractor.0.bias"),
                               PK(h, bp + "chroma_extractor.2.weight"), P(h, bp + "chroma_extractor.2.bias"),
                               PK(h, bp + "demosaic_refine.0.weight"), P(h, bp + "demosaic_refine.0.bias"),
                               PK(h, bp + "demosaic_refine.2.weight"), P(h, bp + "demosaic_refine.2.bias"),
                               ws + p.gscratch, B, H, W, levels, st));
        for (int l = 0; l < 4; ++l)
            RF_TRY(launch_tc_guide_level(ws + p.gscratch, ws + p.guide[l], B, H, W, levels, H >> l, W >> l, st));
    }
    // embedding (reads the mosaic through the Bayer pack)
    Conv3x3Args e{};
    e.x = in; e.x_bstride = (int64_t)4 * H * W; e.wp = PK(h, "embedding.weight"); e.bias = P(h, "embedding.bias");
    e.out = ws + p.tA; e.out_bstride = (int64_t)d * H * W; e.B = B; e.Cin = 4; e.Cout = d; e.h = H; e.w = W;
    e.unshuffle_in = mosaic; e.clamp_in = cfg.clamp_io;
    RF_TRY(launch_conv3x3(e, st));

    // encoder
    float* skip[3] = {ws + p.skip[0], ws + p.skip[1], ws + p.skip[2]};
    for (int i = 1; i <= 3; ++i) {
        const int lvl = i - 1, C = d << lvl, hh = H >> lvl, ww = W >> lvl;
        RF_TRY(run_stage(h, i, lvl, ws + p.tA, skip[lvl], ws, p, B, H, W, st, side));
        Conv3x3Args dn{};
        dn.x = skip[lvl]; dn.x_bstride = (int64_t)C * hh * ww; dn.wp = PK(h, "down" + std::to_string(i) + ".body.0.weight");
        dn.out = ws + p.tA; dn.out_bstride = (int64_t)2 * C * (hh / 2) * (ww / 2);
        dn.B = B; dn.Cin = C; dn.Cout = C / 2; dn.h = hh; dn.w = ww; dn.store = 1;
        if (p.ks_floats) { dn.ks_scratch = ws + p.ks; dn.ks_floats = p.ks_floats; }
        RF_TRY(launch_conv3x3(dn, st));
    }
    RF_TRY(run_stage(h, 4, 3, ws + p.tA, ws + p.tB, ws, p, B, H, W, st, side));
    // decoder
    for (int i = 1; i <= 3; ++i) {
        const int lvl = 3 - i, C = d << lvl, hh = H >> lvl, ww = W >> lvl, Pn = hh * ww;
        const std::string u = "up" + std::to_string(i), r = "channel_reduce" + std::to_string(i);
        bool fuse_up = upcat_supported(C, hh / 2, ww / 2, ws + p.tB, skip[lvl], ws + p.tA);
#ifdef RF_DIAG   // diagnostic build only (build.py --diag): force the two-kernel decoder step
        if (getenv("RF_NO_UPCAT")) fuse_up = false;
#endif
        if (fuse_up) {
            // ConvTranspose2d + cat + 1x1 as one kernel on composed weights: `up` never reaches HBM
            RF_TRY(launch_upcat(ws + p.tB, skip[lvl], ws + p.tA, h->packed + h->upcat_offset[i - 1], B, C, hh / 2, ww / 2, st));
            RF_TRY(run_stage(h, 4 + i, lvl, ws + p.tA, ws + p.tB, ws, p, B, H, W, st, side));
            continue;
        }
        Conv1x1Args up{};
        up.x1 = ws + p.tB; up.C1 = 2 * C; up.x1_bstride = (int64_t)2 * C * (Pn / 4);
        up.wp = PK(h, u + ".weight"); up.bias = P(h, u + ".bias");
        up.out = ws + p.tU; up.out_bstride = (int64_t)C * Pn; up.Cout = 4 * C; up.B = B; up.P = Pn / 4; up.w = ww / 2; up.mode = 1;
        RF_TRY(launch_conv1x1(up, st));
        Conv1x1Args cr{};
        cr.x1 = ws + p.tU; cr.C1 = C; cr.x1_bstride = (int64_t)C * Pn;
        cr.x2 = skip[lvl]; cr.C2 = C; cr.x2_bstride = (int64_t)C * Pn;
        cr.wp = PK(h, r + ".weight"); cr.bias = P(h, r + ".bias");
        cr.out = ws + p.tA; cr.out_bstride = (int64_t)C * Pn; cr.Cout = C; cr.B = B; cr.P = Pn; cr.w = ww;
        RF_TRY(launch_conv1x1(cr, st));
        RF_TRY(run_stage(h, 4 + i, lvl, ws + p.tA, ws + p.tB, ws, p, B, H, W, st, side));
    }
    // conv_out + LeakyReLU + PixelShuffle (+ clamp)
    Conv3x3Args o{};
    o.x = ws + p.tB; o.x_bstride = (int64_t)d * H * W; o.wp = PK(h, "conv_out.weight"); o.bias = P(h, "conv_out.bias");
    o.out = out; o.out_bstride = (int64_t)cfg.out_channels * 4 * H * W;
    o.B = B; o.Cin = d; o.Cout = 4 * cfg.out_channels; o.h = H; o.w = W; o.act = 1; o.store = 2; o.clamp_out = cfg.clamp_io;
    if (cfg.variant == RF_VARIANT_TRUECOLOR) o.act = 2;      // F.relu before the PixelShuffle (BayerTORGBColorMultiLvl.py:458)
    RF_TRY(launch_conv3x3(o, st));
    if (cfg.variant == RF_VARIANT_TRUECOLOR) {
        const std::string cc = "color_correction.";
        const float* prm[9] = {P(h, cc + "gamma_param"), P(h, cc + "color_transform.0.weight"), P(h, cc + "color_transform.0.bias"),
                               P(h, cc + "color_transform.2.weight"), P(h, cc + "color_transform.2.bias"), P(h, cc + "tone_curve.0.weight"),
                               P(h, cc + "tone_curve.0.bias"), P(h, cc + "tone_curve.2.weight"), P(h, cc + "tone_curve.2.bias")};
        RF_TRY(launch_tc_color_head(out, prm, B, (size_t)4 * H * W, st));
    }
    return RF_OK;
}

}  // extern "C"
