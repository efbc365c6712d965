// FEB / ProcessBlock / FFAB host schedules (RawFomer_WFB_FFAB/blocks.py:11-92) on the kernels of rf_fft.hip and
// rf_gemm1x1.hip, and the Mamba-free wavelet branch of WMB (RawFomer_WFB_FFAB/model.py:215-245).  Host code only.
//
//   FEB(x)          = clamp( irfft2( m e^{i p} ) + xc ),  xc = clamp(x, +-10),  (|F| + 1e-6, angle F) = polar(rfft2(fpre(xc))),
//                     m = clamp(process1(|F| + 1e-6), 0, 1e4),  p = process2(angle F);  process = 1x1 -> LeakyReLU(0.1) -> 1x1
//   ProcessBlock(x) = cat(FEB(x)) + x                     (cat = 1x1)
//   FFAB(x)         = seven ProcessBlocks with dense concatenations (blocks.py:83-92)
// Kernel launches per FEB: clamp, 1x1, 2 (rfft2 + polar), 4 x 1x1 (activation / clamp in the epilogue), 2 (polar + irfft2 with
// the residual and the final clamp in its epilogue).
#include <cmath>
#include "rf_common.h"

namespace rf {

#define RF_TRY(expr)            \
    do {                        \
        const int rc_ = (expr); \
        if (rc_) return rc_;    \
    } while (0)

namespace {

struct Bump {
    float* base;
    size_t off = 0;
    float* take(size_t floats) {
        float* p = base ? base + off : nullptr;
        off += align_up(floats, 64);
        return p;
    }
};

// one nn.Conv2d(cin, cout, 1) on [B, cin, P] planes with raw weights (packed on the fly into `wpack`)
int conv1x1_raw(const float* x, float* out, const float* w, const float* bias, const float* res, float* wpack, int B, int cin, int cout,
                int P, int act, hipStream_t st) {
    RF_TRY(pack_1x1(w, wpack, cout, cin, cin, 1, st));
    Conv1x1Args a{};
    a.x1 = x; a.C1 = cin; a.x1_bstride = (int64_t)cin * P;
    a.wp = wpack; a.bias = bias; a.res = res; a.res_bstride = (int64_t)cout * P;
    a.out = out; a.out_bstride = (int64_t)cout * P; a.Cout = cout; a.B = B; a.P = P; a.w = P; a.act = act;
    return launch_conv1x1(a, st);
}

struct FebBufs { float *xc, *y, *mag, *pha, *t, *mag2, *pha2, *wpack; float2* cplx; };

size_t feb_plan(Bump& b, FebBufs& f, int B, int nc, int h, int w) {
    const size_t U = (size_t)B * nc * h * w, Uf = (size_t)B * nc * h * (w / 2 + 1);
    f.xc = b.take(U); f.y = b.take(U);
    f.mag = b.take(Uf); f.pha = b.take(Uf); f.t = b.take(Uf); f.mag2 = b.take(Uf); f.pha2 = b.take(Uf);
    f.cplx = reinterpret_cast<float2*>(b.take(2 * Uf));
    f.wpack = b.take(packed1x1_floats(nc, nc));
    return b.off;
}

// prm: fpre.{w,b}, process1.0.{w,b}, process1.2.{w,b}, process2.0.{w,b}, process2.2.{w,b}
int feb_forward(const float* x, float* out, const float* const* prm, const FebBufs& f, int B, int nc, int h, int w, hipStream_t st) {
    const int P = h * w, Pf = h * (w / 2 + 1);
    RF_TRY(launch_clamp(x, f.xc, (size_t)B * nc * P, -10.f, 10.f, st));
    RF_TRY(conv1x1_raw(f.xc, f.y, prm[0], prm[1], nullptr, f.wpack, B, nc, nc, P, 0, st));
    RF_TRY(launch_rfft2_polar(f.y, f.mag, f.pha, f.cplx, B * nc, h, w, st));
    RF_TRY(conv1x1_raw(f.mag, f.t, prm[2], prm[3], nullptr, f.wpack, B, nc, nc, Pf, 3, st));
    RF_TRY(conv1x1_raw(f.t, f.mag2, prm[4], prm[5], nullptr, f.wpack, B, nc, nc, Pf, 4, st));
    RF_TRY(conv1x1_raw(f.pha, f.t, prm[6], prm[7], nullptr, f.wpack, B, nc, nc, Pf, 3, st));
    RF_TRY(conv1x1_raw(f.t, f.pha2, prm[8], prm[9], nullptr, f.wpack, B, nc, nc, Pf, 0, st));
    return launch_polar_irfft2(f.mag2, f.pha2, f.xc, out, f.cplx, B * nc, h, w, 10.f, st);
}

// prm: the 10 FEB tensors, then cat.{w,b};  out = cat(FEB(x)) + x   (out may not alias x)
int process_block_forward(const float* x, float* out, const float* const* prm, const FebBufs& f, float* feb_out, int B, int nc, int h, int w,
                          hipStream_t st) {
    RF_TRY(feb_forward(x, feb_out, prm, f, B, nc, h, w, st));
    return conv1x1_raw(feb_out, out, prm[10], prm[11], x, f.wpack, B, nc, nc, h * w, 0, st);
}

int check_geometry(const char* what, int B, int nc, int h, int w) {
    RF_CHECK_ARG(B > 0 && nc > 0 && nc % 4 == 0 && h >= 2 && w >= 2 && w % 2 == 0, "%s: needs channels %% 4 == 0 and an even width (B=%d nc=%d %dx%d)",
                 what, B, nc, h, w);
    return RF_OK;
}

struct FfabPlan { float *x, *x1, *x2, *x3, *x4, *x5, *cat, *pb, *feb_out, *wpack2; FebBufs f; size_t total; };

void ffab_plan(float* base, FfabPlan& p, int B, int nc, int h, int w) {
    Bump b{base};
    const size_t U = (size_t)B * nc * h * w;
    p.x = b.take(U); p.x1 = b.take(U); p.x2 = b.take(U); p.x3 = b.take(U); p.x4 = b.take(U); p.x5 = b.take(U);
    p.cat = b.take(2 * U); p.pb = b.take(2 * U); p.feb_out = b.take(2 * U);
    p.wpack2 = b.take(packed1x1_floats(2 * nc, 2 * nc));
    feb_plan(b, p.f, B, 2 * nc, h, w);
    p.f.wpack = p.wpack2;
    p.total = b.off;
}

// torch.cat((a, b), dim=1) for [B, nc, P] planes
__global__ void __launch_bounds__(256) cat2_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t per_image, size_t total) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t img = i / (2 * per_image), r = i - img * 2 * per_image;
        out[i] = r < per_image ? a[img * per_image + r] : b[img * per_image + r - per_image];
    }
}

int launch_cat2(const float* a, const float* b, float* out, int B, size_t per_image, hipStream_t st) {
    const size_t total = 2 * per_image * B;
    int gx = (int)((total + 255) / 256); if (gx > 4096) gx = 4096;
    cat2_kernel<<<gx, 256, 0, st>>>(a, b, out, per_image, total);
    return check_launch("cat2");
}

// out = add + clamp(in * scale + shift, lo, hi): the tail of the wavelet branch of WMB (inverse_data_transform + residual,
// RawFomer_WFB_FFAB/model.py:13-15, 241-243)
__global__ void __launch_bounds__(256) affine_clamp_add_kernel(const float* __restrict__ in, const float* __restrict__ add, float* __restrict__ out,
                                                               size_t n, float scale, float shift, float lo, float hi) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = add[i] + fminf(fmaxf(fmaf(in[i], scale, shift), lo), hi);
}

}  // namespace
}  // namespace rf

using namespace rf;

extern "C" {

int rf_rfft2_polar_scratch_bytes(int planes, int h, int w, size_t* bytes) {
    RF_CHECK_ARG(bytes && planes > 0 && h > 0 && w > 0, "rfft2_polar_scratch_bytes: bad arguments");
    *bytes = (size_t)planes * h * (w / 2 + 1) * sizeof(float2);
    return RF_OK;
}

int rf_rfft2_polar(const float* in, float* mag, float* pha, void* scratch, int planes, int h, int w, void* stream) {
    RF_CHECK_ARG(in && mag && pha && scratch && planes > 0, "rfft2_polar: bad arguments");
    return launch_rfft2_polar(in, mag, pha, (float2*)scratch, planes, h, w, (hipStream_t)stream);
}

int rf_polar_irfft2(const float* mag, const float* pha, float* out, void* scratch, int planes, int h, int w, void* stream) {
    RF_CHECK_ARG(mag && pha && out && scratch && planes > 0, "polar_irfft2: bad arguments");
    return launch_polar_irfft2(mag, pha, nullptr, out, (float2*)scratch, planes, h, w, INFINITY, (hipStream_t)stream);
}

int rf_feb_scratch_bytes(int B, int nc, int h, int w, size_t* bytes) {
    RF_CHECK_ARG(bytes, "feb_scratch_bytes: null argument");
    RF_TRY(check_geometry("feb", B, nc, h, w));
    Bump b{nullptr};
    FebBufs f;
    *bytes = feb_plan(b, f, B, nc, h, w) * sizeof(float);
    return RF_OK;
}

int rf_feb(const float* in, float* out, const float* const* prm, void* scratch, int B, int nc, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && prm && scratch && aligned16(scratch) && in != out, "feb: bad arguments");
    RF_TRY(check_geometry("feb", B, nc, h, w));
    for (int i = 0; i < 10; ++i) RF_CHECK_ARG(prm[i] != nullptr, "feb: parameter %d is null", i);
    Bump b{(float*)scratch};
    FebBufs f;
    feb_plan(b, f, B, nc, h, w);
    return feb_forward(in, out, prm, f, B, nc, h, w, (hipStream_t)stream);
}

int rf_affine_clamp_add(const float* in, const float* add, float* out, size_t n, float scale, float shift, float lo, float hi, void* stream) {
    RF_CHECK_ARG(in && add && out, "affine_clamp_add: bad arguments");
    int gx = (int)((n + 255) / 256); if (gx > 4096) gx = 4096; if (gx < 1) gx = 1;
    affine_clamp_add_kernel<<<gx, 256, 0, (hipStream_t)stream>>>(in, add, out, n, scale, shift, lo, hi);
    return check_launch("affine_clamp_add");
}

int rf_ffab_scratch_bytes(int B, int nc, int h, int w, size_t* bytes) {
    RF_CHECK_ARG(bytes, "ffab_scratch_bytes: null argument");
    RF_TRY(check_geometry("ffab", B, nc, h, w));
    FfabPlan p;
    ffab_plan(nullptr, p, B, nc, h, w);
    *bytes = p.total * sizeof(float);
    return RF_OK;
}

int rf_ffab(const float* in, float* out, const float* const* prm, void* scratch, int B, int nc, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && prm && scratch && aligned16(scratch), "ffab: bad arguments");
    RF_TRY(check_geometry("ffab", B, nc, h, w));
    for (int i = 0; i < 92; ++i) RF_CHECK_ARG(prm[i] != nullptr, "ffab: parameter %d is null", i);
    hipStream_t st = (hipStream_t)stream;
    FfabPlan p;
    ffab_plan((float*)scratch, p, B, nc, h, w);
    const int P = h * w;
    const size_t per = (size_t)nc * P;
    // state_dict order: conv0.0.{w,b}, conv0.1.<12>, conv1.<12>, conv2.<12>, conv3.<12>, conv4.0.<12>, conv4.1.{w,b},
    //                   conv5.0.<12>, conv5.1.{w,b}, convout.0.<12>, convout.1.{w,b}
    const float* const* q = prm;
    RF_TRY(conv1x1_raw(in, p.pb, q[0], q[1], nullptr, p.wpack2, B, nc, nc, P, 0, st));
    RF_TRY(process_block_forward(p.pb, p.x, q + 2, p.f, p.feb_out, B, nc, h, w, st));
    RF_TRY(process_block_forward(p.x, p.x1, q + 14, p.f, p.feb_out, B, nc, h, w, st));
    RF_TRY(process_block_forward(p.x1, p.x2, q + 26, p.f, p.feb_out, B, nc, h, w, st));
    RF_TRY(process_block_forward(p.x2, p.x3, q + 38, p.f, p.feb_out, B, nc, h, w, st));
    auto tail = [&](const float* a, const float* b, const float* const* pp, float* dst) -> int {
        RF_TRY(launch_cat2(a, b, p.cat, B, per, st));
        RF_TRY(process_block_forward(p.cat, p.pb, pp, p.f, p.feb_out, B, 2 * nc, h, w, st));
        return conv1x1_raw(p.pb, dst, pp[12], pp[13], nullptr, p.wpack2, B, 2 * nc, nc, P, 0, st);
    };
    RF_TRY(tail(p.x2, p.x3, q + 50, p.x4));
    RF_TRY(tail(p.x1, p.x4, q + 64, p.x5));
    return tail(p.x, p.x5, q + 78, out);
}

}  // extern "C"
