// Evaluation-harness kernels: what test.py:117-124 does on the host
// with numpy / skimage, done on the device with exact integer arithmetic.
//   to_uint8_hwc   clamp(pred, 0, 1) * 255 -> uint8 by truncation (numpy .astype(np.uint8)),
//                  CHW float -> HWC uint8 (pred[0].cpu().numpy().transpose(1, 2, 0))
//   u8_sse         sum over an image of (a - b)^2 as uint64: PSNR = 10 log10(255^2 * n / sse)
//   u8_channel_sums per-channel sums of an HWC uint8 image (auto_correct_rb compares channel means)
// Integer sums are order-independent, so the atomics here are bitwise reproducible.
#include "rf_common.h"

namespace rf {

__global__ void __launch_bounds__(256) to_uint8_hwc_kernel(const float* __restrict__ in, unsigned char* __restrict__ out,
                                                           int C, size_t hw, size_t total) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t b = i / (hw * C), r = i % (hw * C);
        const size_t p = r / C;
        const int c = (int)(r % C);
        float v = in[(b * C + c) * hw + p];
        v = fminf(fmaxf(v, 0.f), 1.f) * 255.0f;       // float32 product, as torch.clamp(...) then numpy * 255
        out[i] = (unsigned char)(int)v;               // truncation toward zero
    }
}

__global__ void __launch_bounds__(256) u8_sse_kernel(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                                     unsigned long long* __restrict__ sse, size_t n_per_image) {
    const size_t img = blockIdx.y;
    const unsigned char* pa = a + img * n_per_image;
    const unsigned char* pb = b + img * n_per_image;
    unsigned long long s = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_per_image; i += (size_t)gridDim.x * 256) {
        const int d = (int)pa[i] - (int)pb[i];
        s += (unsigned long long)(d * d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(sse + img, s);
}

__global__ void __launch_bounds__(256) u8_channel_sums_kernel(const unsigned char* __restrict__ a, unsigned long long* __restrict__ sums,
                                                              int C, size_t hw) {
    const size_t img = blockIdx.y;
    const unsigned char* pa = a + img * hw * C;
    for (int c = 0; c < C; ++c) {
        unsigned long long s = 0;
        for (size_t p = blockIdx.x * 256ull + threadIdx.x; p < hw; p += (size_t)gridDim.x * 256) s += pa[p * C + c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(sums + img * C + c, s);
    }
}

}  // namespace rf

using namespace rf;

// ------------------------------------------------------------------------------------------
// SID front-end (correctdataloader.py:58-72, 86, 103): uint16 Bayer frame -> normalised, amplified
// float planes, fused into one pass (2 B read + 4 B written per sensor pixel):
//   v = clip((raw - black) / (white - black), 0, 1);  v = min(v * ratio, 1)
// evaluated in double and rounded once, which is what the reference's numpy expressions do under
// NumPy >= 2 (float32 array - np.int64 scalar promotes to float64; the loader casts to float32 last).
//   mode 0: the loader's packing   [B,4,h,w], channels (0,0) (0,1) (1,1) (1,0)   = R G B G
//   mode 1: pixel_unshuffle order  [B,4,h,w], channels (0,0) (0,1) (1,0) (1,1)   = R G1 G2 B (a1)
//   mode 2: the normalised mosaic  [B,1,2h,2w]  (what RawFormer.forward takes)
// A thread owns a 2 x 8 block of the mosaic (two 16-byte loads, four 16-byte stores).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sid_norm(unsigned v, double black, double inv_unused, double denom, double ratio) {
    double t = ((double)v - black) / denom;
    t = fmin(fmax(t, 0.0), 1.0);
    return (float)fmin(t * ratio, 1.0);
}

__global__ void __launch_bounds__(256) sid_pack_kernel(const unsigned short* __restrict__ raw, float* __restrict__ out,
                                                       int B, int h, int w, double black, double denom, double ratio, int mode) {
    const int w2 = 2 * w;
    const int gw = w / 4;                                  // blocks of 8 mosaic columns = 4 packed columns
    const size_t items = (size_t)B * h * gw;
    for (size_t it = blockIdx.x * (size_t)256 + threadIdx.x; it < items; it += (size_t)gridDim.x * 256) {
        const int gx = (int)(it % gw);
        const int y = (int)((it / gw) % h);
        const size_t b = it / ((size_t)gw * h);
        const unsigned short* r0 = raw + (b * 2 * h + 2 * y) * (size_t)w2 + 8 * gx;
        const uint4 e = *reinterpret_cast<const uint4*>(r0);            // even row, 8 pixels
        const uint4 o = *reinterpret_cast<const uint4*>(r0 + w2);       // odd row
        const unsigned ew[4] = {e.x, e.y, e.z, e.w}, ow[4] = {o.x, o.y, o.z, o.w};
        float p00[4], p01[4], p10[4], p11[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p00[q] = sid_norm(ew[q] & 0xffffu, black, 0.0, denom, ratio);
            p01[q] = sid_norm(ew[q] >> 16, black, 0.0, denom, ratio);
            p10[q] = sid_norm(ow[q] & 0xffffu, black, 0.0, denom, ratio);
            p11[q] = sid_norm(ow[q] >> 16, black, 0.0, denom, ratio);
        }
        if (mode == 2) {
            float* m0 = out + (b * 2 * h + 2 * y) * (size_t)w2 + 8 * gx;
            *reinterpret_cast<float4*>(m0) = make_float4(p00[0], p01[0], p00[1], p01[1]);
            *reinterpret_cast<float4*>(m0 + 4) = make_float4(p00[2], p01[2], p00[3], p01[3]);
            *reinterpret_cast<float4*>(m0 + w2) = make_float4(p10[0], p11[0], p10[1], p11[1]);
            *reinterpret_cast<float4*>(m0 + w2 + 4) = make_float4(p10[2], p11[2], p10[3], p11[3]);
        } else {
            const size_t P = (size_t)h * w;
            float* ob = out + b * 4 * P + (size_t)y * w + 4 * gx;
            *reinterpret_cast<float4*>(ob) = make_float4(p00[0], p00[1], p00[2], p00[3]);
            *reinterpret_cast<float4*>(ob + P) = make_float4(p01[0], p01[1], p01[2], p01[3]);
            const float* c2 = mode == 0 ? p11 : p10;
            const float* c3 = mode == 0 ? p10 : p11;
            *reinterpret_cast<float4*>(ob + 2 * P) = make_float4(c2[0], c2[1], c2[2], c2[3]);
            *reinterpret_cast<float4*>(ob + 3 * P) = make_float4(c3[0], c3[1], c3[2], c3[3]);
        }
    }
}

extern "C" {

int rf_to_uint8_hwc(const float* in, unsigned char* out, int B, int C, int h, int w, void* stream) {
    RF_CHECK_ARG(in && out && B > 0 && C > 0 && h > 0 && w > 0, "to_uint8_hwc: bad arguments");
    const size_t hw = (size_t)h * w, total = (size_t)B * C * hw;
    int g = (int)((total + 255) / 256);
    if (g > 8192) g = 8192;
    ProfScope prof((hipStream_t)stream, "to_uint8_hwc_kernel", 0.0, 5.0 * total);
    to_uint8_hwc_kernel<<<g, 256, 0, (hipStream_t)stream>>>(in, out, C, hw, total);
    return check_launch("to_uint8_hwc");
}

int rf_u8_sse(const unsigned char* a, const unsigned char* b, unsigned long long* sse, int B, size_t n_per_image, void* stream) {
    RF_CHECK_ARG(a && b && sse && B > 0 && B <= 65535 && n_per_image > 0, "u8_sse: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int rc = check_hip(hipMemsetAsync(sse, 0, sizeof(unsigned long long) * B, st), "u8_sse memset");
    if (rc) return rc;
    int g = (int)((n_per_image + 255) / 256);
    if (g > 1024) g = 1024;
    u8_sse_kernel<<<dim3((unsigned)g, (unsigned)B), 256, 0, st>>>(a, b, sse, n_per_image);
    return check_launch("u8_sse");
}

int rf_u8_channel_sums(const unsigned char* a, unsigned long long* sums, int B, int C, size_t hw, void* stream) {
    RF_CHECK_ARG(a && sums && B > 0 && B <= 65535 && C > 0 && hw > 0, "u8_channel_sums: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int rc = check_hip(hipMemsetAsync(sums, 0, sizeof(unsigned long long) * B * C, st), "u8_channel_sums memset");
    if (rc) return rc;
    int g = (int)((hw + 255) / 256);
    if (g > 1024) g = 1024;
    u8_channel_sums_kernel<<<dim3((unsigned)g, (unsigned)B), 256, 0, st>>>(a, sums, C, hw);
    return check_launch("u8_channel_sums");
}

int rf_sid_pack(const unsigned short* raw, float* out, int B, int h, int w, int black, int white, double ratio, int mode, void* stream) {
    RF_CHECK_ARG(raw && out && B > 0 && h > 0 && w > 0, "sid_pack: bad arguments");
    RF_CHECK_ARG(mode >= 0 && mode <= 2, "sid_pack: mode %d", mode);
    RF_CHECK_ARG(white > black, "sid_pack: white level %d <= black level %d", white, black);
    RF_CHECK_ARG(w % 4 == 0 && aligned16(raw) && aligned16(out), "sid_pack: packed width %d must be a multiple of 4 and the buffers 16-byte aligned", w);
    hipStream_t st = (hipStream_t)stream;
    const size_t items = (size_t)B * h * (w / 4);
    size_t gx = (items + 255) / 256;
    if (gx > 8192) gx = 8192;
    ProfScope prof(st, "sid_pack_kernel", 0.0, 6.0 * 4.0 * B * h * w);
    sid_pack_kernel<<<dim3((unsigned)gx), 256, 0, st>>>(raw, out, B, h, w, (double)black, (double)(white - black), ratio, mode);
    return check_launch("sid_pack");
}
}  // extern "C"
