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

}  // extern "C"
