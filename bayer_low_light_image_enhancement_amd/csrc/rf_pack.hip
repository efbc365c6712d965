// Weight repacking into v_mfma_f32_16x16x4_f32 A-operand order (layouts: rf_common.h).
// Runs once per parameter load (rf_pack_params), not per forward.
#include "rf_common.h"

namespace rf {

__global__ void __launch_bounds__(256) pack_1x1_kernel(const float* __restrict__ w, float* __restrict__ packed,
                                                       int Cout, int K, int64_t row_stride, int64_t col_stride) {
    const int NT = (Cout + 15) >> 4, NS = (K + 3) >> 2;
    const size_t total = (size_t)NT * NS * 64;
    for (size_t idx = blockIdx.x * 256ull + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int l = (int)(idx & 63);
        const int t = (int)((idx >> 6) % NT);
        const int s = (int)((idx >> 6) / NT);
        const int co = 16 * t + (l & 15), k = 4 * s + (l >> 4);
        packed[idx] = (co < Cout && k < K) ? w[co * row_stride + k * col_stride] : 0.f;
    }
}

int pack_1x1(const float* w, float* packed, int Cout, int K, int64_t row_stride, int64_t col_stride, hipStream_t st) {
    const size_t total = packed1x1_floats(K, Cout);
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    pack_1x1_kernel<<<g, 256, 0, st>>>(w, packed, Cout, K, row_stride, col_stride);
    return check_launch("pack_1x1");
}

// b3 form of the same matrix (rf_common.h): three bf16 pieces per weight, zero outside [Cout] x [K]
__global__ void __launch_bounds__(256) pack_1x1_b3_kernel(const float* __restrict__ w, unsigned short* __restrict__ packed3,
                                                          int Cout, int K, int64_t row_stride, int64_t col_stride) {
    const int NT = (Cout + 15) >> 4, NB = (K + 31) >> 5;
    const size_t total = (size_t)NT * NB * 512;          // (co, k) pairs of the padded matrix
    for (size_t idx = blockIdx.x * 256ull + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int k = (int)(idx % ((size_t)NB * 32)), co = (int)(idx / ((size_t)NB * 32));
        b3_store(packed3, NT, co, k, (co < Cout && k < K) ? w[co * row_stride + k * col_stride] : 0.f);
    }
}

int pack_1x1_b3(const float* w, void* packed3, int Cout, int K, int64_t row_stride, int64_t col_stride, hipStream_t st) {
    const size_t total = (size_t)cdiv(Cout, 16) * cdiv(K, 32) * 512;
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    pack_1x1_b3_kernel<<<g, 256, 0, st>>>(w, (unsigned short*)packed3, Cout, K, row_stride, col_stride);
    return check_launch("pack_1x1_b3");
}

// nn.ConvTranspose2d weight [Cin][Cout][2][2]: GEMM row 4*o + 2*i + j, column k
int pack_convT(const float* w, float* packed, int Cin, int Cout, hipStream_t st) {
    return pack_1x1(w, packed, 4 * Cout, Cin, 1, (int64_t)4 * Cout, st);
}

// 3x3 weights in Winograd F(4,3) form along x (rf_conv3x3.hip): per kernel row dy the three taps g0 g1 g2 become
//   u = (g0/4, -(g0+g1+g2)/6, -(g0-g1+g2)/6, g0/24+g1/12+g2/6, g0/24-g1/12+g2/6, g2);  entry index = dy * 6 + j.
__global__ void __launch_bounds__(256) pack_3x3_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin) {
    const int NT = (Cout + 15) >> 4, NS = ((Cin + 7) >> 3) * 2;
    const size_t total = (size_t)NS * 18 * NT * 64;
    for (size_t idx = blockIdx.x * 256ull + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int l = (int)(idx & 63);
        const int t = (int)((idx >> 6) % NT);
        const int e = (int)(((idx >> 6) / NT) % 18);
        const int s = (int)((idx >> 6) / ((size_t)NT * 18));
        const int co = 16 * t + (l & 15), ci = 4 * s + (l >> 4);
        float u = 0.f;
        if (co < Cout && ci < Cin) {
            const float* g = w + ((size_t)co * Cin + ci) * 9 + (e / 6) * 3;
            const int j = e % 6;
            const float s02 = g[0] + g[2];
            const float e02 = fmaf(g[0], 1.0f / 24.0f, g[2] * (1.0f / 6.0f));
            u = j == 0 ? 0.25f * g[0] : j == 1 ? -(s02 + g[1]) * (1.0f / 6.0f) : j == 2 ? -(s02 - g[1]) * (1.0f / 6.0f)
              : j == 3 ? fmaf(g[1], 1.0f / 12.0f, e02) : j == 4 ? fmaf(g[1], -1.0f / 12.0f, e02) : g[2];
        }
        packed[idx] = u;
    }
}

// ---- batched form (training step): every packed / transposed / flipped weight of the step in a few launches ----------------
// kind 0: 1x1 generic, element (row, col) = src[row * rs + col * cs]           (pack_1x1 layout; transposes are strides)
// kind 1: 3x3 Winograd, element (row, col, tap) = src[row * rs + col * cs + (flip ? 8 - tap : tap)]   (pack_3x3 layout;
//         the dX weights of a 3x3 conv = rows and columns exchanged, taps flipped)
// kind 2: depthwise taps flipped, dst[c * 9 + 8 - t] = src[c * 9 + t]
// kind 3: kind 0's matrix in b3 form (pack_1x1_b3 layout)
__device__ __forceinline__ size_t pack_item_total(const PackDesc& d) {
    if (d.kind == 0) return (size_t)((d.rows + 15) >> 4) * ((d.cols + 3) >> 2) * 64;
    if (d.kind == 1) return (size_t)(((d.cols + 7) >> 3) * 2) * 18 * ((d.rows + 15) >> 4) * 64;
    if (d.kind == 3) return (size_t)((d.rows + 15) >> 4) * ((d.cols + 31) >> 5) * 512;      // (row, col) pairs of the padded matrix
    return (size_t)d.rows * 9;
}

__global__ void __launch_bounds__(256) pack_batch_kernel(PackBatch batch) {
    const PackDesc& d = batch.d[blockIdx.y];
    const size_t total = pack_item_total(d);
    const float* __restrict__ w = d.src;
    float* __restrict__ packed = d.dst;
    for (size_t idx = blockIdx.x * 256ull + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        if (d.kind == 0) {
            const int NT = (d.rows + 15) >> 4;
            const int l = (int)(idx & 63), t = (int)((idx >> 6) % NT), s = (int)((idx >> 6) / NT);
            const int co = 16 * t + (l & 15), k = 4 * s + (l >> 4);
            packed[idx] = (co < d.rows && k < d.cols) ? w[co * d.rs + k * d.cs] : 0.f;
        } else if (d.kind == 1) {
            const int NT = (d.rows + 15) >> 4;
            const int l = (int)(idx & 63), t = (int)((idx >> 6) % NT), e = (int)(((idx >> 6) / NT) % 18), s = (int)((idx >> 6) / ((size_t)NT * 18));
            const int co = 16 * t + (l & 15), ci = 4 * s + (l >> 4);
            float u = 0.f;
            if (co < d.rows && ci < d.cols) {
                const float* g = w + co * d.rs + ci * d.cs;
                const int dy = e / 6, j = e % 6;
                const float g0 = d.flip ? g[8 - 3 * dy] : g[3 * dy], g1 = d.flip ? g[7 - 3 * dy] : g[3 * dy + 1], g2 = d.flip ? g[6 - 3 * dy] : g[3 * dy + 2];
                const float s02 = g0 + g2;
                const float e02 = fmaf(g0, 1.0f / 24.0f, g2 * (1.0f / 6.0f));
                u = j == 0 ? 0.25f * g0 : j == 1 ? -(s02 + g1) * (1.0f / 6.0f) : j == 2 ? -(s02 - g1) * (1.0f / 6.0f)
                  : j == 3 ? fmaf(g1, 1.0f / 12.0f, e02) : j == 4 ? fmaf(g1, -1.0f / 12.0f, e02) : g2;
            }
            packed[idx] = u;
        } else if (d.kind == 3) {
            const int NT = (d.rows + 15) >> 4, NB = (d.cols + 31) >> 5;
            const int k = (int)(idx % ((size_t)NB * 32)), co = (int)(idx / ((size_t)NB * 32));
            b3_store(reinterpret_cast<unsigned short*>(packed), NT, co, k, (co < d.rows && k < d.cols) ? w[co * d.rs + k * d.cs] : 0.f);
        } else {
            const int t = (int)(idx % 9);
            packed[idx - t + 8 - t] = w[idx];
        }
    }
}

size_t pack_desc_floats(const PackDesc& d) {
    if (d.kind == 0) return packed1x1_floats(d.cols, d.rows);
    if (d.kind == 1) return packed3x3_floats(d.cols, d.rows);
    if (d.kind == 3) return packed1x1_b3_floats(d.cols, d.rows);
    return (size_t)d.rows * 9;
}

int launch_pack_batch(const PackDesc* d, int n, hipStream_t st) {
    for (int i0 = 0; i0 < n; i0 += kPackBatch) {
        PackBatch b;
        const int m = n - i0 < kPackBatch ? n - i0 : kPackBatch;
        size_t biggest = 0;
        for (int i = 0; i < m; ++i) {
            b.d[i] = d[i0 + i];
            const size_t t = pack_desc_floats(d[i0 + i]);
            if (t > biggest) biggest = t;
        }
        int gx = (int)((biggest + 255) / 256);
        if (gx > 256) gx = 256;
        pack_batch_kernel<<<dim3((unsigned)gx, (unsigned)m), 256, 0, st>>>(b);
    }
    return check_launch("pack_batch");
}

int pack_3x3(const float* w, float* packed, int Cout, int Cin, hipStream_t st) {
    const size_t total = packed3x3_floats(Cin, Cout);
    int g = (int)((total + 255) / 256);
    if (g > 4096) g = 4096;
    pack_3x3_kernel<<<g, 256, 0, st>>>(w, packed, Cout, Cin);
    return check_launch("pack_3x3");
}

}  // namespace rf
