// Microbenchmark (GPU box): do f32 MFMA (v_mfma_f32_16x16x4_f32) and f32 VALU work overlap on one SIMD of gfx950?
// One workgroup of 256 or 512 threads per CU = 1 or 2 waves per SIMD.  mode per wave: 0 = MFMA stream, 1 = VALU fma
// stream, 2 = interleaved (1 MFMA + K VALU), 3 = idle.  Prints cycles (s_memtime) of wave 0 / wave 4 of workgroup 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int K>
__device__ __forceinline__ void body_mix(f32x4 (&acc)[8], float (&v)[8], float a, float b) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < K; ++k) v[(i + k) & 7] = fmaf(v[(i + k) & 7], a, b);
    }
}

__global__ void __launch_bounds__(512) k(int modeA, int modeB, int iters, int kmix, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    const int mode = wave < 4 ? modeA : modeB;
    f32x4 acc[8];
    float v[8];
    for (int i = 0; i < 8; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; v[i] = (float)threadIdx.x * 1e-3f + i; }
    const float a = 1.0f + 1e-6f * threadIdx.x, b = 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
        for (int it = 0; it < iters; ++it) body_mix<0>(acc, v, a, b);
    } else if (mode == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], a, b);
        }
    } else if (mode == 2) {
        if (kmix == 2) for (int it = 0; it < iters; ++it) body_mix<2>(acc, v, a, b);
        else if (kmix == 4) for (int it = 0; it < iters; ++it) body_mix<4>(acc, v, a, b);
        else if (kmix == 6) for (int it = 0; it < iters; ++it) body_mix<6>(acc, v, a, b);
        else for (int it = 0; it < iters; ++it) body_mix<8>(acc, v, a, b);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    if (s == 123.456f) sink[threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

int main() {
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, 64);
    hipMalloc(&sink, 4096);
    const int iters = 2000;
    struct Case { const char* name; int threads, mA, mB, kmix; };
    std::vector<Case> cases = {
        {"1 wave/SIMD  MFMA only (8 MFMA/iter)", 256, 0, 3, 0},
        {"1 wave/SIMD  VALU only (64 fma/iter)", 256, 1, 3, 0},
        {"2 waves/SIMD MFMA | MFMA", 512, 0, 0, 0},
        {"2 waves/SIMD VALU | VALU", 512, 1, 1, 0},
        {"2 waves/SIMD MFMA | VALU", 512, 0, 1, 0},
        {"2 waves/SIMD MFMA | idle", 512, 0, 3, 0},
        {"2 waves/SIMD VALU | idle", 512, 1, 3, 0},
        {"1 wave/SIMD  mix 8 MFMA + 16 fma", 256, 2, 3, 2},
        {"1 wave/SIMD  mix 8 MFMA + 32 fma", 256, 2, 3, 4},
        {"1 wave/SIMD  mix 8 MFMA + 48 fma", 256, 2, 3, 6},
        {"1 wave/SIMD  mix 8 MFMA + 64 fma", 256, 2, 3, 8},
        {"2 waves/SIMD mix(32) | mix(32)", 512, 2, 2, 4},
        {"2 waves/SIMD mix(64) | mix(64)", 512, 2, 2, 8},
    };
    for (auto& c : cases) {
        unsigned long long h[8] = {0};
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d, 0, 64);
            hipLaunchKernelGGL(k, dim3(256), dim3(c.threads), 0, 0, c.mA, c.mB, iters, c.kmix, d, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("%-40s wave0 %8.1f cyc/iter   wave4 %8.1f cyc/iter\n", c.name, (double)h[0] / iters, (double)h[4] / iters);
    }
    return 0;
}
