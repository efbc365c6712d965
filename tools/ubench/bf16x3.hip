// Microbenchmark (GPU box): (1) does the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16) run beside f32 VALU work of the
// sibling wave?  (2) accuracy of a 3-way bf16 split (6 cross terms, f32 accumulate) of an f32 dot product against the
// f32 MFMA chain and an f64 host reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(512) overlap(int modeA, int modeB, int iters, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6;
    const int mode = wave < 4 ? modeA : modeB;
    f32x4 acc[8];
    float v[8];
    for (int i = 0; i < 8; ++i) { acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; v[i] = (float)threadIdx.x * 1e-3f + i; }
    const float a = 1.0f + 1e-6f * threadIdx.x, b = 1e-3f;
    u32x4 ua = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    bf16x8 pa = __builtin_bit_cast(bf16x8, ua), pb = pa;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {          // 8 bf16 MFMAs per iteration
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[i], 0, 0, 0);
    } else if (mode == 1) {   // 64 fma per iteration
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], a, b);
    } else if (mode == 2) {   // 8 bf16 MFMAs + 32 fma interleaved in one wave
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, pb, acc[i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[(i + k) & 7] = fmaf(v[(i + k) & 7], a, b);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
    if (s == 123.456f) sink[threadIdx.x] = s;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

// D[16][16] = A[16][K] * B[K][16], one wave.  f32 chain (16x16x4) and bf16x3 (16x16x32, 6 terms).
__device__ __forceinline__ void split3(float x, unsigned& p0, unsigned& p1, unsigned& p2) {   // truncation split: x = p0 + p1 + p2 exactly
    const unsigned u = __float_as_uint(x);
    p0 = u & 0xffff0000u;
    const float r1 = x - __uint_as_float(p0);
    p1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(p1);
    p2 = __float_as_uint(r2);       // high half taken when packed (<= 8 significant bits left)
}
__device__ __forceinline__ unsigned pack_hi(unsigned lo_elem, unsigned hi_elem) { return (lo_elem >> 16) | (hi_elem & 0xffff0000u); }

__global__ void __launch_bounds__(64) dots(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ Df32, float* __restrict__ Dsplit) {
    const int l = threadIdx.x, i = l & 15, kq = l >> 4;
    f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c2 = c1;
    for (int k = 0; k < K; k += 4) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k + kq], B[(k + kq) * 16 + i], c1, 0, 0, 0);
    for (int k = 0; k < K; k += 32) {
        unsigned a0[8], a1[8], a2[8], b0[8], b1[8], b2[8];
        for (int e = 0; e < 8; ++e) {
            split3(A[i * K + k + 8 * kq + e], a0[e], a1[e], a2[e]);
            split3(B[(k + 8 * kq + e) * 16 + i], b0[e], b1[e], b2[e]);
        }
        auto pk = [](const unsigned (&p)[8]) {
            u32x4 r = {pack_hi(p[0], p[1]), pack_hi(p[2], p[3]), pack_hi(p[4], p[5]), pack_hi(p[6], p[7])};
            return __builtin_bit_cast(bf16x8, r);
        };
        const bf16x8 A0 = pk(a0), A1 = pk(a1), A2 = pk(a2), B0 = pk(b0), B1 = pk(b1), B2 = pk(b2);
        // smallest terms first
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B0, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B2, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B0, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B1, c2, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B0, c2, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) { Df32[(4 * kq + r) * 16 + i] = c1[r]; Dsplit[(4 * kq + r) * 16 + i] = c2[r]; }
}

int main() {
    unsigned long long* d;
    float* sink;
    (void)hipMalloc(&d, 64);
    (void)hipMalloc(&sink, 4096);
    const int iters = 2000;
    struct Case { const char* name; int threads, mA, mB; };
    std::vector<Case> cases = {
        {"1 wave/SIMD  bf16 MFMA only (8/iter)", 256, 0, 3}, {"2 waves/SIMD bf16 MFMA | bf16 MFMA", 512, 0, 0},
        {"2 waves/SIMD bf16 MFMA | VALU(64 fma)", 512, 0, 1}, {"2 waves/SIMD VALU | bf16 MFMA", 512, 1, 0},
        {"1 wave/SIMD  8 bf16 MFMA + 32 fma mixed", 256, 2, 3}, {"2 waves/SIMD mixed | mixed", 512, 2, 2},
    };
    for (auto& c : cases) {
        unsigned long long h[8] = {0};
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(d, 0, 64);
            hipLaunchKernelGGL(overlap, dim3(256), dim3(c.threads), 0, 0, c.mA, c.mB, iters, d, sink);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("%-42s wave0 %8.1f cyc/iter   wave4 %8.1f cyc/iter\n", c.name, (double)h[0] / iters, (double)h[4] / iters);
    }
    // accuracy
    std::mt19937 rng(7);
    for (int K : {32, 128, 512, 2304}) {
        for (int dist = 0; dist < 3; ++dist) {
            std::vector<float> A(16 * K), B(K * 16);
            std::uniform_real_distribution<float> U(-1.f, 1.f);
            std::normal_distribution<float> N(0.f, 1.f);
            for (auto& v : A) v = dist == 2 ? U(rng) * std::pow(10.f, U(rng) * 3.f) : U(rng) / std::sqrt((float)K);
            for (auto& v : B) v = dist == 0 ? U(rng) : dist == 1 ? N(rng) : U(rng) * std::pow(10.f, U(rng) * 3.f);
            float *dA, *dB, *d1, *d2;
            (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&d1, 1024); (void)hipMalloc(&d2, 1024);
            (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(dots, dim3(1), dim3(64), 0, 0, dA, dB, K, d1, d2);
            float h1[256], h2[256];
            (void)hipMemcpy(h1, d1, 1024, hipMemcpyDeviceToHost);
            (void)hipMemcpy(h2, d2, 1024, hipMemcpyDeviceToHost);
            double e1 = 0, e2 = 0, mag = 0, e12 = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double s = 0, sa = 0;
                    for (int k = 0; k < K; ++k) { s += (double)A[i * K + k] * B[k * 16 + j]; sa += std::fabs((double)A[i * K + k] * B[k * 16 + j]); }
                    e1 = std::fmax(e1, std::fabs(h1[i * 16 + j] - s) / sa);
                    e2 = std::fmax(e2, std::fabs(h2[i * 16 + j] - s) / sa);
                    e12 = std::fmax(e12, std::fabs((double)h2[i * 16 + j] - h1[i * 16 + j]) / sa);
                    mag = std::fmax(mag, std::fabs(s));
                }
            printf("K=%4d dist=%d  max|D|=%9.3e  err/sum|ab|: f32 MFMA %.3e   bf16x3 %.3e   (f32 vs split %.3e)\n", K, dist, mag, e1, e2, e12);
            (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(d1); (void)hipFree(d2);
        }
    }
    return 0;
}
