// Microbenchmark (GPU box), second form of pkmath_probe.hip: the victim loop is plain C++ on float2 values -- hipcc itself
// emits the v_pk_mul_f32 / v_pk_fma_f32 (with whatever hazard handling it knows) -- and the reference is computed with
// scalar v_mul_f32 / v_fma_f32 forced through inline assembly.  -DHAMMER selects what the co-resident workgroups issue:
//   1 bf16 MFMA bursts, 2 f32 MFMA bursts, 4 LDS reads, 8 plain VALU, 16 packed VALU (any sum; 0 = they run the victim loop too)
//
//   hipcc -O3 --offload-arch=gfx950 -DHAMMER=1 pkmath_probe2.hip -o pkmath_probe2 && ./pkmath_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#ifndef HAMMER
#define HAMMER 1
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ float s_mul(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float s_fma(float a, float b, float c) { float d; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

__global__ void __launch_bounds__(256, 2) probe(unsigned* bad, int iters, int with_hammer, float* sink) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    const bool hammer = with_hammer && blockIdx.x >= gridDim.x / 2;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    if (hammer) {
        bf16x8 ones;
        for (int i = 0; i < 8; ++i) ones[i] = (short)0x3f80;
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float t = 0.f;
        f32x2 p = {0.5f, 0.25f};
        for (int it = 0; it < iters * 4; ++it) {
#if HAMMER & 1
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ones, acc[i], 0, 0, 0);
#endif
#if HAMMER & 2
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, acc[i], 0, 0, 0);
#endif
#if HAMMER & 4
            t += lds[(lane * 4 + it) & 4095];
#endif
#if HAMMER & 8
#pragma unroll
            for (int i = 0; i < 16; ++i) t = fmaf(t, 0.999f, 0.001f);
#endif
#if HAMMER & 16
#pragma unroll
            for (int i = 0; i < 16; ++i) p = p * (f32x2){0.999f, 0.998f} + (f32x2){0.001f, 0.002f};
#endif
        }
        float s = t + p[0] + p[1];
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s < 0.f) sink[0] = s;
        return;
    }
    unsigned nbad = 0;
    f32x2 xy = {0.25f + 0.001f * lane, -0.5f + 0.002f * lane};
    for (int it = 0; it < iters; ++it) {
        f32x2 gam = {1.0f + 0.01f * (it & 15), 0.9f + 0.02f * (it & 7)}, bet = {0.01f * (it & 31), 0.078f}, r = {1.5f, 2.5f};
        asm volatile("" : "+v"(gam), "+v"(bet), "+v"(r));
        // victim: compiler-generated packed arithmetic (the LayerNorm tail: (d * rstd) * gamma + beta)
        const f32x2 t0 = xy * r;
        const f32x2 u0 = __builtin_elementwise_fma(t0, gam, bet);
        // reference: the same two lanes of arithmetic as scalar instructions
        const float s0 = s_fma(s_mul(xy[0], r[0]), gam[0], bet[0]), s1 = s_fma(s_mul(xy[1], r[1]), gam[1], bet[1]);
        if (u0[0] != s0 || u0[1] != s1) ++nbad;
        xy = xy * (f32x2){0.999f, 0.998f} + (f32x2){0.0007f, -0.0003f};
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    const int grid = 512;
    unsigned* bad; float* sink;
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 3; ++rep)
        for (int with_hammer = 0; with_hammer < 2; ++with_hammer) {
            CK(hipMemset(bad, 0, 4));
            probe<<<grid, 256, 72 * 1024, 0>>>(bad, 200000, with_hammer, sink);
            CK(hipGetLastError()); CK(hipDeviceSynchronize());
            unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            printf("HAMMER=%d %s: %u mismatching results of %.3g\n", HAMMER, with_hammer ? "victims beside hammer workgroups" : "victims only               ", h, 2.0 * 256 * 256 * 200000 * (with_hammer ? 1 : 2));
        }
    return 0;
}
