// Microbenchmark (GPU box): with TWO workgroups per CU (two waves per SIMD) competing for the matrix pipe, are the wait
// states hipcc inserts between an MFMA and (a) a VALU read of its result, (b) a VALU overwrite of its result registers /
// of its A/B source registers still sufficient?  Even workgroups ("hammer") issue long MFMA bursts; odd workgroups
// ("probe") run short MFMA chains whose results are known in closed form, read them as early as the compiler allows,
// reuse the registers and check everything.  Any mismatch = a result that landed late / a source that was read late.
//
//   hipcc -O3 --offload-arch=gfx950 mfma_hazard_probe.hip -o mfma_hazard_probe && ./mfma_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void __launch_bounds__(256, 2) probe(unsigned* bad, int iters, int mode, float* sink) {
    extern __shared__ float lds[];          // 72 KB per workgroup: exactly two workgroups per CU
    const int lane = threadIdx.x & 63;
    if (threadIdx.x == 0) lds[0] = 0.f;
    const bool hammer = (mode == 0) ? false : (mode == 1 ? (blockIdx.x & 1) == 0 : false);
    // bf16 1.0 = 0x3f80: A = B = all ones => every element of D += 32 per MFMA (K = 32)
    bf16x8 ones;
    for (int i = 0; i < 8; ++i) ones[i] = (short)0x3f80;
    if (hammer) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters * 8; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ones, acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s < 0.f) sink[0] = s;
        return;
    }
    unsigned nbad = 0;
    for (int it = 0; it < iters; ++it) {
        // chain of 6 dependent MFMAs (like one bf16x3 product), 4 interleaved accumulators
        f32x4 acc[4];
        bf16x8 a = ones, b = ones;
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){(float)i, (float)i, (float)i, (float)i};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        // (a) read the results as early as the compiler allows
        float got[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) got[i][r] = acc[i][r];
        // (b) reuse the accumulator registers and the source registers right away
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc[i] = (f32x4){-7.f, -7.f, -7.f, -7.f}; asm volatile("" : "+v"(acc[i])); }
        for (int i = 0; i < 8; ++i) { a[i] = (short)0x4000; b[i] = (short)0x4000; }      // bf16 2.0: a late source read would give 128 per MFMA
        asm volatile("" : "+v"(a), "+v"(b));
        for (int s = 0; s < 4; ++s) __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (got[i][r] != 192.f + (float)i) ++nbad;              // 6 x 32 + i
                if (acc[i][r] != -7.f) ++nbad;                           // a result that landed after the reuse
            }
        if (a[0] != (short)0x4000) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
    if (lane == 0 && lds[0] < -1.f) sink[0] = 1.f;
}

int main() {
    unsigned* bad; float* sink;
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 4));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const char* names[] = {"probe only, two workgroups per CU", "hammer + probe sharing every SIMD", "probe only, one workgroup per CU"};
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(bad, 0, 4));
        const int lds = mode == 2 ? 120 * 1024 : 72 * 1024;
        probe<<<mode == 2 ? 256 : 512, 256, lds, 0>>>(bad, 20000, mode, sink);
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        printf("%-40s mismatches %u\n", names[mode], h);
    }
    return 0;
}
