// Microbenchmark (GPU box): does packed-f32 VALU arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, the instructions
// hipcc emits for the LayerNorm of the fused kernels) stay exact on a wave that shares its SIMD with a wave issuing MFMA
// bursts?  Workgroups of the first half of the grid ("probe") evaluate packed chains whose results are also computed with
// scalar f32 instructions in the same lane; workgroups of the second half ("hammer") issue bf16 and f32 MFMA bursts, LDS
// traffic and DPP moves.  72 KB of LDS per workgroup = two workgroups per CU; HW_ID of every workgroup is recorded so
// the host can say how many probe waves really shared a SIMD with a hammer wave.
//
//   hipcc -O3 --offload-arch=gfx950 pkmath_probe.hip -o pkmath_probe && ./pkmath_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_fma_bcast(f32x2 a, f32x2 b, f32x2 c) {     // lo/hi both use b.hi and c.hi (op_sel:[0,1,1])
    f32x2 d;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1]" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
    f32x2 d;
    asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ unsigned g_rec[8 + 64 * 6];      // [0..3] mismatch counts per checked result; records: lane, which, got, want, it, wave

__global__ void __launch_bounds__(256, 2) probe(unsigned* bad, unsigned* hwid, int iters, int with_hammer, float* sink) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool hammer = with_hammer && blockIdx.x >= gridDim.x / 2;
    if (lane == 0) hwid[blockIdx.x * 4 + wave] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) | (hammer ? 0x80000000u : 0u) |
                                                  ((unsigned)__builtin_amdgcn_s_getreg((7 << 11) | (0 << 6) | 20) << 20);   // HW_ID | XCC_ID
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
    __syncthreads();
    if (hammer) {
        bf16x8 ones;
        for (int i = 0; i < 8; ++i) ones[i] = (short)0x3f80;
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float t = 0.f;
        for (int it = 0; it < iters * 4; ++it) {
            if (with_hammer & 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, ones, acc[i], 0, 0, 0);
            }
            if (with_hammer & 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, 1.0f, acc[i], 0, 0, 0);
            }
            if (with_hammer & 4) t += lds[(lane * 4 + it) & 4095];
            if (with_hammer & 8) {                      // plain (unpacked) VALU stream
#pragma unroll
                for (int i = 0; i < 16; ++i) t = fmaf(t, 0.999f, 0.001f);
            }
        }
        float s = t;
        for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s < 0.f) sink[0] = s;
        return;
    }
    unsigned nbad = 0;
    float x = 0.25f + 0.001f * lane, y = -0.5f + 0.002f * lane;
    for (int it = 0; it < iters; ++it) {
        const float g0 = 1.0f + 0.01f * (it & 15), g1 = 0.9f + 0.02f * (it & 7), b0 = 0.01f * (it & 31), b1 = 0.078f;
        f32x2 a = {x, y}, gam = {g0, g1}, bet = {b0, b1}, r = {1.5f, 2.5f};
        asm volatile("" : "+v"(a), "+v"(gam), "+v"(bet), "+v"(r));
        // the LayerNorm tail of the fused kernels: (d * rstd) * gamma + beta with per-pair and broadcast operand selects
        f32x2 t0 = pk_mul(a, r);
        f32x2 u0 = pk_fma(t0, gam, bet);
        f32x2 u1 = pk_fma_bcast(t0, gam, bet);
        const float s0 = fmaf(x * 1.5f, g0, b0), s1 = fmaf(y * 2.5f, g1, b1), s2 = fmaf(x * 1.5f, g1, b1);
        const float got[4] = {u0[0], u0[1], u1[0], u1[1]}, want[4] = {s0, s1, s2, s1};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (got[k] != want[k]) {
                ++nbad;
                atomicAdd(&g_rec[k], 1u);
                const unsigned n = atomicAdd(&g_rec[4], 1u);
                if (n < 64) { unsigned* r = g_rec + 8 + 6 * n; r[0] = lane; r[1] = k; r[2] = __float_as_uint(got[k]); r[3] = __float_as_uint(want[k]); r[4] = it; r[5] = wave | (blockIdx.x << 8); }
            }
        x = x * 0.999f + 0.0007f; y = y * 0.998f - 0.0003f;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main() {
    const int grid = 512;
    unsigned *bad, *hwid; float* sink;
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&hwid, grid * 4 * 4));
    CK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int modes[] = {0, 1, 2, 4, 8, 3, 5, 6, 7, 15};
    const char* mname[] = {"probe only", "bf16 MFMA hammer", "f32 MFMA hammer", "LDS-read hammer", "plain VALU hammer", "bf16 + f32 MFMA hammer",
                           "bf16 MFMA + LDS", "f32 MFMA + LDS", "bf16 + f32 MFMA + LDS", "all four"};
    for (int mi = 0; mi < 10; ++mi) {
        const int with_hammer = modes[mi];
        unsigned zero[8 + 64 * 6] = {0};
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_rec), zero, sizeof(zero)));
        CK(hipMemset(bad, 0, 4));
        probe<<<grid, 256, 72 * 1024, 0>>>(bad, hwid, 200000, with_hammer, sink);
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        unsigned h; CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
        std::vector<unsigned> id(grid * 4);
        CK(hipMemcpy(id.data(), hwid, grid * 16, hipMemcpyDeviceToHost));
        std::map<unsigned, int> roles;      // (xcc, se, cu, simd) -> bit 0: a probe wave ran there, bit 1: a hammer wave
        for (unsigned v : id) {
            const unsigned key = ((v >> 20) & 0xf) << 16 | ((v >> 13) & 7) << 12 | ((v >> 8) & 15) << 4 | ((v >> 4) & 3);
            roles[key] |= (v & 0x80000000u) ? 2 : 1;
        }
        int shared = 0;
        for (auto& kv : roles) shared += kv.second == 3;
        unsigned rec[8 + 64 * 6];
        CK(hipMemcpyFromSymbol(rec, HIP_SYMBOL(g_rec), sizeof(rec)));
        printf("%-24s: %u mismatching packed results [pk_fma.lo %u, pk_fma.hi %u, pk_fma op_sel.lo %u, op_sel.hi %u]; %zu distinct SIMD ids, %d shared\n",
               mname[mi], h, rec[0], rec[1], rec[2], rec[3], roles.size(), shared);
        for (unsigned i = 0; i < rec[4] && i < 12; ++i) {
            const unsigned* r = rec + 8 + 6 * i;
            float g, w; memcpy(&g, &r[2], 4); memcpy(&w, &r[3], 4);
            printf("      lane %2u result %u got %+.8f want %+.8f (bits %08x %08x) iteration %u wave %u workgroup %u\n", r[0], r[1], g, w, r[2], r[3], r[4], r[5] & 255, r[5] >> 8);
        }
    }
    return 0;
}
