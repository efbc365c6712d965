// Microbenchmark (GPU box): is a wave's private (scratch) memory really private when several workgroups share a CU?
//
// Every lane keeps 32 words in a private array that the compiler must place in scratch (volatile, runtime index),
// fills them with values unique to (workgroup, lane, word), idles, and checks them again -- `rounds` times.  A word
// that reads back as something else was written by another wave: the mismatch counter then also records whose value it
// was (same CU slot? another workgroup?).  LDS per workgroup selects how many workgroups fit a CU.
//
//   hipcc -O3 --offload-arch=gfx950 scratch_probe.hip -o scratch_probe && ./scratch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

struct Report { unsigned bad; unsigned first_want; unsigned first_got; unsigned pad; };

template <int WORDS>
__global__ void __launch_bounds__(256) probe(Report* rep, int rounds, int sleep, int lds_floats, float* sink) {
    extern __shared__ float lds[];
    volatile unsigned priv[WORDS];
    const unsigned gid = blockIdx.x * 256u + threadIdx.x;
    for (int i = threadIdx.x; i < lds_floats; i += 256) lds[i] = (float)i;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < WORDS; ++i) priv[(i + r) % WORDS] = gid * 64u + (unsigned)((i + r) % WORDS);
        for (int s = 0; s < sleep; ++s) __builtin_amdgcn_s_sleep(64);
        __syncthreads();
        for (int i = 0; i < WORDS; ++i) {
            const int k = (i * 7 + r) % WORDS;
            const unsigned want = gid * 64u + (unsigned)k, got = priv[k];
            if (got != want) {
                if (atomicAdd(&rep->bad, 1u) == 0u) { rep->first_want = want; rep->first_got = got; }
            }
        }
    }
    if (lds_floats > 0 && lds[(gid * 13u) % (unsigned)lds_floats] < -1.f) sink[0] = 1.f;
}

int main() {
    Report* rep; float* sink;
    CK(hipMalloc(&rep, sizeof(Report)));
    CK(hipMalloc(&sink, 4));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)probe<32>));
    printf("probe<32>: localSizeBytes (scratch per lane) %zu, numRegs %d\n", fa.localSizeBytes, fa.numRegs);
    CK(hipFuncSetAttribute((const void*)probe<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int lds_cfg[] = {0, 36 * 1024, 72 * 1024, 120 * 1024};      // bytes: 8 (wave-limited), 4, 2, 1 workgroups per CU
    const int grids[] = {256, 512, 1024, 2048, 8192};
    for (int lb : lds_cfg)
        for (int g : grids) {
            Report z{0, 0, 0, 0};
            CK(hipMemcpy(rep, &z, sizeof(z), hipMemcpyHostToDevice));
            probe<32><<<g, 256, lb, 0>>>(rep, 8, 20, lb / 4, sink);
            CK(hipGetLastError());
            CK(hipDeviceSynchronize());
            Report r; CK(hipMemcpy(&r, rep, sizeof(r), hipMemcpyDeviceToHost));
            printf("lds %6d B/workgroup, grid %5d workgroups: %u mismatching words of %lld", lb, g, r.bad, (long long)g * 256 * 32 * 8);
            if (r.bad) printf("   first: wanted lane %u word %u, got lane %u word %u", r.first_want / 64, r.first_want % 64, r.first_got / 64, r.first_got % 64);
            printf("\n");
        }
    return 0;
}
