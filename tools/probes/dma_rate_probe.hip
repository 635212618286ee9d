// How fast can one workgroup pull bytes into LDS with global_load_lds (1-KB pieces), as a function of the number of waves that
// issue, the stage size and the number of stages kept in flight?  The k-loops of the small-grid GEMMs are this loop plus MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/dma_rate_probe.hip -o tools/probes/dma_rate_probe.bin
//   dma_rate_probe.bin [workgroups = 256]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void dma16(const void* src, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_off) : "memory");
}
__device__ __forceinline__ void wait_le(int n) {
#define W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) { W(1) W(2) W(3) W(4) W(6) W(8) W(9) W(12) W(16) W(18) W(24) W(32) W(48) default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#undef W
}

// each workgroup streams `nstages` stages of `pieces` KB; depth stages of LDS; shared != 0: all workgroups read the same bytes
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stream(const char* __restrict__ src, size_t wg_stride, int nstages, int pieces, int depth, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)lds);
    const char* base = src + (size_t)blockIdx.x * wg_stride + lane * 16;
    const int pw = pieces / WAVES;                       // pieces per wave and stage
    const int stage_bytes = pieces * 1024;
    auto issue = [&](int it) {
        const unsigned dst = lds0 + (it % depth) * stage_bytes + wave * 1024;
        const char* s = base + (size_t)it * stage_bytes + wave * 1024;
        for (int q = 0; q < pw; ++q) dma16(s + q * WAVES * 1024, dst + q * WAVES * 1024);
    };
    for (int j = 0; j < depth - 1 && j < nstages; ++j) issue(j);
    unsigned acc = 0;
    for (int it = 0; it < nstages; ++it) {
        const int left = nstages - 1 - it;
        wait_le((left < depth - 2 ? left : depth - 2) * pw);
        __syncthreads();
        if (it + depth - 1 < nstages) issue(it + depth - 1);
        acc += *reinterpret_cast<const unsigned*>(lds + (it % depth) * stage_bytes + threadIdx.x * 4);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int WAVES>
int run(const char* src, size_t wg_stride, int wgs, int nstages, int pieces, int depth, unsigned* sink, const char* what) {
    const size_t lds = (size_t)depth * pieces * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream<WAVES>), dim3(wgs), dim3(WAVES * 64), lds, 0, src, wg_stride, nstages, pieces, depth, sink);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double us = best * 1e3, per_wg = (double)nstages * pieces * 1024;
    printf("%-7s waves %d  stage %2d KB  depth %d  %3d stages: %7.1f us  %5.2f us/stage  %6.1f GB/s per workgroup  %6.2f TB/s all %d\n",
           what, WAVES, pieces, depth, nstages, us, us / nstages, per_wg / us * 1e-3, per_wg * wgs / us * 1e-6, wgs);
    return 0;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 256;
    const int nstages = 48;
    const size_t per_wg = (size_t)nstages * 64 * 1024;
    char* src; unsigned* sink;
    CK(hipMalloc(&src, per_wg * wgs)); CK(hipMemset(src, 1, per_wg * wgs)); CK(hipMalloc(&sink, 64));
    for (int shared = 1; shared >= 0; --shared) {
        const size_t stride = shared ? 0 : per_wg;
        const char* what = shared ? "shared" : "unique";
        for (int pieces : {16, 24, 48})
            for (int depth : {2, 3, 6}) {
                if ((size_t)depth * pieces * 1024 > 160 * 1024) continue;
                if (pieces % 4 == 0 && run<4>(src, stride, wgs, nstages, pieces, depth, sink, what)) return 1;
                if (pieces % 8 == 0 && run<8>(src, stride, wgs, nstages, pieces, depth, sink, what)) return 1;
            }
    }
    return 0;
}
