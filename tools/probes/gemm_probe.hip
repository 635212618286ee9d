// Phase timing of gemm_pl_bf16<3, BM, K1> on synthetic operands (planes in, fp32 out): ticks (s_memtime) wave 0 of workgroup 0
// spends per k-step in: barrier wait | DMA issue | fragment reads + MFMAs | chunk-end A refill (conv form), plus the kernel time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFS2_GEMM_TIMING -I fastspeech2_amd/csrc -I tools/probes tools/probes/gemm_probe.hip -o tools/probes/gemm_probe.bin
//   gemm_probe.bin R C N ktaps BM     (BM in 64,128,256; ktaps 1 -> k = 1 form)
// Conv form: also runs gemm_plr_bf16 (weights straight to registers) on the same operands and compares the outputs bit for bit.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_plr.h"
using namespace fs2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int BM, bool K1>
int run(GemmArgs a, int steps) {
    constexpr size_t lds = pl_lds_bytes<BM, K1>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pl_bf16<3, BM, K1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((a.N + 127) / 128, (a.R + BM - 1) / BM);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
#ifdef FS2_GEMM_TIMING
        long long zero[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_phase), zero, sizeof zero));
#endif
        hipEventRecord(e0);
        hipLaunchKernelGGL((gemm_pl_bf16<3, BM, K1>), grid, dim3(256), lds, 0, a);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long ph[8] = {0};
#ifdef FS2_GEMM_TIMING
        CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_gemm_phase), sizeof ph));
#endif
        printf("BM=%d k=%d R=%d C=%d N=%d: %.1f us, %u workgroups x %d steps; per step: barrier %lld | dma issue %lld | reads+mfma %lld | A refill (per step) %lld ticks\n",
               BM, a.ktaps, a.R, a.C, a.N, ms * 1e3, grid.x * grid.y, steps, ph[0] / steps, ph[1] / steps, ph[2] / steps, ph[3] / steps);
        if (ph[7]) printf("   k-loop of workgroup 0: %lld shader cycles in %.2f us (100-MHz counter) -> %.0f MHz effective clock\n", ph[6], ph[7] * 0.01, ph[6] / (ph[7] * 0.01));
    }
    return 0;
}
template <int BM>
int run_plr(GemmArgs a, size_t wbytes) {
    constexpr size_t lds = plr_lds_bytes<BM>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_plr_bf16<3, BM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* wf; CK(hipMalloc(&wf, wbytes));
    const int niter = a.Cpad / 32 * a.ktaps, Npad = (a.N + 127) / 128 * 128;
    const long long nvec = (long long)Npad * niter * 8;
    hipLaunchKernelGGL(frag_weight_image, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, 0, reinterpret_cast<const __bf16*>(a.Wb), Npad, niter, reinterpret_cast<uint4*>(wf));
    float* y2; CK(hipMalloc(&y2, (size_t)a.R * a.N * 4));
    GemmArgs b = a; b.xp_scratch = wf; b.Y = y2;
    dim3 grid((a.N + 127) / 128, (a.R + BM - 1) / BM);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((gemm_plr_bf16<3, BM>), grid, dim3(256), lds, 0, b);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("plr BM=%d: %.1f us\n", BM, ms * 1e3);
    }
    std::vector<float> h1((size_t)a.R * a.N), h2(h1.size());
    CK(hipMemcpy(h1.data(), a.Y, h1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), y2, h2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < h1.size(); ++i) bad += memcmp(&h1[i], &h2[i], 4) != 0;
    printf("plr vs pl: %zu of %zu outputs differ\n", bad, h1.size());
    return 0;
}
int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 30208, C = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 256;
    const int k = argc > 4 ? atoi(argv[4]) : 5, BM = argc > 5 ? atoi(argv[5]) : 128;
    const int nchunks = (C + 31) / 32, Npad = (N + 127) / 128 * 128;
    std::vector<unsigned short> h((size_t)(R + 64) * nchunks * 64);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) ? 0x8000 : 0);
    std::vector<unsigned short> w((size_t)Npad * nchunks * k * 64);
    for (auto& v : w) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) ? 0x8000 : 0);
    void *xp, *wb; float* y;
    CK(hipMalloc(&xp, h.size() * 2)); CK(hipMalloc(&wb, w.size() * 2)); CK(hipMalloc(&y, (size_t)R * N * 4));
    CK(hipMemcpy(xp, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(wb, w.data(), w.size() * 2, hipMemcpyHostToDevice));
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.C = C; a.Cpad = nchunks * 32; a.ktaps = k; a.N = N; a.R = R; a.W = (const float*)wb; a.Wb = wb; a.Xp = xp; a.Y = y; a.ldy = N; a.x_scale = 1.f;
    const int steps = nchunks * k;
    if (k == 1) return BM == 128 ? run<128, true>(a, steps) : run<64, true>(a, steps);
    const size_t wbytes = w.size() * 2;
    if (BM == 256) { if (run<256, false>(a, steps)) return 1; return run_plr<256>(a, wbytes); }
    if (BM == 128) { if (run<128, false>(a, steps)) return 1; return run_plr<128>(a, wbytes); }
    if (run<64, false>(a, steps)) return 1;
    return run_plr<64>(a, wbytes);
}
