// gemm_conv4_mx4 (gemm_conv4.h) against gemm_pl_bf16<1, 256, false, 3> (gemm_planes.h) on synthetic mx4 operands: kernel times at a given row count and
// tile height, the output planes compared bit for bit, and -- with -DFS2_CONV4_TIMING -- where wave 0 of workgroup (0, 0) spends its cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DFS2_CONV4_TIMING] [-DFS2_CONV4_ABL=n] -I fastspeech2_amd/csrc -I tools/probes tools/probes/conv4_probe.hip -o tools/probes/conv4_probe.bin
//   conv4_probe.bin R(rows; c3: 35636)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gemm_mx.h"
#include "rejected/gemm_conv4.h"      // the one-wave-per-SIMD conv kernel (bit-identical, measured: not faster at steady state -- DESIGN.md section 3)
using namespace fs2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MT>
int run_conv4(GemmArgs a, float* ms_out) {
    constexpr size_t lds = conv4_lds_bytes<MT>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_conv4_mx4<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(a.N / 128, (a.R + 64 * MT - 1) / (64 * MT));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((gemm_conv4_mx4<MT>), grid, dim3(256), lds, 0, a);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    *ms_out = best;
    printf("gemm_conv4_mx4<%d>: %.1f us, %u workgroups of %d rows (%.2f rounds of 256), LDS %zu B", MT, best * 1e3, grid.x * grid.y, 64 * MT, grid.x * grid.y / 256.0, lds);
#ifdef FS2_CONV4_TIMING
    long long ph[8] = {0};
    CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_conv4_phase), sizeof ph));
    printf("; workgroup 0: prologue %lld | k-loop %lld (%.0f per step; barrier waits %lld = %.0f per step; MFMA issue %d per step) | epilogue %lld cycles",
           ph[2], ph[1], ph[1] / 81.0, ph[0], ph[0] / 81.0, 16 * 16 * MT, ph[3]);
#endif
    printf("\n");
    return 0;
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 35636;
    const int C = 384, N = 1024, k = 9, XU = 12, NU = 9;
    srand(1);
    // mx4 planes: per row 6 units of fp16 channels | 3 cross units (fp4 nibbles: any byte) | 3 units of e4m3 residual (not read); row scales 32 B per row
    std::vector<unsigned char> h((size_t)(R + 512) * XU * 128), hs((size_t)(R + 512) * 32);
    auto gauss = [] { float s = 0; for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX; return s - 6.f; };
    for (size_t r = 0; r < (size_t)R + 512; ++r) {
        unsigned char* row = &h[r * XU * 128];
        for (int c = 0; c < C; ++c) { const _Float16 v = (_Float16)gauss(); memcpy(row + 2 * c, &v, 2); }
        for (int b = 6 * 128; b < XU * 128; ++b) row[b] = (unsigned char)(rand() & 0x77);      // small e2m1 magnitudes
        for (int b = 0; b < 32; ++b) hs[r * 32 + b] = (unsigned char)(118 + rand() % 4);
    }
    std::vector<unsigned char> w((size_t)N * NU * k * 128), ws((size_t)(N / 128) * 3 * k * 1024);
    for (int n = 0; n < N; ++n)
        for (int u = 0; u < NU; ++u)
            for (int t = 0; t < k; ++t) {
                unsigned char* p = &w[(((size_t)n * NU + u) * k + t) * 128];
                if (u < 6) for (int c = 0; c < 64; ++c) { const _Float16 v = (_Float16)(gauss() * 0.02f); memcpy(p + 2 * c, &v, 2); }
                else for (int b = 0; b < 128; ++b) p[b] = (unsigned char)(rand() & 0x77);
            }
    for (auto& b : ws) b = (unsigned char)(118 + rand() % 4);
    std::vector<float> bias(N);
    for (auto& b : bias) b = gauss() * 0.1f;
    void *xp, *wb, *xs, *wsd, *bd, *y0, *y1;
    const size_t ybytes = (size_t)(R + 512) * (N / 32) * 128;      // mx planes of the hidden layer: 4 bytes per channel
    CK(hipMalloc(&xp, h.size())); CK(hipMalloc(&wb, w.size())); CK(hipMalloc(&xs, hs.size())); CK(hipMalloc(&wsd, ws.size())); CK(hipMalloc(&bd, N * 4));
    CK(hipMalloc(&y0, ybytes)); CK(hipMalloc(&y1, ybytes));
    CK(hipMemcpy(xp, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(wb, w.data(), w.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(xs, hs.data(), hs.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(wsd, ws.data(), ws.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(bd, bias.data(), N * 4, hipMemcpyHostToDevice));
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.C = C; a.Cpad = C; a.ktaps = k; a.N = N; a.R = R; a.W = (const float*)wb; a.Wb = wb; a.Xp = xp; a.x_scale = 1.f; a.bias = (const float*)bd;
    a.x_rowscale = (const unsigned char*)xs; a.w_rowscale = (const unsigned char*)wsd; a.mx = 2;
    a.act_post = 1; a.yp_chunks = N / 32; a.yp_f16 = 2; a.yp_scale = 4.f;
    // reference: the two-workgroups-per-CU kernel
    {
        const size_t lds = pl_lds_bytes<256, false>() + kMx4RowScaleLds + 2048;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pl_bf16<1, 256, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipMemset(y0, 0, ybytes));
        a.Yp = y0;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((gemm_pl_bf16<1, 256, false, 3>), dim3(N / 128, (R + 255) / 256), dim3(256), lds, 0, a);
            hipEventRecord(e1); CK(hipDeviceSynchronize());
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        printf("gemm_pl_bf16<1,256,false,3>: %.1f us, %d workgroups of 256 rows, two per CU\n", best * 1e3, (N / 128) * ((R + 255) / 256));
    }
    std::vector<unsigned> r0(ybytes / 4), r1(ybytes / 4);
    CK(hipMemcpy(r0.data(), y0, ybytes, hipMemcpyDeviceToHost));
    for (int mt = 4; mt <= 6; ++mt) {
        CK(hipMemset(y1, 0, ybytes));
        a.Yp = y1;
        float ms;
        if (mt == 4 ? run_conv4<4>(a, &ms) : (mt == 5 ? run_conv4<5>(a, &ms) : run_conv4<6>(a, &ms))) return 1;
        CK(hipMemcpy(r1.data(), y1, ybytes, hipMemcpyDeviceToHost));
        size_t diff = 0, nz = 0;
        for (size_t i = 0; i < r0.size(); ++i) { diff += r0[i] != r1[i]; nz += r0[i] != 0; }
        printf("   output planes: %zu of %zu words differ from gemm_pl_bf16's (%zu non-zero)\n", diff, r0.size(), nz);
    }
    return 0;
}
