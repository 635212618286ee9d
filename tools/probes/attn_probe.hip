// Phase timing of attn_bf16<192,3> on synthetic operands: B utterances of L frames, 2 heads.  Prints the cycles wave 0 of
// workgroup 0 spent per phase of the tile loop (s_memtime) and the whole-kernel time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFS2_ATT_TIMING -I fastspeech2_amd/csrc tools/probes/attn_probe.hip -o tools/probes/attn_probe.bin
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_bf16.h"
#include "attn_bf16.h"
using namespace fs2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, L = argc > 2 ? atoi(argv[2]) : 448, D = 384, DK = 192, heads = 2;
    std::vector<int> start(B), len(B, L), klen(B, L);
    int row = 8;
    for (int b = 0; b < B; ++b) { row = (row + 31) & ~31; start[b] = row; row += L + 8; }
    const int Rvt = (row + 127) & ~127;
    std::vector<int2> work;
    for (int b = 0; b < B; ++b) for (int q = 0; q * 64 < L; ++q) work.push_back(make_int2(b, q));
    std::vector<unsigned short> h((size_t)Rvt * 2 * D);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) - ((rand() & 1) ? 0x8000 : 0);     // bf16 around +-0.01..0.03
    __bf16 *qkh, *qkl, *vth, *vtl; float* ctx; int *dstart, *dlen, *dklen; int2* dwork;
    CK(hipMalloc(&qkh, h.size() * 2)); CK(hipMalloc(&qkl, h.size() * 2));
    CK(hipMalloc(&vth, (size_t)D * Rvt * 2)); CK(hipMalloc(&vtl, (size_t)D * Rvt * 2)); CK(hipMalloc(&ctx, (size_t)Rvt * D * 4));
    CK(hipMemcpy(qkh, h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(qkl, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(vth, h.data(), (size_t)D * Rvt * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(vtl, h.data(), (size_t)D * Rvt * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&dstart, B * 4)); CK(hipMalloc(&dlen, B * 4)); CK(hipMalloc(&dklen, B * 4)); CK(hipMalloc(&dwork, work.size() * 8));
    CK(hipMemcpy(dstart, start.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dlen, len.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dklen, klen.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwork, work.data(), work.size() * 8, hipMemcpyHostToDevice));
    AttnB16Args a;
    memset(&a, 0, sizeof a);
    a.qk_hi = qkh; a.qk_lo = qkl; a.ldqk = 2 * D; a.vt_hi = vth; a.vt_lo = vtl; a.Rvt = Rvt; a.ctx = ctx; a.ldc = D; a.ctxp = nullptr; a.ctxp_chunks = D / 32;
    a.start = dstart; a.len = dlen; a.klen = dklen; a.work = dwork; a.nwork = nullptr; a.D = D; a.mask_q = 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bf16<192, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_b16_lds_bytes<192>()));
    dim3 grid((unsigned)work.size(), heads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        long long zero[8] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_att_phase), zero, sizeof zero));
        hipEventRecord(e0);
        hipLaunchKernelGGL((attn_bf16<192, 3>), grid, dim3(256), attn_b16_lds_bytes<192>(), 0, a);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long ph[8]; CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_att_phase), sizeof ph));
        const int tiles = (L + 31) / 32;
        printf("B=%d L=%d: %.1f us, %zu workgroups x %d tiles; wave 0 of workgroup 0, cycles per tile: barrierA %lld | QK^T %lld | softmax %lld | Vstore+barrierB %lld | PV %lld | Kstore %lld\n",
               B, L, ms * 1e3, work.size() * heads, tiles, ph[0] / tiles, ph[1] / tiles, ph[2] / tiles, ph[3] / tiles, ph[4] / tiles, ph[5] / tiles);
    }
    return 0;
}
