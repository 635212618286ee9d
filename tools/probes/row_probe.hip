// gemm_row4_bf16 (one wave per SIMD, accumulators in literal AGPRs) against gemm_row8_bf16 on synthetic operands: bit-for-bit comparison of
// the fp32 rows and the planes, and launch times at a given shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fastspeech2_amd/csrc tools/probes/row_probe.hip -o tools/probes/row_probe.bin
//   (-DFS2_ROW_TIMING: phase stamps of gemm_row4_bf16, printed per sampled workgroup)
//   row_probe.bin R C [reps]      N = 384; C = 384 (out-proj + LN, decoder input layer) or 1024 (FFN2 + LN)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "gemm_row4.h"
#include "gemm_mx.h"
#include "elementwise.h"
using namespace fs2;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static unsigned short bf16_of(float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float f_of(unsigned short h) { unsigned u = (unsigned)h << 16; float v; memcpy(&v, &u, 4); return v; }

struct Bufs { float* y; void* yp; };

template <class K>
static float time_kernel(K launch, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch(i);
    hipEventRecord(e1); CK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

template <int MT>
static void launch_row8(const GemmArgs& a) {
    constexpr size_t lds = row8_lds_bytes<3, MT>();
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_row8_bf16<3, 3, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL((gemm_row8_bf16<3, 3, MT>), dim3((a.R + 64 * MT - 1) / (64 * MT)), dim3(512), lds, 0, a);
}
template <int MT>
static void launch_row4_mx(const GemmArgs& a) {
    constexpr size_t lds = row4_lds_bytes<3, MT>();
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_row4_bf16<3, 3, MT, 0, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL((gemm_row4_bf16<3, 3, MT, 0, 2, 2>), dim3((a.R + 32 * MT - 1) / (32 * MT)), dim3(256), lds, 0, a);
}
template <int MT, int EPI, int SCHED>
static void launch_row4(const GemmArgs& a) {
    constexpr size_t lds = row4_lds_bytes<3, MT>();
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_row4_bf16<3, 3, MT, EPI, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL((gemm_row4_bf16<3, 3, MT, EPI, SCHED>), dim3((a.R + 32 * MT - 1) / (32 * MT)), dim3(256), lds, 0, a);
}

template <int MT, int EPI, int ARITH, int RES>
static void launch_row4_po(const GemmArgs& a) {      // the planes-only forms the library runs since round 6 (gemm_row4.h: RES)
    constexpr size_t lds = row4_lds_bytes<3, MT>();
    static bool done = false;
    if (!done) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_row4_bf16<3, 3, MT, EPI, 2, ARITH, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
    hipLaunchKernelGGL((gemm_row4_bf16<3, 3, MT, EPI, 2, ARITH, RES>), dim3((a.R + 32 * MT - 1) / (32 * MT)), dim3(256), lds, 0, a);
}
// mx planes -> the fp32 value their reader reconstructs: fp16(x) + e4m3 residual x scale (gemm_row4.h: resid4)
__global__ void mx_planes_to_rows(const void* planes, int C, int R, float scale, float* dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * C) return;
    const int row = (int)(i / C), c = (int)(i % C);
    const char* p = reinterpret_cast<const char*>(planes) + (size_t)row * 4 * C;
    const _Float16 hf = *reinterpret_cast<const _Float16*>(p + 2 * c);
    const int w = *reinterpret_cast<const int*>(p + 2 * C + (c & ~3));
    float e;
    switch (c & 3) { case 0: e = __builtin_amdgcn_cvt_f32_fp8(w, 0); break; case 1: e = __builtin_amdgcn_cvt_f32_fp8(w, 1); break;
                     case 2: e = __builtin_amdgcn_cvt_f32_fp8(w, 2); break; default: e = __builtin_amdgcn_cvt_f32_fp8(w, 3); }
    dst[i] = (float)hf + e * scale;
}

static size_t compare(const char* what, const void* d_ref, const void* d_got, size_t bytes, bool bf16_pairs = false) {
    std::vector<unsigned> r(bytes / 4), g(bytes / 4);
    CK(hipMemcpy(r.data(), d_ref, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(g.data(), d_got, bytes, hipMemcpyDeviceToHost));
    size_t bad = 0, first = (size_t)-1;
    // (-0 == +0, per 16-bit half: rows that are padding leave gemm_qkv8_bf16 as v x 0 = +-0 with the sign of a value nobody defined)
    auto canon = [](unsigned w) { if (!(w & 0x7fff0000u)) w &= 0x0000ffffu; if (!(w & 0x00007fffu)) w &= 0xffff0000u; return w; };
    for (size_t i = 0; i < r.size(); ++i) if (r[i] != g[i] && !(bf16_pairs && canon(r[i]) == canon(g[i]))) { if (!bad) first = i; ++bad; }
    printf("    %-8s %zu of %zu words differ%s", what, bad, r.size(), bad ? "" : "\n");
    if (bad) printf(" (first at word %zu: %08x vs %08x)\n", first, r[first], g[first]);
    return bad;
}

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 36611, C = argc > 2 ? atoi(argv[2]) : 384, reps = argc > 3 ? atoi(argv[3]) : 20;
    const int N = 384, nchunks = C / 32;
    const int Rpad = (R + 191) / 192 * 192 + 192;
    srand(12345);
    // A planes [Rpad][nchunks][hi 32 | lo 32] and the weight image [N][nchunks][hi 32 | lo 32]: any finite bf16 pairs (hi + small lo)
    auto fill_planes = [&](std::vector<unsigned short>& h, size_t rows, float amp) {
        h.resize(rows * nchunks * 64);
        for (size_t i = 0; i < rows * nchunks; ++i)
            for (int k = 0; k < 32; ++k) {
                const float v = amp * ((rand() & 0xffff) - 32768) / 32768.f;
                const unsigned short hi = bf16_of(v);
                h[i * 64 + k] = hi;
                h[i * 64 + 32 + k] = bf16_of(v - f_of(hi));
            }
    };
    std::vector<unsigned short> hx, hw;
    fill_planes(hx, Rpad, 1.f);
    fill_planes(hw, N, 0.06f);
    std::vector<float> hres((size_t)Rpad * N), hb(N), hg(N), hbe(N), hpe((size_t)1024 * N);
    for (auto& v : hres) v = ((rand() & 0xffff) - 32768) / 32768.f;
    for (int i = 0; i < N; ++i) { hb[i] = 0.01f * (i % 7); hg[i] = 0.8f + 0.001f * i; hbe[i] = 0.02f * (i % 5) - 0.03f; }
    for (auto& v : hpe) v = ((rand() & 0xffff) - 32768) / 32768.f;
    std::vector<int> hpos(Rpad);
    for (int r = 0; r < Rpad; ++r) hpos[r] = (r % 509 < 8 || r >= R) ? -1 : (r % 509) - 8;      // an "utterance" of 501 rows behind every 8 gap rows
    void *xp[2], *wb; float *res[2], *bias, *g, *be, *pe, *alpha; int* pos;
    for (int s = 0; s < 2; ++s) {
        CK(hipMalloc(&xp[s], hx.size() * 2)); CK(hipMemcpy(xp[s], hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&res[s], hres.size() * 4)); CK(hipMemcpy(res[s], hres.data(), hres.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&wb, hw.size() * 2)); CK(hipMemcpy(wb, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&bias, N * 4)); CK(hipMemcpy(bias, hb.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&g, N * 4)); CK(hipMemcpy(g, hg.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&be, N * 4)); CK(hipMemcpy(be, hbe.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&pe, hpe.size() * 4)); CK(hipMemcpy(pe, hpe.data(), hpe.size() * 4, hipMemcpyHostToDevice));
    const float al = 0.9f;
    CK(hipMalloc(&alpha, 4)); CK(hipMemcpy(alpha, &al, 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&pos, Rpad * 4)); CK(hipMemcpy(pos, hpos.data(), Rpad * 4, hipMemcpyHostToDevice));
    const size_t ybytes = (size_t)Rpad * N * 4;
    Bufs ref, out[2];
    CK(hipMalloc(&ref.y, ybytes)); CK(hipMalloc(&ref.yp, ybytes));
    for (int s = 0; s < 2; ++s) { CK(hipMalloc(&out[s].y, ybytes)); CK(hipMalloc(&out[s].yp, ybytes)); }

    auto args = [&](int epi, int set, const Bufs& o) {
        GemmArgs a;
        memset(&a, 0, sizeof a);
        a.C = C; a.Cpad = C; a.ktaps = 1; a.N = N; a.R = R; a.W = (const float*)wb; a.Wb = wb; a.Xp = xp[set]; a.row_pos = pos;
        a.bias = bias; a.resid = epi == 2 ? nullptr : res[set]; a.ldr = N; a.ln_g = g; a.ln_b = be; a.ln_eps = 1e-5f;
        a.Y = o.y; a.ldy = N; a.Yp = o.yp; a.yp_chunks = N / 32; a.x_scale = 1.f;
        if (epi == 1) { a.yp_f16 = 2; a.yp_scale = 8.f; }
        if (epi == 2) { a.act_post = 1; a.pe = pe; a.pe_ld = N; a.pe_alpha = alpha; a.x_scale = 1.25f; }
        return a;
    };
    const double flop = 2.0 * R * (double)N * C;
    printf("row-complete GEMM + LayerNorm: R = %d rows, C = %d, N = %d (%.1f GFLOP)\n", R, C, N, flop * 1e-9);
    size_t bad = 0;
    for (int epi = 0; epi < 3; ++epi) {
        printf("  EPI %d (%s)\n", epi, epi == 0 ? "LN -> rows + split-bf16 planes" : (epi == 1 ? "LN -> rows + mx planes" : "LN -> ReLU -> PE -> rows + planes"));
        CK(hipMemset(ref.y, 0xff, ybytes)); CK(hipMemset(ref.yp, 0xff, ybytes));
        launch_row8<2>(args(epi, 0, ref));
        CK(hipDeviceSynchronize());
        auto check = [&](const char* nm, auto launch) {
            CK(hipMemset(out[0].y, 0xff, ybytes)); CK(hipMemset(out[0].yp, 0xff, ybytes));
            launch(args(epi, 0, out[0]));
            CK(hipDeviceSynchronize());
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { printf("    %s: %s\n", nm, hipGetErrorString(e)); ++bad; return; }
            printf("   %s vs row8<128 rows>:\n", nm);
            bad += compare("rows", ref.y, out[0].y, (size_t)R * N * 4);
            bad += compare("planes", ref.yp, out[0].yp, (size_t)R * N * 4);
        };
        if (epi == 0) { check("row4<128,s2>", [&](const GemmArgs& a) { launch_row4<4, 0, 2>(a); }); check("row4<160,s2>", [&](const GemmArgs& a) { launch_row4<5, 0, 2>(a); }); check("row4<160,s0>", [&](const GemmArgs& a) { launch_row4<5, 0, 0>(a); }); check("row4<160,s4>", [&](const GemmArgs& a) { launch_row4<5, 0, 4>(a); }); }
        if (epi == 1) { check("row4<128,s2>", [&](const GemmArgs& a) { launch_row4<4, 1, 2>(a); }); check("row4<160,s2>", [&](const GemmArgs& a) { launch_row4<5, 1, 2>(a); }); }
        if (epi == 2) { check("row4<128,s2>", [&](const GemmArgs& a) { launch_row4<4, 2, 2>(a); }); check("row4<160,s2>", [&](const GemmArgs& a) { launch_row4<5, 2, 2>(a); }); }
    }
    printf("bit-identity: %s\n", bad ? "FAILED" : "ok");
    // ---- the planes-only forms (RES 1 / 2 / 3): residual read from split-bf16 / mx planes, no fp32 rows out.  hi + lo and fp16 + e4m3 x 2^-(ka+11) are
    // exact in fp32, so gemm_row8_bf16 on the RECONSTRUCTED fp32 residual must give the same planes bit for bit.
    {
        void *rpb[2], *rpm[2]; float *rrb, *rrm;
        const int kra = 3;      // static scale 2^ka of the mx residual planes (|x| < 1 here; the model's comes from the LayerNorm bound)
        const int64_t nres = (int64_t)Rpad * (N / 4);
        CK(hipMalloc(&rrb, hres.size() * 4)); CK(hipMalloc(&rrm, hres.size() * 4));
        for (int sI = 0; sI < 2; ++sI) {
            CK(hipMalloc(&rpb[sI], hres.size() * 4)); CK(hipMalloc(&rpm[sI], hres.size() * 4));
            hipLaunchKernelGGL(to_planes, dim3((unsigned)((nres + 255) / 256)), dim3(256), 0, 0, res[sI], N, N, Rpad, N / 32, rpb[sI], 0, 1.f);
            hipLaunchKernelGGL(to_planes, dim3((unsigned)((nres + 255) / 256)), dim3(256), 0, 0, res[sI], N, N, Rpad, N / 32, rpm[sI], 2, exp2f((float)kra));
        }
        hipLaunchKernelGGL(planes_to_rows, dim3((unsigned)((nres + 255) / 256)), dim3(256), 0, 0, rpb[0], N / 32, Rpad, N, rrb, N);
        hipLaunchKernelGGL(mx_planes_to_rows, dim3((unsigned)(((int64_t)Rpad * N + 255) / 256)), dim3(256), 0, 0, rpm[0], N, Rpad, exp2f(-(float)(kra + 11)), rrm);
        CK(hipDeviceSynchronize());
        auto po_args = [&](int epi, int res_kind, int set, const Bufs& o) {
            GemmArgs a = args(epi, set, o);
            a.Y = nullptr; a.resid = nullptr;
            if (res_kind == 1) { a.residp = rpb[set]; a.residp_chunks = N / 32; a.residp_mx = 0; a.residp_scale = 1.f; }
            if (res_kind == 2) { a.residp = rpm[set]; a.residp_chunks = N / 32; a.residp_mx = 1; a.residp_scale = exp2f(-(float)(kra + 11)); }
            return a;
        };
        size_t bad2 = 0;
        auto check_po = [&](const char* nm, int epi, int res_kind, auto launch) {
            GemmArgs r8 = args(epi, 0, ref);
            if (res_kind) r8.resid = res_kind == 1 ? rrb : rrm;
            CK(hipMemset(ref.yp, 0xff, ybytes)); CK(hipMemset(out[0].yp, 0xff, ybytes)); CK(hipMemset(out[0].y, 0xee, ybytes));
            launch_row8<2>(r8);
            launch(po_args(epi, res_kind, 0, out[0]));
            CK(hipDeviceSynchronize());
            printf("   %s vs row8<128 rows> on the reconstructed residual:\n", nm);
            bad2 += compare("planes", ref.yp, out[0].yp, (size_t)R * N * 4);
            std::vector<unsigned> probe(64);
            CK(hipMemcpy(probe.data(), out[0].y, 256, hipMemcpyDeviceToHost));
            for (unsigned w : probe) if (w != 0xeeeeeeeeu) { printf("    fp32 rows were written\n"); ++bad2; break; }
        };
        check_po("EPI 0, residual from split-bf16 planes, 128 rows", 0, 1, [&](const GemmArgs& a) { launch_row4_po<4, 0, 0, 1>(a); });
        check_po("EPI 0, residual from split-bf16 planes, 160 rows", 0, 1, [&](const GemmArgs& a) { launch_row4_po<5, 0, 0, 1>(a); });
        check_po("EPI 0, residual from mx planes, 160 rows", 0, 2, [&](const GemmArgs& a) { launch_row4_po<5, 0, 0, 2>(a); });
        check_po("EPI 0, residual from mx planes, 128 rows", 0, 2, [&](const GemmArgs& a) { launch_row4_po<4, 0, 0, 2>(a); });
        check_po("EPI 1 (mx planes out), residual from split-bf16 planes, 160 rows", 1, 1, [&](const GemmArgs& a) { launch_row4_po<5, 1, 0, 1>(a); });
        check_po("EPI 1 (mx planes out), residual from split-bf16 planes, 128 rows", 1, 1, [&](const GemmArgs& a) { launch_row4_po<4, 1, 0, 1>(a); });
        check_po("EPI 2 (input layer), no residual, 160 rows", 2, 0, [&](const GemmArgs& a) { launch_row4_po<5, 2, 0, 3>(a); });
        check_po("EPI 2 (input layer), no residual, 128 rows", 2, 0, [&](const GemmArgs& a) { launch_row4_po<4, 2, 0, 3>(a); });
        printf("planes-only forms: %s\n", bad2 ? "FAILED" : "ok");
        bad += bad2;
        for (int pass = 0; pass < 3; ++pass) {      // (interleaved with the rounds-1-5 forms, three times: the order of measurement matters by ~5 % on a warm chip)
            auto Tp = [&](const char* nm, int epi, int res_kind, auto launch) {
                const float us = time_kernel([&](int i) { launch(po_args(epi, res_kind, i & 1, out[i & 1])); }, reps);
                printf("  planes-only  %-44s %8.1f us   %7.1f TFLOP/s\n", nm, us, flop / us * 1e-6);
            };
            Tp("EPI 0 resid bf16 planes row4<160>", 0, 1, [&](const GemmArgs& a) { launch_row4_po<5, 0, 0, 1>(a); });
            Tp("EPI 0 resid mx planes   row4<160>", 0, 2, [&](const GemmArgs& a) { launch_row4_po<5, 0, 0, 2>(a); });
            Tp("EPI 1 resid bf16 planes row4<160>", 1, 1, [&](const GemmArgs& a) { launch_row4_po<5, 1, 0, 1>(a); });
            Tp("EPI 2 no residual       row4<160>", 2, 0, [&](const GemmArgs& a) { launch_row4_po<5, 2, 0, 3>(a); });
            Tp("EPI 0 resid bf16 planes row4<128>", 0, 1, [&](const GemmArgs& a) { launch_row4_po<4, 0, 0, 1>(a); });
            auto Tl = [&](const char* nm, int epi, auto launch) {
                const float us = time_kernel([&](int i) { launch(args(epi, i & 1, out[i & 1])); }, reps);
                printf("  rows+planes  %-44s %8.1f us   %7.1f TFLOP/s\n", nm, us, flop / us * 1e-6);
            };
            Tl("EPI 0 resid fp32 rows    row4<160>", 0, [&](const GemmArgs& a) { launch_row4<5, 0, 2>(a); });
            Tl("EPI 1 resid fp32 rows    row4<160>", 1, [&](const GemmArgs& a) { launch_row4<5, 1, 2>(a); });
            Tl("EPI 2 no residual        row4<160>", 2, [&](const GemmArgs& a) { launch_row4<5, 2, 2>(a); });
        }
    }
    // ---- timing (two operand / output sets in turn: 2 x (A planes + residual + rows + planes) exceed the MALL at the c3 shape)
    for (int epi = 0; epi < 3; ++epi) {
        auto T = [&](const char* nm, auto launch) {
            const float us = time_kernel([&](int i) { launch(args(epi, i & 1, out[i & 1])); }, reps);
            printf("  EPI %d  %-16s %8.1f us   %7.1f TFLOP/s\n", epi, nm, us, flop / us * 1e-6);
        };
        T("row8<128>", [&](const GemmArgs& a) { launch_row8<2>(a); });
        T("row8<192>", [&](const GemmArgs& a) { launch_row8<3>(a); });
        if (epi == 0) { T("row4<128,s2>", [&](const GemmArgs& a) { launch_row4<4, 0, 2>(a); }); T("row4<160,s2>", [&](const GemmArgs& a) { launch_row4<5, 0, 2>(a); });
                        T("row4<160,s0>", [&](const GemmArgs& a) { launch_row4<5, 0, 0>(a); }); T("row4<160,s4>", [&](const GemmArgs& a) { launch_row4<5, 0, 4>(a); }); }
        if (epi == 1) { T("row4<128,s2>", [&](const GemmArgs& a) { launch_row4<4, 1, 2>(a); }); T("row4<160,s2>", [&](const GemmArgs& a) { launch_row4<5, 1, 2>(a); }); }
        if (epi == 2) { T("row4<128,s2>", [&](const GemmArgs& a) { launch_row4<4, 2, 2>(a); }); T("row4<160,s2>", [&](const GemmArgs& a) { launch_row4<5, 2, 2>(a); }); }
    }
    // ---- the mx arithmetic (ARITH = 2) against split-bf16 on CONSISTENT operands: one fp32 matrix (non-negative, like a ReLU output) and one fp32 weight,
    // turned into both plane / image formats by the library's own kernels; static scales from a deliberately loose bound (x 200, as in the model)
    if (C % 128 == 0 && (size_t)R * C * 4 < (size_t)3 << 30) {
        std::vector<float> hxf((size_t)Rpad * C), hwf((size_t)N * C);
        for (auto& v : hxf) v = 3.f * (rand() & 0xffff) / 65536.f * ((rand() & 3) ? 1.f : 0.f);
        for (auto& v : hwf) v = 0.06f * ((rand() & 0xffff) - 32768) / 32768.f;
        float *xf, *wf; void *xb, *xm, *wbf, *wmx;
        CK(hipMalloc(&xf, hxf.size() * 4)); CK(hipMemcpy(xf, hxf.data(), hxf.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&wf, hwf.size() * 4)); CK(hipMemcpy(wf, hwf.data(), hwf.size() * 4, hipMemcpyHostToDevice));
        CK(hipMalloc(&xb, hxf.size() * 4)); CK(hipMalloc(&xm, hxf.size() * 4)); CK(hipMalloc(&wbf, hwf.size() * 4)); CK(hipMalloc(&wmx, hwf.size() * 4));
        const int kh = (int)floorf(log2f(448.f / (3.f * 200.f))), kw = (int)floorf(log2f(448.f / 0.06f));
        const int64_t nx = (int64_t)Rpad * (C / 4);
        hipLaunchKernelGGL(to_planes, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, 0, xf, C, C, Rpad, nchunks, xb, 0, 1.f);
        hipLaunchKernelGGL(to_planes, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, 0, xf, C, C, Rpad, nchunks, xm, 2, exp2f((float)kh));
        const int64_t tb = (int64_t)N * nchunks * 32;
        hipLaunchKernelGGL(repack_weight_bf16, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, 0, wf, N, C, 1, N, nchunks, (const float*)nullptr, (const float*)nullptr, 0.f, (__bf16*)wbf, 0, 0);
        const size_t mxb = mx_image_bytes(N, C, 1);
        hipLaunchKernelGGL(repack_weight_mx, dim3((unsigned)((mxb / 2 + 255) / 256)), dim3(256), 0, 0, wf, N, C, 1, N, kw, (unsigned short*)wmx);
        CK(hipDeviceSynchronize());
        auto b4 = [](int e) { const int b = e < 1 ? 1 : (e > 254 ? 254 : e); return b * 0x01010101; };
        GemmArgs ab = args(0, 0, ref);
        ab.Xp = xb; ab.W = (const float*)wbf; ab.Wb = wbf;
        GemmArgs am = args(0, 0, out[0]);
        am.Xp = xm; am.W = (const float*)wmx; am.Wb = wmx; am.mx = 1; am.mx_scale = b4(127 - kh - 11); am.mx_scale_b = b4(127 - kw);
        launch_row8<2>(ab);
        CK(hipDeviceSynchronize());
        std::vector<float> yr((size_t)R * N), ym((size_t)R * N);
        CK(hipMemcpy(yr.data(), ref.y, yr.size() * 4, hipMemcpyDeviceToHost));
        for (int mtv = 4; mtv <= 5; ++mtv) {
            CK(hipMemset(out[0].y, 0xff, ybytes));
            if (mtv == 4) launch_row4_mx<4>(am); else launch_row4_mx<5>(am);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(ym.data(), out[0].y, ym.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0, mag = 0; size_t nanc = 0;
            for (size_t i = 0; i < yr.size(); ++i) { if (!(ym[i] == ym[i])) ++nanc; worst = fmax(worst, fabs((double)ym[i] - yr[i])); mag = fmax(mag, fabs((double)yr[i])); }
            printf("  mx (fp16 + e4m3 cross terms, scales 2^%d / 2^%d) row4<%d rows> vs split-bf16 row8: max |diff| %.3e at max |y| %.2f, %zu NaN%s\n", kh, kw, 32 * mtv, worst, mag, nanc,
                   (worst > 1e-3 || nanc) ? "  FAILED" : "");
            if (worst > 1e-3 || nanc) ++bad;
        }
        auto Tm = [&](const char* nm, auto launch) {
            const float us = time_kernel([&](int) { launch(); }, reps);
            printf("  FFN2-like  %-18s %8.1f us   %7.1f TFLOP/s\n", nm, us, flop / us * 1e-6);
        };
        Tm("row8<128> bf16x3", [&] { launch_row8<2>(ab); });
        Tm("row4<160> bf16x3", [&] { launch_row4<5, 0, 2>(ab); });
        Tm("row4<128> mx", [&] { launch_row4_mx<4>(am); });
        Tm("row4<160> mx", [&] { launch_row4_mx<5>(am); });
    }
    // ---- the fused QKV projection's passes (EPI 3) against gemm_qkv8_bf16 (passes apart, 128 rows): Q | K planes and V^T planes bit for bit, then times
    if (C == 384) {
        const int D = 384, Rvt = Rpad;
        std::vector<unsigned short> hw3;
        {   // weight image [3 D][nchunks][hi 32 | lo 32]
            std::vector<unsigned short> tmp;
            hw3.resize((size_t)3 * D * nchunks * 64);
            for (size_t i = 0; i < (size_t)3 * D * nchunks; ++i)
                for (int k = 0; k < 32; ++k) {
                    const float v = 0.06f * ((rand() & 0xffff) - 32768) / 32768.f;
                    const unsigned short hi = bf16_of(v);
                    hw3[i * 64 + k] = hi; hw3[i * 64 + 32 + k] = bf16_of(v - f_of(hi));
                }
        }
        std::vector<float> hb3(3 * D);
        for (int i = 0; i < 3 * D; ++i) hb3[i] = 0.01f * (i % 11) - 0.05f;
        void *w3; float* b3; void *qh[2], *ql[2], *vh[2], *vl[2];
        CK(hipMalloc(&w3, hw3.size() * 2)); CK(hipMemcpy(w3, hw3.data(), hw3.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&b3, hb3.size() * 4)); CK(hipMemcpy(b3, hb3.data(), hb3.size() * 4, hipMemcpyHostToDevice));
        const size_t qkb = (size_t)Rvt * 2 * D * 2, vtb = (size_t)D * Rvt * 2;
        for (int s = 0; s < 2; ++s) { CK(hipMalloc(&qh[s], qkb)); CK(hipMalloc(&ql[s], qkb)); CK(hipMalloc(&vh[s], vtb)); CK(hipMalloc(&vl[s], vtb)); }
        auto qargs = [&](int s) {
            GemmArgs a;
            memset(&a, 0, sizeof a);
            a.C = C; a.Cpad = C; a.ktaps = 1; a.N = 3 * D; a.R = R; a.W = (const float*)w3; a.Wb = w3; a.Xp = xp[0]; a.row_pos = pos; a.bias = b3;
            a.qk_hi = qh[s]; a.qk_lo = ql[s]; a.vt_hi = vh[s]; a.vt_lo = vl[s]; a.att_D = D; a.Rvt = Rvt; a.q_scale = 0.104f; a.x_scale = 1.f;
            return a;
        };
        auto l8 = [&](const GemmArgs& a) {
            constexpr size_t lds = qkv8_lds_bytes<3, 2>();
            static bool done = false;
            if (!done) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_qkv8_bf16<3, 3, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); done = true; }
            hipLaunchKernelGGL((gemm_qkv8_bf16<3, 3, 2, true>), dim3((a.Rvt + 127) / 128, 3), dim3(512), lds, 0, a);
        };
        auto l4 = [&](const GemmArgs& a, int mt) {
            if (mt == 4) {
                static bool d4 = false;
                if (!d4) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_row4_bf16<3, 3, 4, 3, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)row4_lds_bytes<3, 4>())); d4 = true; }
                hipLaunchKernelGGL((gemm_row4_bf16<3, 3, 4, 3, 2, 0>), dim3((a.Rvt + 127) / 128, 3), dim3(256), (row4_lds_bytes<3, 4>()), 0, a);
            } else {
                static bool d5 = false;
                if (!d5) { CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_row4_bf16<3, 3, 5, 3, 2, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)row4_lds_bytes<3, 5>())); d5 = true; }
                hipLaunchKernelGGL((gemm_row4_bf16<3, 3, 5, 3, 2, 0>), dim3((a.Rvt + 159) / 160, 3), dim3(256), (row4_lds_bytes<3, 5>()), 0, a);
            }
        };
        for (int s = 0; s < 2; ++s) { CK(hipMemset(qh[s], 0x7f, qkb)); CK(hipMemset(ql[s], 0x7f, qkb)); CK(hipMemset(vh[s], 0x7f, vtb)); CK(hipMemset(vl[s], 0x7f, vtb)); }
        l8(qargs(0));
        CK(hipDeviceSynchronize());
        for (int mt = 4; mt <= 5; ++mt) {
            CK(hipMemset(qh[1], 0x7f, qkb)); CK(hipMemset(ql[1], 0x7f, qkb)); CK(hipMemset(vh[1], 0x7f, vtb)); CK(hipMemset(vl[1], 0x7f, vtb));
            l4(qargs(1), mt);
            CK(hipDeviceSynchronize());
            printf("   QKV passes on row4<%d rows> vs gemm_qkv8_bf16<128 rows, passes apart>:\n", 32 * mt);
            // (compare the rows every kernel must write: [0, Rvt rounded down to the common tile coverage) = all Rvt rows)
            bad += compare("qk hi", qh[0], qh[1], qkb, true); bad += compare("qk lo", ql[0], ql[1], qkb, true);
            bad += compare("vt hi", vh[0], vh[1], vtb, true); bad += compare("vt lo", vl[0], vl[1], vtb, true);
        }
        const double qflop = 2.0 * R * 3.0 * D * C;
        auto Tq = [&](const char* nm, auto launch) {
            const float us = time_kernel([&](int i) { launch(i & 1); }, reps);
            printf("  QKV  %-22s %8.1f us   %7.1f TFLOP/s\n", nm, us, qflop / us * 1e-6);
        };
        Tq("qkv8<128, apart>", [&](int s) { l8(qargs(s)); });
        Tq("row4<128> passes", [&](int s) { l4(qargs(s), 4); });
        Tq("row4<160> passes", [&](int s) { l4(qargs(s), 5); });
    }
#ifdef FS2_ROW_TIMING
    for (int mtv = 4; mtv <= 5; ++mtv) {
        long long z[8][8]; memset(z, 0, sizeof z);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_row_phase), z, sizeof z));
        if (mtv == 4) launch_row4<4, 0, 2>(args(0, 0, out[0])); else launch_row4<5, 0, 2>(args(0, 0, out[0]));
        CK(hipDeviceSynchronize());
        CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_row_phase), sizeof z));
        printf("  phase stamps row4<%d rows> (shader cycles from entry; wave 0 of workgroups 0, 32, ...): landed | k-loop done | LN stats | stores issued\n", 32 * mtv);
        for (int w = 0; w < 8; ++w) if (z[w][4]) printf("    wg %3d: %7lld | %7lld | %7lld | %7lld\n", 32 * w, z[w][1] - z[w][0], z[w][2] - z[w][0], z[w][3] - z[w][0], z[w][4] - z[w][0]);
    }
#endif
    printf("probe: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
