// attn_w32<DK> against attn_bf16<DK,3> on the same split-bf16 operands: max |difference| of the context (both are the same
// arithmetic up to the summation order and the deferred rescale), then the time of each kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fastspeech2_amd/csrc tools/probes/attn_w32_probe.hip -o tools/probes/attn_w32_probe.bin
//   attn_w32_probe.bin <B> <Lmin> <Lmax> [dk = 192] [reps = 5] [klen_slack = 0] [spike = 0]
// spike > 0: one key per utterance (in its third tile) gets K scaled by `spike`, so that rows meet a score far above their first
// tile's maximum -- above 2^64 the wave leaves the fast path (attn_w32_rows_slow).  Small cases (<= 2e8 score pairs) are also checked
// against a double-precision host reference of the same split operands.
// Utterance b has a length in [Lmin, Lmax] (deterministic spread); with klen_slack > 0 the last `slack` rows of every utterance are
// masked keys holding NaN (the padded_compat form).  Rows between utterances hold NaN in K and V^T as well.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "gemm_bf16.h"
#include "attn_bf16.h"
#include "attn_w32.h"
using namespace fs2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffff) > 0x7f800000) return 0x7fc0;
    u += 0x7fff + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <int DK>
int run(int B, int Lmin, int Lmax, int reps, int slack, float spike) {
    const int heads = 2, D = heads * DK;
    std::vector<int> start(B), len(B), klen(B);
    int row = 8;
    long long pairs = 0;
    for (int b = 0; b < B; ++b) {
        len[b] = Lmin + (int)(((long long)(Lmax - Lmin) * ((b * 37) % 101)) / 100);
        klen[b] = std::max(1, len[b] - slack);
        row = (row + 7) & ~7; start[b] = row; row += len[b] + 8;
        pairs += (long long)len[b] * klen[b];
    }
    row += 8;
    const int Rvt = (row + 127) & ~127;
    std::vector<int2> work;
    {   // eight interleaved queues, longest first (as build_work_list)
        std::vector<int> order(B);
        for (int b = 0; b < B; ++b) order[b] = b;
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return klen[x] > klen[y]; });
        std::vector<int2> q[8];
        for (int b : order) {
            int j = 0;
            for (int t = 1; t < 8; ++t) if (q[t].size() < q[j].size()) j = t;
            for (int i = 0; i * kAttBlk < len[b]; ++i) q[j].push_back(make_int2(b, i));
        }
        size_t depth = 0;
        for (int j = 0; j < 8; ++j) depth = std::max(depth, q[j].size());
        work.assign(depth * 8, make_int2(-1, 0));
        for (int j = 0; j < 8; ++j) for (size_t i = 0; i < q[j].size(); ++i) work[i * 8 + j] = q[j][i];
    }
    // operands: fp32 values -> hi / lo bf16 planes; NaN wherever a key must never be read
    std::vector<unsigned short> qkh((size_t)Rvt * 2 * D, 0x7fc0), qkl((size_t)Rvt * 2 * D, 0x7fc0), vth((size_t)D * Rvt, 0x7fc0), vtl((size_t)D * Rvt, 0x7fc0);
    srand(1234);
    auto rnd = [] { return (float)((rand() & 0xffff) - 32768) / 32768.f; };
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < len[b]; ++t) {
            const size_t r = (size_t)start[b] + t;
            for (int c = 0; c < 2 * D; ++c) {
                const bool is_k = c >= D;
                if (is_k && t >= klen[b]) continue;                  // masked key rows stay NaN
                float x = rnd() * (is_k ? 1.5f : 0.6f);
                if (is_k && spike > 0.f && t == 70 && t < klen[b]) x *= spike;
                const unsigned short hh = f2bf(x);
                qkh[r * 2 * D + c] = hh; qkl[r * 2 * D + c] = f2bf(x - bf2f(hh));
            }
            if (t < klen[b])
                for (int n = 0; n < D; ++n) {
                    const float x = rnd() * 2.f;
                    const unsigned short hh = f2bf(x);
                    vth[(size_t)n * Rvt + r] = hh; vtl[(size_t)n * Rvt + r] = f2bf(x - bf2f(hh));
                }
        }
    __bf16 *dq, *dv; float *ctx0, *ctx1; int* dmeta; int2* dwork;
    const size_t qbytes = qkh.size() * 2, vbytes = vth.size() * 2;
    CK(hipMalloc(&dq, 2 * qbytes)); CK(hipMalloc(&dv, 2 * vbytes));
    CK(hipMemcpy(dq, qkh.data(), qbytes, hipMemcpyHostToDevice)); CK(hipMemcpy((char*)dq + qbytes, qkl.data(), qbytes, hipMemcpyHostToDevice));
    CK(hipMemcpy(dv, vth.data(), vbytes, hipMemcpyHostToDevice)); CK(hipMemcpy((char*)dv + vbytes, vtl.data(), vbytes, hipMemcpyHostToDevice));
    CK(hipMalloc(&ctx0, (size_t)Rvt * D * 4)); CK(hipMalloc(&ctx1, (size_t)Rvt * D * 4));
    CK(hipMemset(ctx0, 0, (size_t)Rvt * D * 4)); CK(hipMemset(ctx1, 0, (size_t)Rvt * D * 4));
    CK(hipMalloc(&dmeta, 3 * B * 4)); CK(hipMalloc(&dwork, work.size() * 8));
    CK(hipMemcpy(dmeta, start.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dmeta + B, len.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmeta + 2 * B, klen.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dwork, work.data(), work.size() * 8, hipMemcpyHostToDevice));
    AttnB16Args a;
    memset(&a, 0, sizeof a);
    a.qk_hi = dq; a.qk_lo = (const __bf16*)((const char*)dq + qbytes); a.ldqk = 2 * D;
    a.vt_hi = dv; a.vt_lo = (const __bf16*)((const char*)dv + vbytes); a.Rvt = Rvt; a.ldc = D; a.ctxp = nullptr; a.ctxp_chunks = D / 32;
    a.start = dmeta; a.len = dmeta + B; a.klen = dmeta + 2 * B; a.work = dwork; a.nwork = nullptr; a.nitems = (int)work.size(); a.D = D; a.mask_q = slack > 0;
    a.qk_lo_bytes = (unsigned)qbytes; a.vt_lo_bytes = (unsigned)vbytes;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bf16<DK, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_b16_lds_bytes<DK>()));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_w32<DK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_w32_lds_bytes<DK>()));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flops = 4.0 * DK * heads * (double)pairs;
    float best0 = 1e9f, best1 = 1e9f;
    for (int rep = 0; rep < reps; ++rep) {
        float ms;
        a.ctx = ctx0;
        hipEventRecord(e0);
        hipLaunchKernelGGL((attn_bf16<DK, 3>), dim3(att_grid64((int)work.size()), heads), dim3(256), attn_b16_lds_bytes<DK>(), 0, a);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        hipEventElapsedTime(&ms, e0, e1); best0 = std::min(best0, ms);
        a.ctx = ctx1;
        if (getenv("W32_PLANES")) { a.ctx = nullptr; a.ctxp = ctx1; }      // the model's output form (context as planes only: the LDS-staged epilogue); timing only, the comparison below is then meaningless
#ifdef FS2_W32_TIMING
        { long long zero[16] = {0}; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_w32_phase), zero, sizeof zero)); }
#endif
        hipEventRecord(e0);
        hipLaunchKernelGGL((attn_w32<DK>), dim3((unsigned)work.size(), heads), dim3(256), attn_w32_lds_bytes<DK>(), 0, a);
        hipEventRecord(e1); CK(hipDeviceSynchronize());
        hipEventElapsedTime(&ms, e0, e1); best1 = std::min(best1, ms);
        a.ctxp = nullptr;
#ifdef FS2_W32_TIMING
        if (rep == reps - 1) {
            long long ph[16]; CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_w32_phase), sizeof ph));
            const double n = (double)std::max(1LL, ph[4]);
            printf("  wave 0 of workgroup 0, cycles per tile over %lld tiles: phase A (Q.K^T + exponentials + DMA issue) %.0f | phase B (P.V + next head) %.0f | DMA wait %.0f | barrier %.0f\n",
                   ph[4], ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n);
            printf("  the same wave, cycles outside the tile loop: kernel entry -> K(0) landed + barrier %lld | -> Q.K^T(0), first maximum, K(1) K(2) V(0) V(1) landed + barrier %lld | epilogue (drain, stores issued and landed) %lld\n",
                   ph[5], ph[6], ph[7]);
            printf("  entry in detail: work item + start/len/klen loaded %lld | DMA offsets + fragment addresses computed %lld | K(0) DMA + Q loads issued %lld | (then O = 0, wait, barrier: the rest of the first figure)\n", ph[8], ph[9], ph[10]);
        }
#endif
    }
    std::vector<float> c0((size_t)Rvt * D), c1((size_t)Rvt * D);
    CK(hipMemcpy(c0.data(), ctx0, c0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), ctx1, c1.size() * 4, hipMemcpyDeviceToHost));
    double maxd = 0, maxv = 0; long long bad = 0, nan0 = 0, nan1 = 0;
    for (int b = 0; b < B; ++b)
        for (int t = 0; t < len[b]; ++t)
            for (int c = 0; c < D; ++c) {
                const size_t i = ((size_t)start[b] + t) * D + c;
                if (std::isnan(c0[i])) ++nan0;
                if (std::isnan(c1[i])) ++nan1;
                const double d = fabs((double)c0[i] - c1[i]);
                if (!(d <= 2e-5)) ++bad;
                if (d > maxd) maxd = d;
                maxv = std::max(maxv, (double)fabs(c0[i]));
            }
    // host reference (double) of the same operands for small cases
    double maxr0 = -1, maxr1 = -1;
    if ((double)pairs * heads <= 2e8 / DK * 16) {
        maxr0 = maxr1 = 0;
        std::vector<double> sc, ov(DK);
        for (int b = 0; b < B; ++b)
            for (int hh = 0; hh < heads; ++hh)
                for (int t = 0; t < len[b]; ++t) {
                    const size_t rq = (size_t)start[b] + t;
                    sc.assign(klen[b], 0.0);
                    double m = -1e300;
                    for (int k = 0; k < klen[b]; ++k) {
                        const size_t rk = (size_t)start[b] + k;
                        double d = 0;
                        for (int c = 0; c < DK; ++c) {
                            const size_t iq = rq * 2 * D + hh * DK + c, ik = rk * 2 * D + D + hh * DK + c;
                            d += ((double)bf2f(qkh[iq]) + bf2f(qkl[iq])) * ((double)bf2f(qkh[ik]) + bf2f(qkl[ik]));
                        }
                        sc[k] = d; m = std::max(m, d);
                    }
                    double l = 0;
                    std::fill(ov.begin(), ov.end(), 0.0);
                    for (int k = 0; k < klen[b]; ++k) {
                        const double pk = exp2(sc[k] - m);
                        l += pk;
                        const size_t rk = (size_t)start[b] + k;
                        for (int c = 0; c < DK; ++c) {
                            const size_t iv = (size_t)(hh * DK + c) * Rvt + rk;
                            ov[c] += pk * ((double)bf2f(vth[iv]) + bf2f(vtl[iv]));
                        }
                    }
                    const bool dead = slack > 0 && t >= klen[b];
                    for (int c = 0; c < DK; ++c) {
                        const double ref = dead ? 0.0 : ov[c] / l;
                        const size_t i = rq * D + hh * DK + c;
                        maxr0 = std::max(maxr0, fabs(ref - c0[i]));
                        const double d1 = fabs(ref - c1[i]);
                        if (!(d1 <= 1e-4) && bad < 8) { printf("  w32 wrong at utt %d row %d head %d ch %d: %g, reference %g, attn_bf16 %g\n", b, t, hh, c, c1[i], ref, c0[i]); }
                        maxr1 = std::max(maxr1, std::isnan(c1[i]) ? 1e30 : d1);
                    }
                }
    }
    printf("dk=%d B=%d L=[%d,%d] slack=%d spike=%g items=%zu: attn_bf16 %.1f us (%.1f TF/s)  attn_w32 %.1f us (%.1f TF/s)  speedup %.3f | max|diff| %.3e (max|ctx| %.2f) bad %lld nan %lld/%lld | vs host reference: attn_bf16 %.3e attn_w32 %.3e\n",
           DK, B, Lmin, Lmax, slack, (double)spike, work.size(), best0 * 1e3, flops / best0 * 1e-9, best1 * 1e3, flops / best1 * 1e-9, best0 / best1, maxd, maxv, bad, nan0, nan1, maxr0, maxr1);
    hipFree(dq); hipFree(dv); hipFree(ctx0); hipFree(ctx1); hipFree(dmeta); hipFree(dwork);
    return (bad || nan1) ? 2 : 0;
}

// Issue rate of v_mfma_f32_32x32x16_bf16 from one wave: 96 MFMAs rotating over NACC accumulators (NACC = 1: every MFMA depends on its
// predecessor), cycles per MFMA from s_memtime.  One wave per SIMD (256 threads), one workgroup.
template <int NACC>
__global__ __launch_bounds__(256, 1) void mfma_chain(long long* out, const float* seed) {
    bf16x8_t a = __builtin_bit_cast(bf16x8_t, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}), b = a;
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = seed[threadIdx.x & 15];
    __builtin_amdgcn_sched_barrier(0);
    const long long t0 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 96; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NACC], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    const long long t1 = __builtin_readcyclecounter();
    __builtin_amdgcn_sched_barrier(0);
    float s_ = 0.f;
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s_ += acc[k][r];
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s_; }
}
// The same stream from every SIMD of the chip for long enough to reach the sustained clock: what a kernel of nothing but MFMAs gets.
template <int NACC>
__global__ __launch_bounds__(256, 1) void mfma_chip(long long* out, const float* seed, int iters, int random_data) {
    bf16x8_t a = __builtin_bit_cast(bf16x8_t, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}), b = a;
    if (random_data) {      // bf16 values of random sign and mantissa, exponents within +-2 of 1.0 (the accumulators stay finite: the products average out)
        unsigned x = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u), w[8];
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u; const unsigned lo = (x >> 8) & 0x81ffu; x = x * 1664525u + 1013904223u; const unsigned hi2 = (x >> 8) & 0x81ffu;
            w[i] = (0x3e80u | lo) | ((0x3e80u | hi2) << 16);
        }
        a = __builtin_bit_cast(bf16x8_t, u32x4{w[0], w[1], w[2], w[3]}); b = __builtin_bit_cast(bf16x8_t, u32x4{w[4], w[5], w[6], w[7]});
    }
    f32x16 acc[NACC];
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) acc[k][r] = seed[threadIdx.x & 15];
    long long tsum = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        const long long t0 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 96; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i % NACC], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (it >= iters / 2) tsum += __builtin_readcyclecounter() - t0;
        __builtin_amdgcn_sched_barrier(0);
    }
    float s_ = 0.f;
    for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s_ += acc[k][r];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = tsum; out[1] = (long long)s_; }
}
// The two MFMA streams of attn_w32's tile, bare (the kernel's own asm forms, nothing between them), from every SIMD of the chip:
// FORM 0 = phase A (accumulators in VGPRs, B operand in AGPRs, order ac am ac | ac am ac ...), FORM 1 = phase B (accumulators in literal
// AGPRs a[0:95], both operands VGPRs, two accumulators alternating), FORM 2 = phase A's order with all operands in VGPRs.
template <int FORM>
__global__ __launch_bounds__(256, 1) void mfma_form(long long* out, const float* seed, int iters) {
    bf16x8_t a = __builtin_bit_cast(bf16x8_t, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}), b = a, qa = a;
    f32x16 ac, am, z;
    for (int r = 0; r < 16; ++r) { ac[r] = seed[threadIdx.x & 15]; am[r] = ac[r]; z[r] = 0.f; }
    asm volatile("" : "+a"(qa));
    asm volatile("" : "+v"(a), "+v"(b));
    if (FORM == 1) fs2::for_seq([&](auto n_tag) __attribute__((always_inline)) { fs2::mfma_o0<decltype(n_tag)::value>(a); }, std::make_integer_sequence<int, 6>{});
    long long tsum = 0;
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_sched_barrier(0);
        const long long t0 = __builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        fs2::for_seq([&](auto s_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(s_tag)::value;
            if constexpr (FORM == 0) {
                if constexpr (S % 3 == 1) fs2::mfma_s<0>(am, a, qa, z); else fs2::mfma_s<0>(ac, b, qa, z);
            } else if constexpr (FORM == 2) {
                if constexpr (S % 3 == 1) am = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, am, 0, 0, 0); else ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, ac, 0, 0, 0);
            } else {
                constexpr int G = S / 6, U = S % 6, N0 = (G % 3) * 2;
                if constexpr (U & 1) fs2::mfma_o<N0 + 1, false>(a, b); else fs2::mfma_o<N0, false>(a, b);
            }
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 36>{});
        __builtin_amdgcn_sched_barrier(0);
        if (it >= iters / 2) tsum += __builtin_readcyclecounter() - t0;
        __builtin_amdgcn_sched_barrier(0);
    }
    float s_ = 0.f;
    for (int r = 0; r < 16; ++r) s_ += ac[r] + am[r];
    if (FORM == 1) { fs2::mfma_drain(); float e[16]; fs2::read_o<0>(e); s_ += e[0]; }
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = tsum; out[1] = (long long)s_; }
}
static int mfma_bench() {
    long long* d; float* sd; CK(hipMalloc(&d, 16)); CK(hipMalloc(&sd, 64)); CK(hipMemset(sd, 0, 64));
    long long h[2];
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int grids[4] = {256, 2048, 2048, 2048}, iters[4] = {2000, 2000, 250, 2000};
        for (int g = 0; g < 4; ++g) {
            float best = 1e30f;
            if (g == 3) printf("(next: operands of random sign / mantissa instead of all ones)\n");
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0); hipLaunchKernelGGL((mfma_chip<2>), dim3(grids[g]), dim3(256), 0, 0, d, sd, iters[g], g == 3 ? 1 : 0); hipEventRecord(e1);
                CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, e0, e1); best = std::min(best, ms);
            }
            CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            const double fl = (double)grids[g] * 4 * iters[g] * 96 * 32768.0;
            printf("whole chip, %d workgroups x 4 waves x %d x 96 MFMAs (2 accumulators): %.3f ms = %.0f TFLOP/s dense bf16; s_memtime cycles per MFMA in wave 0 (second half) %.1f\n",
                   grids[g], iters[g], best, fl / best * 1e-9, h[0] / (96.0 * (iters[g] - iters[g] / 2)));
        }
    }
    {
        #define FORM(F, what) { hipLaunchKernelGGL((mfma_form<F>), dim3(2048), dim3(256), 0, 0, d, sd, 1000); CK(hipDeviceSynchronize()); CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost)); \
            printf("whole chip, attn_w32's %s stream bare: %.1f s_memtime cycles per MFMA\n", what, h[0] / (36.0 * 500)); }
        FORM(0, "phase A (acc VGPR, B operand AGPR; ac am ac)") FORM(2, "phase A order, all VGPR (builtin)") FORM(1, "phase B (acc literal AGPR, two alternating)")
        #undef FORM
    }
    #define RUN(N) hipLaunchKernelGGL((mfma_chain<N>), dim3(1), dim3(256), 0, 0, d, sd); CK(hipDeviceSynchronize()); hipLaunchKernelGGL((mfma_chain<N>), dim3(1), dim3(256), 0, 0, d, sd); CK(hipDeviceSynchronize()); \
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost)); printf("v_mfma_f32_32x32x16_bf16, 96 in a row over %d accumulator(s): %.1f cycles per MFMA\n", N, h[0] / 96.0);
    RUN(1) RUN(2) RUN(3) RUN(4)
    #undef RUN
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "mfma")) return mfma_bench();
    const int B = argc > 1 ? atoi(argv[1]) : 64, Lmin = argc > 2 ? atoi(argv[2]) : 300, Lmax = argc > 3 ? atoi(argv[3]) : 800;
    const int dk = argc > 4 ? atoi(argv[4]) : 192, reps = argc > 5 ? atoi(argv[5]) : 5, slack = argc > 6 ? atoi(argv[6]) : 0;
    const float spike = argc > 7 ? (float)atof(argv[7]) : 0.f;
    if (dk == 192) return run<192>(B, Lmin, Lmax, reps, slack, spike);
    if (dk == 128) return run<128>(B, Lmin, Lmax, reps, slack, spike);
    printf("dk must be 128 or 192\n");
    return 1;
}
