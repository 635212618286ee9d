// gemm_pl8_conv: the conv form of gemm_pl_bf16 (gemm_planes.h) for grids that fill the chip, restructured around what the phase probe
// (tools/probes/mx_conv_probe.hip) showed once the MFMA work per k-step dropped from 96 to 64 issue slots (mx arithmetic): with ONE operand
// stage in flight a k-step cannot be shorter than an LDS-DMA round trip (~1.2-1.5 us under load), and two 4-wave workgroups per CU
// spent 30 % (split-bf16) to 45 % (mx) of a step parked at the barrier waiting for it.
//   * ONE 8-wave workgroup per CU: tile 512 x 128, waves 4(M) x 2(N), wave tile 128 x 64 as before (128 accumulator registers).  The
//     weight stage is shared by twice the rows: half the L2 -> LDS weight bytes and half the DMA instructions per MFMA.
//   * weight stages form a RING of NSTAGE (4) x 16 KB: the stage of step it + NSTAGE - 1 is requested at the top of step it, i.e. three
//     steps before it is read; completion is awaited with a COUNTED s_waitcnt vmcnt (LDS-DMA completes in issue order), never with
//     vmcnt(0) inside a chunk, so the round trip is off the critical path.
//   * the A tile (512 + halo rows of one 128-byte unit, 66 KB) stays single-buffered; at the last tap of a chunk every wave first reads
//     ALL its A fragments of that step into registers, the workgroup meets at a barrier, the refill for the next chunk is issued, and only
//     then the step's MFMAs run: the refill's round trip hides behind them.
//   * per-element arithmetic (order of the units, taps and MFMAs) is exactly that of gemm_pl_bf16 in the same arithmetic mode: results
//     are bit-identical, the launcher may pick either kernel from the grid size alone.
// LDS: (512 + 16) x 128 + 4 x 16384 = 133,120 B of the 160 KB.
#pragma once
#include "gemm_planes.h"      // REJECTED experiment (round 3): kept for tools/probes/mx_conv_probe.hip; not part of libfs2_hip.so

namespace fs2 {

constexpr int kPl8BM = 512;
template <int NSTAGE> constexpr size_t pl8_lds_bytes() { return (size_t)(kPl8BM + kMaxHalo) * 128 + (size_t)NSTAGE * kB16BN * 128; }

// ARITH as in gemm_pl_bf16: 0 split-bf16 (NSPLIT 3) / plain bf16 (NSPLIT 1), 1 fp16 images, 2 mx planes / mx weight image
template <int NSPLIT, int ARITH, int NSTAGE>
__global__ __launch_bounds__(512, 1) void gemm_pl8_conv(GemmArgs a) {
    constexpr int BM = kPl8BM, MT = 8, AROWS = BM + kMaxHalo;
    constexpr bool F16 = ARITH == 1;
    static_assert(NSTAGE >= 3 && NSTAGE <= 5, "ring of 3-5 weight stages");
    extern __shared__ __attribute__((aligned(16))) char smem_8[];
    char* As = smem_8;
    char* Bs0 = smem_8 + AROWS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: every LDS-DMA destination (M0) is SGPR arithmetic
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * kB16BN, m0 = blockIdx.y * BM;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int ktaps = a.ktaps;
    const int P = (ktaps - 1) >> 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int lp = rperm(lr);
    const __bf16* Wb = reinterpret_cast<const __bf16*>(a.W);
    const __bf16* Xp = reinterpret_cast<const __bf16*>(a.Xp);
    const int nchunks = a.Cpad / 32;
    const int niter = nchunks * ktaps;
    const int jrow = lane >> 3, jslot = lane & 7;

    // A: instruction q = w, w + 8, ... fills tile rows 8q + jrow (q & 1 == w & 1: the swizzle term is a per-lane constant, as in gemm_pl_bf16)
    const int a_instr = (BM + 2 * P + 7) >> 3;
    const int sA = jslot ^ (jrow >> 1) ^ ((wave & 1) << 2);
    const int arow0 = m0 - P + wave * 8 + jrow;
    const __bf16* a_src0 = Xp + (ptrdiff_t)arow0 * nchunks * 64 + sA * 8;
    const size_t a_qstride = (size_t)64 * nchunks * 64;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)smem_8);
    const unsigned ldsA = lds0 + wave * 1024, ldsB = lds0 + AROWS * 128 + wave * 1024;
    auto dma_A = [&](int ch) {
        unsigned dst = ldsA;
        const __bf16* src = a_src0 + (size_t)ch * 64;
        int row = arow0;
        for (int q = wave; q < a_instr; q += 8) {
            const bool ok = row >= 0 && row < a.R;
            const void* sp = ok ? static_cast<const void*>(src) : static_cast<const void*>(g_zero16);
            dma16(sp, dst);
            dst += 8192; src += a_qstride; row += 64;
        }
    };
    // B: instruction q = w + 8u (u = 0, 1) fills LDS rows 8q + jrow = 64u + 16 (w >> 1) + jB, i.e. n-tile (w >> 1) of column half u, tile row
    // jB = 8 (w & 1) + jrow; it receives weight row 64u + 4 rperm_inv(jB) + (w >> 1) (the permuted order of gemm_pl_bf16: a lane's four
    // accumulators of an m-tile are four consecutive channels)
    const int jB = (wave & 1) * 8 + jrow;
    const int sB = jslot ^ ((jB >> 1) & 7);
    const __bf16* b_src0 = Wb + ((size_t)(n0 + 4 * rperm_inv(jB) + (wave >> 1)) * niter) * 64 + sB * 8;
    const size_t b_u = (size_t)64 * niter * 64;
    auto dma_B = [&](int it, int stage) {
        const unsigned dst = ldsB + stage * (kB16BN * 128);
        const __bf16* src = b_src0 + (size_t)it * 64;
        dma16(src, dst);
        dma16(src + b_u, dst + 8192);
    };

    const int it_end = niter;
    dma_A(0);
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < it_end) dma_B(s, s);
    // accumulators start at bias + residual (loaded while the first tiles are in flight): acc[mt][nt][r] = row (mt, r), channel col + nt
    const int col = n0 + wn * 64 + 4 * lr;
    f32x4 acc[MT][4];
    {
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias && col < a.N) bv = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (BM / 4) + mt * 16 + rperm(lg * 4 + r);
                f32x4 v = bv;
                v += load4_or_zero(a.resid + (size_t)row * a.ldr + col, a.resid != nullptr && row < a.R && col < a.N);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt][r] = v[nt];
            }
    }
    int it = 0, stage = 0;      // stage = it % NSTAGE

    // the MFMAs of m-tile mt of one k-step, given the two 16-byte pieces (slot lg | slot 4 + lg) of its A row and of the four B rows;
    // KIND 0: split arithmetic per NSPLIT / ARITH (piece 0 = hi, piece 1 = lo), 1: mx unit of 64 fp16 channels, 2: mx unit of 128 e4m3 channels
    auto mfma_mt = [&](auto kind_tag, int mt, const bf16x8_t a0, const bf16x8_t a1, const bf16x8_t (&b0)[4], const bf16x8_t (&b1)[4]) {
        constexpr int KIND = decltype(kind_tag)::value;
        if constexpr (KIND == 2) {
            const v8i_t av = __builtin_shufflevector(__builtin_bit_cast(v4i_t, a0), __builtin_bit_cast(v4i_t, a1), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const v8i_t bv = __builtin_shufflevector(__builtin_bit_cast(v4i_t, b0[nt]), __builtin_bit_cast(v4i_t, b1[nt]), 0, 1, 2, 3, 4, 5, 6, 7);
                acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[mt][nt], 0, 0, 0, a.mx_scale, 0, a.mx_scale_b);
            }
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<true>(a0, b0[nt], acc[mt][nt]);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<true>(a1, b1[nt], acc[mt][nt]);
        } else {
            if (NSPLIT >= 2) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(a1, b0[nt], acc[mt][nt]);
            }
            if (NSPLIT == 3) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(a0, b1[nt], acc[mt][nt]);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(a0, b0[nt], acc[mt][nt]);
        }
    };

    auto k_loop = [&](auto kind_tag, const int cb, const int ce) {
        for (int chunk = cb; chunk < ce; ++chunk) {
            for (int tap = 0; tap < ktaps; ++tap, ++it) {
                // the weight stage of this step (requested NSTAGE - 1 steps ago) and, at the first tap of a chunk, the A tile have landed
                const int ahead = min(NSTAGE - 2, it_end - 1 - it);       // weight stages requested after this step's
                if (tap == 0 || ahead <= 0) dma_wait_barrier<0>();
                else if (ahead == 1) dma_wait_barrier<2>();
                else if (ahead == 2) dma_wait_barrier<4>();
                else dma_wait_barrier<6>();
                if (it + NSTAGE - 1 < it_end) dma_B(it + NSTAGE - 1, stage == 0 ? NSTAGE - 1 : stage - 1);
                const char* Bs = Bs0 + stage * (kB16BN * 128);
                bf16x8_t b0[4], b1[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n = wn * 64 + nt * 16 + lp;
                    b0[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + swz(n, lg));
                    b1[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + swz(n, 4 + lg));
                }
                const int rbase = wm * (BM / 4) + lp + tap;
#ifndef FS2_PL8_PRELOAD
#define FS2_PL8_PRELOAD 0
#endif
                if (FS2_PL8_PRELOAD && tap == ktaps - 1 && chunk + 1 < nchunks) {
                    // last tap of the chunk: fragments first, then the workgroup agrees that the A tile is dead and the refill goes out
                    // BEFORE this step's MFMAs
                    bf16x8_t af0[MT], af1[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        af0[mt] = *reinterpret_cast<const bf16x8_t*>(As + swz(rbase + mt * 16, lg));
                        af1[mt] = *reinterpret_cast<const bf16x8_t*>(As + swz(rbase + mt * 16, 4 + lg));
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    dma_A(chunk + 1);
                    if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) mfma_mt(kind_tag, mt, af0[mt], af1[mt], b0, b1);
                    if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(0);
                } else {
                    if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(As + swz(rbase + mt * 16, lg));
                        const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(As + swz(rbase + mt * 16, 4 + lg));
                        mfma_mt(kind_tag, mt, a0, a1, b0, b1);
                    }
                    if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(0);
                    if (!FS2_PL8_PRELOAD && tap == ktaps - 1 && chunk + 1 < nchunks) {
                        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        dma_A(chunk + 1);
                    }
                }
                stage = stage == NSTAGE - 1 ? 0 : stage + 1;
            }
        }
    };
    if constexpr (ARITH == 2) {
        k_loop(std::integral_constant<int, 1>{}, 0, nchunks >> 1);
        k_loop(std::integral_constant<int, 2>{}, nchunks >> 1, nchunks);
    } else {
        k_loop(std::integral_constant<int, 0>{}, 0, nchunks);
    }
    pl_epilogue<MT>(a, acc, m0 + wm * (BM / 4), col, lg);
}

}  // namespace fs2
