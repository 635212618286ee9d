// gemm_pp_conv: the conv form of gemm_pl_bf16 (gemm_planes.h) for grids that fill the chip, as a PING-PONG of two wave groups.
//
// What the phase probe and the PMC passes said about gemm_pl_bf16: a k-step of a workgroup is [barrier | LDS-DMA
// issue | fragment reads + MFMAs | A refill]; the two workgroups of a CU settle IN PHASE -- whenever both are in their MFMA phase they
// share the matrix pipes, both slow down and finish together, so a phase difference shrinks -- and a step costs (MFMA time of both) +
// (the non-MFMA time of one): 3 250 cycles for 2 x 1 024 cycles of MFMA issue per SIMD, the matrix pipe 63-65 % busy.  Priorities,
// earlier DMA issue, deeper operand rings and one 8-wave workgroup in lock-step (tools/probes/rejected/) all leave that sum alone.
// Here the alternation is explicit:
//   * ONE 8-wave workgroup per CU = two groups of four waves (waves w and w + 4 share a SIMD); group g owns the BM x 128 output tile
//     of rows m0 + g BM .. and has the wave layout, fragment addressing and accumulators of a 4-wave gemm_pl_bf16 workgroup;
//   * time is cut into phases by workgroup-wide s_barriers; in phase 2s group 0 runs the MFMAs of k-step s while group 1 issues
//     LDS-DMA, in phase 2s + 1 the roles swap: each SIMD always has exactly one wave in its MFMA phase and the other one's DMA issue,
//     barrier skew and waits hide behind it;
//   * weight stages (128 columns x 128 B = 16 KB per k-step, shared by both groups: half the weight DMA per MFMA) form a ring of three;
//     in its DMA phase of step s every wave issues its 2 of the 16 pieces of stage s + 2;
//   * each group's A tile (BM + halo rows of one 128-byte unit) is DOUBLE-buffered: the tile of the next unit arrives one piece per
//     wave and DMA phase during the taps of the current one, so no refill round trip is ever exposed (gemm_pl_bf16: once per unit);
//   * completion: LDS-DMA retires in issue order, so at the end of EVERY phase a wave waits for all its pieces except those of its
//     most recent DMA phase (exact count as the s_waitcnt immediate) and then meets the barrier: a piece has at least two full
//     phases to land before its first reader.
// Per-element arithmetic (order of units, taps and MFMAs per accumulator) is exactly gemm_pl_bf16's: results are bit-identical, the launcher may
// pick either kernel from the grid size alone.  Conv form only, ktaps >= 7 (the A pieces of a wave must fit the DMA phases of a unit), no split-K.
// LDS: 4 (BM + 16) 128 + 3 x 16 384 B = 155,648 B at BM = 192.
//
// REJECTED (round 3, measured on MI355X, profiles/r03_pingpong_conv_probe.txt): bit-identical to gemm_pl_bf16 on 55.8 M and 706 M outputs, but
// 35-45 % slower in the mx arithmetic and 10-15 % slower in split-bf16 at c3- and c4-sized grids, in every variant tried (fragment reads inside the
// MFMA phase; all reads before the first MFMA; SALU-only DMA addressing; fragments fetched in the preceding DMA phase = this file).  What the phase
// counters showed: (1) a wave that is alone on its matrix pipe pays every LDS round trip itself -- with the reads inside the MFMA phase that phase takes
// 1 050-1 360 cycles for 768 cycles of MFMAs; (2) with the reads moved into the DMA phase the MFMA phase shrinks to 680-960 cycles, but the DMA phase then
// takes 860-960: an LDS-DMA instruction costs its wave ~200 cycles when the four feeding waves of a CU issue together (the CU accepts ~19 B per cycle),
// 21.6 KB per step = 1 170 cycles of a 1 280-cycle step budget; (3) every phase adds 200-450 cycles of barrier skew between eight waves.  Two independent
// 4-wave workgroups overlap these costs statistically and come out ahead.  Kept for tools/probes/mx_conv_probe.hip; not part of libfs2_hip.so.
#pragma once
#include "gemm_planes.h"

namespace fs2 {

// One LDS-DMA instruction with the address as SGPR base + 32-bit per-lane byte offset (saddr form): a loop that advances only the base spends
// SALU, not VALU, instructions per piece.
__device__ __forceinline__ void dma16s(unsigned voff, const void* sbase, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_off) : "memory");
}

constexpr int kPpStages = 3;
template <int BM> constexpr size_t pp_lds_bytes() { return (size_t)4 * (BM + kMaxHalo) * 128 + (size_t)kPpStages * kB16BN * 128; }

// end of a phase: (wait_dma) every LDS-DMA piece of this wave except its newest `recent` ones has landed; its LDS reads have returned; then the workgroup meets
__device__ __forceinline__ void pp_phase_end(int recent, bool wait_dma) {
    if (!wait_dma) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (recent <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (recent == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if (recent == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef FS2_PP_TIMING      // tools/probes/mx_conv_probe.hip: cycles of wave 0 (group 0) and wave 4 (group 1) of workgroup (0, 0) per phase kind
__device__ long long g_pp_phase[8];      // 0/1: group 0 compute / its wait at the phase end; 2/3: group 0 DMA phase / wait; 4-7: group 1
#define FS2_PPT(i) { const long long t_ = __builtin_readcyclecounter(); tacc[i] += t_ - tprev; tprev = t_; }      // (kept in registers: a store inside the loop would count on vmcnt)
#else
#define FS2_PPT(i)
#endif

// ARITH / NSPLIT as in gemm_pl_bf16: ARITH 0 split-bf16 (NSPLIT 3) or plain bf16 (1), 1 fp16 images, 2 mx planes / mx weight image (NSPLIT ignored)
template <int NSPLIT, int BM, int ARITH>
__global__ __launch_bounds__(512, 1) void gemm_pp_conv(GemmArgs a) {
    constexpr bool F16 = ARITH == 1;
    constexpr int MT = BM / 32;               // 16-row MFMA tiles per wave (wave tile = BM/2 x 64)
    constexpr int AROWS = BM + kMaxHalo;
    extern __shared__ __attribute__((aligned(16))) char smem_pp[];
    char* As0 = smem_pp;                                       // [group][buffer][AROWS][128]
    char* Bs0 = smem_pp + 4 * AROWS * 128;                     // [stage][128][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: every LDS-DMA destination (M0) is SGPR arithmetic
    const int grp = wave >> 2, v = wave & 3;
    const int wm = v >> 1, wn = v & 1;
    const int n0 = blockIdx.x * kB16BN, m0 = (int)blockIdx.y * 2 * BM + grp * BM;
    if (a.Rp != nullptr && (int)blockIdx.y * 2 * BM >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int ktaps = a.ktaps;
    const int P = (ktaps - 1) >> 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int lp = rperm(lr);
    const __bf16* Wb = reinterpret_cast<const __bf16*>(a.W);
    const __bf16* Xp = reinterpret_cast<const __bf16*>(a.Xp);
    const int nchunks = a.Cpad / 32;
    const int niter = nchunks * ktaps;
    const int jrow = lane >> 3, jslot = lane & 7;        // this lane's (row, physical slot) inside a 1-KB DMA instruction

    // A (per group): wave v issues the pieces q = v + 4j (8 tile rows each), piece j in the DMA phase of tap j; swizzle constant as in gemm_pl_bf16
    const int a_instr = (BM + 2 * P + 7) >> 3;
    const int sA = jslot ^ (jrow >> 1) ^ ((v & 1) << 2);
    const int arow0 = m0 - P + v * 8 + jrow;
    const int xrc = a.xp_row_chunks ? a.xp_row_chunks : nchunks;
    const int cg = a.k_groups > 1 ? (n0 / (a.N / a.k_groups)) * nchunks : 0;
    const __bf16* a_src0 = Xp + (ptrdiff_t)arow0 * xrc * 64 + sA * 8 + cg * 64;      // dereferenced only when the row is in [0, R)
    const size_t a_qstride = (size_t)32 * xrc * 64;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)smem_pp);
    const unsigned ldsA = lds0 + grp * (2 * AROWS * 128) + v * 1024, ldsB = lds0 + 4 * AROWS * 128 + wave * 1024;
    // Interior tiles (every row of the tile + halo exists) take the SALU-only path: per-lane 32-bit offsets that never change + a scalar base
    const bool interior = m0 - P >= 0 && m0 + BM + P <= a.R;
    const unsigned a_voff = (unsigned)((((v * 8 + jrow) * xrc + cg) * 64 + sA * 8) * 2);
    const char* a_sbase = reinterpret_cast<const char*>(Xp) + (ptrdiff_t)(m0 - P) * xrc * 128;
    auto dma_A_piece = [&](int ch, int buf, int j) {      // returns the number of instructions issued (0 / 1)
        if (v + 4 * j >= a_instr) return 0;
        const unsigned dst = ldsA + buf * (AROWS * 128) + j * 4096;
        if (interior) {
            dma16s(a_voff, a_sbase + (size_t)ch * 128 + (size_t)j * 32 * xrc * 128, dst);
        } else {
            const int row = arow0 + 32 * j;
            const bool ok = row >= 0 && row < a.R;
            const __bf16* src = a_src0 + (size_t)ch * 64 + (size_t)j * a_qstride;
            const void* sp = ok ? static_cast<const void*>(src) : static_cast<const void*>(g_zero16);
            dma16(sp, dst);
        }
        return 1;
    };
    // B: wave w issues the pieces q = w, w + 8 of a stage: LDS rows 8q + jrow = 64u + 16 (w >> 1) + jB (u = 0, 1), i.e. n-tile w >> 1 of column
    // half u, tile row jB = 8 (w & 1) + jrow, which receives weight row 64u + 4 rperm_inv(jB) + (w >> 1) (the permuted order of gemm_pl_bf16)
    const int jB = (wave & 1) * 8 + jrow;
    const int sB = jslot ^ ((jB >> 1) & 7);
    const unsigned b_voff = (unsigned)((((4 * rperm_inv(jB) + (wave >> 1)) * niter) * 64 + sB * 8) * 2);
    const char* b_sbase = reinterpret_cast<const char*>(Wb) + (size_t)n0 * niter * 128;
    const size_t b_u = (size_t)64 * niter * 128;
    auto dma_B = [&](int it, int stage) {
        const unsigned dst = ldsB + stage * (kB16BN * 128);
        const char* src = b_sbase + (size_t)it * 128;
        dma16s(b_voff, src, dst);
        dma16s(b_voff, src + b_u, dst + 8192);
    };

    // prologue: the first A tile of this group, weight stages 0 .. 2
    for (int j = 0; v + 4 * j < a_instr; ++j) dma_A_piece(0, 0, j);
    dma_B(0, 0);
    if (niter > 1) dma_B(1, 1);
    if (niter > 2) dma_B(2, 2);
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
                const int row = m0 + wm * (BM / 2) + mt * 16 + rperm(lg * 4 + r);
                f32x4 x = bv;
                x += load4_or_zero(a.resid + (size_t)row * a.ldr + col, a.resid != nullptr && row < a.R && col < a.N);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt][r] = x[nt];
            }
    }
    pp_phase_end(0, true);
#ifdef FS2_PP_TIMING
    long long tacc[4] = {0, 0, 0, 0};
    long long tprev = __builtin_readcyclecounter();
#endif
    int it = 0;
    int recent = 0;               // LDS-DMA instructions this wave issued in its most recent DMA phase

    // The fragments of a k-step: the two 16-byte pieces (slot lg | slot 4 + lg) of MT A rows and 4 B rows -- 20 ds_read_b128 at MT = 6.  They are
    // requested in the wave's DMA phase BEFORE its MFMA phase (the partner is computing then), so an MFMA phase is MFMAs only: a wave that is alone on
    // its matrix pipe has nobody to fill the stall of a fragment read (measured: with the reads inside the MFMA phase that phase took 1 050-1 360 cycles
    // for 768 cycles of MFMAs, the burst of 4 x 20 KB alone occupies the LDS for 320).
    v8i_t fb[4], fa[MT];                  // [piece 0 | piece 1] of a row: already the operand tuple of the scaled MFMA
    auto fetch = [&](int step) {          // step = unit * ktaps + tap
        const int chunk = step / ktaps, tap = step - chunk * ktaps;
        const char* As = As0 + (grp * 2 + (chunk & 1)) * (AROWS * 128);
        const char* Bs = Bs0 + (step % kPpStages) * (kB16BN * 128);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = wn * 64 + nt * 16 + lp;
            fb[nt] = __builtin_shufflevector(*reinterpret_cast<const v4i_t*>(Bs + swz(n, lg)), *reinterpret_cast<const v4i_t*>(Bs + swz(n, 4 + lg)), 0, 1, 2, 3, 4, 5, 6, 7);
        }
        const int rb = wm * (BM / 2) + lp + tap;
        const char* ap0 = As + (rb << 7) + ((lg ^ ((rb >> 1) & 7)) << 4);
        const char* ap1 = As + (rb << 7) + (((4 + lg) ^ ((rb >> 1) & 7)) << 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            fa[mt] = __builtin_shufflevector(*reinterpret_cast<const v4i_t*>(ap0 + mt * 2048), *reinterpret_cast<const v4i_t*>(ap1 + mt * 2048), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto piece = [](const v8i_t x, auto half) {
        if constexpr (decltype(half)::value == 0) return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(x, x, 0, 1, 2, 3));
        else return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(x, x, 4, 5, 6, 7));
    };
    using H0 = std::integral_constant<int, 0>; using H1 = std::integral_constant<int, 1>;
    // the MFMA phase: exactly the MFMAs of gemm_pl_bf16's k-step, in its order
    auto compute = [&](auto kind_tag) {
        constexpr int KIND = decltype(kind_tag)::value;
        if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if constexpr (KIND == 2) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0, a.mx_scale, 0, a.mx_scale_b);
            } else if constexpr (KIND == 1) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<true>(piece(fa[mt], H0{}), piece(fb[nt], H0{}), acc[mt][nt]);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<true>(piece(fa[mt], H1{}), piece(fb[nt], H1{}), acc[mt][nt]);
            } else {
                if (NSPLIT >= 2) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(piece(fa[mt], H1{}), piece(fb[nt], H0{}), acc[mt][nt]);
                }
                if (NSPLIT == 3) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(piece(fa[mt], H0{}), piece(fb[nt], H1{}), acc[mt][nt]);
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(piece(fa[mt], H0{}), piece(fb[nt], H0{}), acc[mt][nt]);
            }
        }
        if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(0);
    };
    // the DMA phase that follows the MFMA phase of step `it` (unit `chunk`, tap `tap`): one piece of this group's A tile of the next unit (into the
    // buffer the group finished with at the last tap of the previous unit) and this wave's two pieces of weight stage it + 3 (the ring slot of step
    // it, whose fragments both groups have fetched by now)
    auto feed = [&](int chunk, int tap) {
        int n = 0;
        if (chunk + 1 < nchunks) n += dma_A_piece(chunk + 1, (chunk + 1) & 1, tap);
        if (it + 3 < niter) { dma_B(it + 3, it % kPpStages); n += 2; }
        recent = n;
    };
    // Both groups run the SAME loop body [MFMA phase of step it | DMA phase + fragment fetch of step it + 1], group 1 one phase behind group 0 (it
    // meets one barrier more before the loop, group 0 one more after it): phase 2 it = group 0 computes step it while group 1 feeds and fetches,
    // phase 2 it + 1 the other way round.  At the end of every EVEN phase (for group 0 its MFMA phases, for group 1 its DMA phases) a wave waits
    // for all its DMA pieces but those of its latest DMA phase: weight stage it + 3 (sent in phases 2 it + 1 and 2 it + 2) is complete at the end
    // of phase 2 it + 4, one phase before its first fetch; an A piece sent after tap j is complete before the fetch that follows tap j + 2.
    fetch(0);
    if (grp == 1) pp_phase_end(0, false);
    const bool even_after_mfma = grp == 0;
    auto k_loop = [&](auto kind_tag, const int cb, const int ce) {
        for (int chunk = cb; chunk < ce; ++chunk) {
            for (int tap = 0; tap < ktaps; ++tap, ++it) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                compute(kind_tag);
                FS2_PPT(0)
                pp_phase_end(recent, even_after_mfma);
                FS2_PPT(1)
                feed(chunk, tap);
                if (it + 1 < niter) fetch(it + 1);
                FS2_PPT(2)
                pp_phase_end(recent, !even_after_mfma);
                FS2_PPT(3)
            }
        }
    };
    if constexpr (ARITH == 2) {
        k_loop(std::integral_constant<int, 1>{}, 0, nchunks >> 1);
        k_loop(std::integral_constant<int, 2>{}, nchunks >> 1, nchunks);
    } else {
        k_loop(std::integral_constant<int, 0>{}, 0, nchunks);
    }
    if (grp == 0) pp_phase_end(0, false);
#ifdef FS2_PP_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && v == 0)
        for (int i = 0; i < 4; ++i) g_pp_phase[grp * 4 + i] = tacc[i];
#endif
    pl_epilogue<MT>(a, acc, m0 + wm * (BM / 2), col, lg);
}

}  // namespace fs2
