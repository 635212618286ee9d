// gemm_plr_bf16: EXPERIMENT (probe only, not part of the library) - the k-tap convolution tile kernel with the WEIGHT operand
// kept out of LDS.  Built and measured in round 2 to find out what the per-step barrier and LDS-DMA issue of gemm_pl_bf16 cost.
//
// In gemm_pl_bf16 (gemm_planes.h) both operands travel through LDS, and the phase probe (gemm_probe.hip, FFN conv of the
// decoder at c3: 256-row tiles, 108 k-steps per tile) shows what the weight side costs a wave per k-step of ~3 790 cycles:
// 630 waiting at the per-step barrier (the 16-KB weight tile of step it + 1 is written by all four waves), 490 issuing its four
// 1-KB LDS-DMA pieces, and a third of the step's ds_read_b128 traffic.  But a wave re-uses a weight fragment across its
// MT = BM/32 row tiles straight from REGISTERS: LDS only serves to share the tile between the two M-waves.  Here every wave
// loads its own eight fragments (4 n-tiles x hi/lo) of the NEXT step with plain 16-byte global loads from a fragment-major image
// (1 KB contiguous per instruction, L2-resident) into a second register set while the MFMAs of this step run:
//   * no per-step barrier, no per-step DMA issue; the workgroup meets once per 32-channel chunk, when the A tile changes;
//   * the freed LDS double-buffers the A tile (the DMA of chunk c + 1 runs under the 9 steps of chunk c);
//   * same MFMA sequence per accumulator as gemm_pl_bf16: results bit-identical (checked by the probe on every output).
// RESULT (MI355X, R = 30 208 / 131 072 rows, C = 384, N = 1024, k = 9, fp32 out): gemm_pl_bf16 465 / 2 138 us, this kernel
// 534 / 2 063 us; with all weight loads redirected to one L1-resident 32 KB: 554 / 2 028 us.  Removing ~23 % of a wave's
// stall time bought 3.5 % on a full machine and lost 15 % at the c3 grid (1.84 rounds: a lone workgroup on a CU gains nothing
// from a freed barrier).  The same probe reads the shader clock: 1 869 MHz effective during this kernel (DVFS, random
// operands) against the 2 400 MHz the 2.5 PFLOP/s roofline assumes - the MFMA pipe is busy 81 % of the real cycles, and
// what a denser instruction stream gains in cycles the power manager takes back in clock.  Not adopted.
// Weight image (frag_weight_image): [panel = n / 128][k-step][wn 2][n-tile 4][hi | lo][lane 64] x 16 B, where lane (lr, lg) of
// n-tile nt holds channel 128 panel + 64 wn + 4 lr + nt, k-group lg: the bytes gemm_pl_bf16's lane reads from its LDS image.
// The image pointer travels in GemmArgs::xp_scratch (unused by the kernels themselves).
#pragma once
#include <type_traits>
#include "gemm_planes.h"

namespace fs2 {

template <int BM> constexpr size_t plr_lds_bytes() { return 2 * (size_t)(BM + kMaxHalo) * 128; }

// Wb: [Npad][niter][hi 32 | lo 32] bf16 (repack_weight_bf16) -> fragment-major image of the same size (Npad % 128 == 0)
__global__ void frag_weight_image(const __bf16* __restrict__ Wb, int Npad, int niter, uint4* __restrict__ Wfrag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte vector
    const int64_t total = (int64_t)Npad * niter * 8;
    if (i >= total) return;
    const int lane = (int)(i & 63), hl = (int)(i >> 6) & 1, nt = (int)(i >> 7) & 3, wn = (int)(i >> 9) & 1;
    const int64_t pi = i >> 10;                                            // panel * niter + it
    const int it = (int)(pi % niter), panel = (int)(pi / niter);
    const int lr = lane & 15, lg = lane >> 4;
    const int n = panel * 128 + 64 * wn + 4 * lr + nt;
    Wfrag[i] = *reinterpret_cast<const uint4*>(Wb + ((size_t)n * niter + it) * 64 + hl * 32 + lg * 8);
}

template <int NSPLIT, int BM, bool F16 = false>
__global__ __launch_bounds__(256, 2) void gemm_plr_bf16(GemmArgs a) {
    constexpr int MT = BM / 32;
    constexpr int AROWS = BM + kMaxHalo;
    extern __shared__ __attribute__((aligned(16))) char smem_q[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * kB16BN, m0 = blockIdx.y * BM;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int ktaps = a.ktaps;
    const int P = (ktaps - 1) >> 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int lp = rperm(lr);
    const __bf16* Xp = reinterpret_cast<const __bf16*>(a.Xp);
    const int nchunks = a.Cpad / 32;
    const int niter = nchunks * ktaps;
    const int jrow = lane >> 3, jslot = lane & 7;

    // A tile: as gemm_pl_bf16 (1-KB LDS-DMA pieces of 8 rows x 128 B, XOR swizzle through the source address), two buffers
    const int a_instr = (BM + 2 * P + 7) >> 3;
    const int sA = jslot ^ (jrow >> 1) ^ ((wave & 1) << 2);
    const int arow0 = m0 - P + wave * 8 + jrow;
    const __bf16* a_src0 = Xp + (ptrdiff_t)arow0 * nchunks * 64 + sA * 8;
    const size_t a_qstride = (size_t)32 * nchunks * 64;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)smem_q);
    const unsigned ldsA = lds0 + wave * 1024;
    auto dma_A = [&](int ch, int buf) {
        unsigned dst = ldsA + buf * (AROWS * 128);
        const __bf16* src = a_src0 + (size_t)ch * 64;
        int row = arow0;
        for (int q = wave; q < a_instr; q += 4) {
            const bool ok = row >= 0 && row < a.R;
            const void* sp = ok ? static_cast<const void*>(src) : static_cast<const void*>(g_zero16);
            dma16(sp, dst);
            dst += 4096; src += a_qstride; row += 32;
        }
    };
    // B fragments of k-step `it`: 8 (NSPLIT == 3) or 4 vectors per lane, 1 KB contiguous per instruction
    const bf16x8_t* b_src0 = reinterpret_cast<const bf16x8_t*>(a.xp_scratch) + ((size_t)blockIdx.x * niter * 2 + wn) * 512 + lane;
    bf16x8_t bh[2][4], bl[2][4];
    auto load_B = [&](int it, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
#ifdef FS2_PLR_SAMEB      // experiment: every step re-reads the same 2 x 16 KB (L1-resident) - same kernel time, so B traffic is not the bound
        const bf16x8_t* src = b_src0 + (size_t)(it & 1) * 1024;
#else
        const bf16x8_t* src = b_src0 + (size_t)it * 1024;
#endif
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            bh[S][nt] = src[nt * 128];
            if (NSPLIT == 3) bl[S][nt] = src[nt * 128 + 64];
        }
    };

    const int it_end = niter;
    dma_A(0, 0);
    load_B(0, std::integral_constant<int, 0>{});
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
                f32x4 v = bv;
                if (a.resid && row < a.R && col < a.N) v += *reinterpret_cast<const f32x4*>(a.resid + (size_t)row * a.ldr + col);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt][r] = v[nt];
            }
    }
    int it = 0, chunk = 0, tap = 0, abuf = 0;
    auto step = [&](auto cur_tag) {
        constexpr int CUR = decltype(cur_tag)::value;
        if (tap == 0) {
            // A tile of this chunk (DMA'd a chunk ago) and this step's B fragments have landed; every wave is done with the
            // previous chunk's tile, whose buffer the next DMA overwrites
            dma_barrier();
            if (chunk + 1 < nchunks) dma_A(chunk + 1, abuf ^ 1);
        }
        // unconditional (the last step re-loads its own fragments): behind a branch hipcc's vmcnt bookkeeping must assume the loads
        // were skipped and waits for ALL of them before the first MFMA
        load_B(it + 1 < it_end ? it + 1 : it, std::integral_constant<int, CUR ^ 1>{});
        const char* As = smem_q + abuf * (AROWS * 128);
        if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int r = wm * (BM / 2) + mt * 16 + lp + tap;
            const bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(As + swz(r, lg));
            if (NSPLIT >= 2) {
                const bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(As + swz(r, 4 + lg));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(al, bh[CUR][nt], acc[mt][nt]);
            }
            if (NSPLIT == 3) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(ah, bl[CUR][nt], acc[mt][nt]);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(ah, bh[CUR][nt], acc[mt][nt]);
        }
        if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(0);
        ++it;
        if (++tap == ktaps) { tap = 0; ++chunk; abuf ^= 1; }
    };
    while (it < it_end) {
        step(std::integral_constant<int, 0>{});
        if (it < it_end) step(std::integral_constant<int, 1>{});
    }
    pl_epilogue<MT>(a, acc, m0 + wm * (BM / 2), col, lg);
}

}  // namespace fs2
