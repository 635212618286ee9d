// gemm_pl_bf16: the split-bf16 conv-as-GEMM with BOTH operands streamed by LDS-DMA and no conversion work in the loop.
//
// The first generation of these kernels read fp32 activations and split them into bf16 hi/lo inside the loop; PMC showed what
// that costs: 0.9 (9-tap conv, split once per chunk) to 8 (k = 1 GEMMs) VALU instructions per MFMA, i.e. the k = 1 GEMMs
// were VALU-bound at ~25 % of the MFMA rate.  Here the PRODUCER of an activation writes it once as "planes",
//     Xp[row][chunk] = 128 B = [hi: 32 bf16 in kperm order | lo: 32 bf16],      (common.h: plane_byte / store_planes4)
// which is byte-for-byte the LDS row image of the MFMA loop, so the A tile is a plain 1-KB-per-instruction DMA
// (8 rows x 128 B) exactly like the weight image.  Same HBM bytes as the fp32 tensor it replaces.
//   * tile BM x 128, 4 waves as 2(M) x 2(N), wave tile (BM/2) x 64; BM = 64 / 128 (3 or 2 workgroups per CU) or,
//     conv form only, 256 (2 per CU, 128 accumulator registers per lane): per MFMA a 256-row tile reads 25 % less
//     LDS and half the weight bytes of the 128-row one and meets half as many barriers.
//   * conv form (ktaps > 1): one A buffer (BM + halo rows) shared by the taps, refilled once per 32-channel chunk;
//     k = 1 form: A double-buffered like B.  One barrier per k-step.
//   * B rows are DMA'd in a permuted order: the LDS row that lane lr reads for n-tile nt holds output channel
//     64 wn + 4 lr + nt, so the four accumulators acc[mt][0..3][r] of a lane are FOUR CONSECUTIVE CHANNELS of one
//     row: the epilogue moves 16-byte pieces (float4 bias / residual / store, or 8 + 8 bytes of hi / lo planes)
//     instead of 4-byte ones, without any shuffle.
//   * A rows keep the conflict-free lane -> row permutation rperm (common.h).
#pragma once
#include <type_traits>
#include "gemm_bf16.h"

namespace fs2 {

template <int BM, bool K1> constexpr int pl_arows() { return K1 ? BM : BM + kMaxHalo; }
template <int BM, bool K1>
constexpr size_t pl_lds_bytes() {
    const size_t ops = (size_t)(K1 ? 2 : 1) * pl_arows<BM, K1>() * 128 + 2 * (size_t)kB16BN * 128;
    const size_t out = (K1 && BM <= 128) ? (size_t)BM * kQkvLd * 4 : 0;       // fused QKV epilogue stages the fp32 tile here
    return ops > out ? ops : out;
}
typedef __attribute__((address_space(3))) const unsigned short lds_u16_t;
constexpr int kMx4ScLd = 24;               // mx4 (ARITH = 3): scale bytes per row of the A tile in LDS (8 per cross unit; C = 384)
constexpr int kMx4RowScaleLds = 272 * kMx4ScLd;      // ... for the 272 rows of a 256-row tile, behind the operand buffers
template <int BM, bool K1> constexpr int pl_occ() { return pl_lds_bytes<BM, K1>() > 53 * 1024 ? 2 : 3; }

// fp32 [R, ldx] -> planes (for activations whose producer is not plane-aware); channels >= C are zero
__global__ void to_planes(const float* __restrict__ X, int ldx, int C, int R, int nchunks, void* __restrict__ Xp, int f16 = 0, float scale = 1.f) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = nchunks * 8;
    if (i >= (int64_t)R * per_row) return;
    const int row = (int)(i / per_row), c = (int)(i - (int64_t)row * per_row) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (c < C) v = *reinterpret_cast<const f32x4*>(X + (size_t)row * ldx + c);
    store_planes4m(Xp, row, nchunks, c, v, f16, scale);
}

// Elementwise epilogue on 4-channel pieces (bias and residual are already in the accumulators): ReLU, activation,
// gap rows -> 0; fp32 and / or planes out.
template <int MT>
__device__ __forceinline__ void pl_epilogue(const GemmArgs& a, f32x4 (&acc)[MT][4], int row_base, int col, int lg) {
    const int* __restrict__ rpos = a.row_pos;
    float* __restrict__ Y = a.Y;
    void* __restrict__ Yp = a.Yp;
    const bool colok = col < a.N;
    const bool pcol = Yp != nullptr && col < a.yp_chunks * 32;
    const bool relu_first = a.relu_pre != 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        bool inb[4], valid[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row_base + mt * 16 + rperm(lg * 4 + r);
            inb[r] = row < a.R;
            valid[r] = inb[r] && loadi_or_zero(rpos + row, rpos != nullptr && inb[r]) >= 0;      // (branch-free: see load4_or_zero)
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row_base + mt * 16 + rperm(lg * 4 + r);
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = acc[mt][j][r];
                if (relu_first) t = fmaxf(t, 0.f);
                t = apply_act(t, a.act_post);
                v[j] = (valid[r] && colok) ? t : 0.f;
            }
            if (Y && inb[r] && colok) *reinterpret_cast<f32x4*>(Y + (size_t)row * a.ldy + col) = v;
            if (pcol && inb[r]) store_planes4m(Yp, row, a.yp_chunks, col, v, a.yp_f16, a.yp_scale);
        }
    }
}

// Phase timing for tools/probes/gemm_probe.hip (compiled only with -DFS2_GEMM_TIMING): s_memtime ticks of wave 0 of workgroup
// (0, 0), accumulated per phase of the k-loop: 0 barrier wait, 1 DMA issue, 2 B fragment reads + MFMAs, 3 chunk-end A refill.
#ifdef FS2_GEMM_TIMING
__device__ long long g_gemm_phase[8];
#define FS2_GT(i) { const long long t_ = __builtin_readcyclecounter(); if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_gemm_phase[i] += t_ - tprev; tprev = t_; }
#else
#define FS2_GT(i)
#endif

// ARITH = 1: operands are fp16 images; NSPLIT = 3: lo*hi + hi*lo + hi*hi, 2: lo*hi + hi*hi (the weight's lo half is never read), 1: hi*hi.
// ARITH = 2 ("mx", conv form only, NSPLIT ignored): operands are mx planes / the mx weight image (gemm_mx.h).  A 128-byte unit is then
// either 64 fp16 channels (the first half of the units of a row: two fp16 MFMAs per fragment pair, a.w ~ ah.wh) or 128 e4m3 channels
// (the second half: ONE block-scaled K = 128 MFMA per fragment pair; first the units of ra8 against wh8, then those of ah8 against
// rw8 -- the two cross terms).  Everything else -- DMA pattern, swizzle, double buffering, barriers, the two 16-byte LDS reads per
// fragment -- is exactly the split-bf16 loop: per k-step 64 fp16 MFMAs or 32 scaled ones (both 1024 MFMA cycles) instead of 96.
// ARITH = 3 ("mx4", round 6; precision mode mix_mx4; conv form only): the same product with BOTH cross terms in fp4 (e2m1): 1.5 MFMA-equivalents per
// product instead of 2.0 (simulated first: tools/arith_sim_ffn_pareto.py, profiles/r06_ffn_arith_pareto.txt; timed first: tools/probes/make_fp4_probe.py).
// A row of C channels is C/64 units of fp16 channels, as above, followed by C/128 CROSS units: unit j holds, for the channels c = 128 j + 4 q .. + 3
// (q = 0 .. 31), the four bytes [ra4(c), ra4(c+1) | ra4(c+2), ra4(c+3) | ah4(c), ah4(c+1) | ah4(c+2), ah4(c+3)] with ra4 = e2m1((a - ah) 2^11 / s_row),
// ah4 = e2m1(ah / s_row); the weight image has [wh4 wh4 | wh4 wh4 | rw4 rw4 | rw4 rw4] with rw4 = e2m1((w - wh) 2^11 / s_n), wh4 = e2m1(wh / s_n) at the
// same byte positions, so that position by position the products are ra.wh 2^11 / (s_row s_n) and ah.rw 2^11 / (s_row s_n): ONE pair of E8M0 scale bytes per
// (row, 16-channel block of the weight row), s_row = 2^e from the row's own maximum (written by the producing LayerNorm epilogue, one byte per row, 2^-11 folded
// in), the weight side's from the block's own maximum (a lane's 16 bytes ARE the hardware's scale block: per-block scales are its native form) -- and since the
// second build the activation side's too: one scale per (row, 16-channel slot), 32 bytes per row.  e2m1 has two exponent bits: static per-tensor scales (what the e4m3 form uses) flush the residuals of small activations -- 4.4e-4
// on the mel with fp6, worse with fp4; the per-row scale costs the producer one more reduction pass and this loop one ds_read_u8 per m-tile and step.
// A cross unit issues TWO v_mfma_scale_f32_16x16x128_f8f6f4 with cbsz = blgp = 4 per fragment pair (16 cycles each; the operands are the two 16-byte
// pieces slot lg / slot 4 + lg of the row: 64 channels each, both terms); the scale of a lane's 32-value block comes from the lane's own register
// (tools/probes/fp4_probe.hip).  Per row 9 units are walked instead of 12 (C = 384): a quarter fewer MFMA cycles, LDS-DMA bytes, fragment reads and barriers.
// The last C/128 units of the row (e4m3 of the residual, as in the mx planes) are never read here: they serve the planes-only residual reader of
// gemm_row4_bf16 (RES = 2).
template <int NSPLIT, int BM, bool K1, int ARITH = 0>
__global__ __launch_bounds__(256, (pl_occ<BM, K1>())) void gemm_pl_bf16(GemmArgs a) {
    constexpr bool F16 = ARITH == 1;
    constexpr bool MX4 = ARITH == 3;
    static_assert((ARITH != 2 && ARITH != 3) || !K1, "the mx arithmetics exist for the conv form");
    constexpr int MT = BM / 32;               // 16-row MFMA tiles per wave (wave tile = BM/2 x 64)
    constexpr int AROWS = pl_arows<BM, K1>();
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    char* As0 = smem_p;
    char* Bs0 = smem_p + (K1 ? 2 : 1) * AROWS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    // wave index as a SCALAR: every LDS-DMA destination (M0) is then SGPR arithmetic instead of a v_readfirstlane per instruction
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // grid.x walks the N tiles: workgroup L of a launch runs on XCD L % 8, so with 8 N tiles each XCD keeps ONE 128-column weight
    // panel in its L2 and streams the A planes past it (measured: giving an XCD all N tiles of an M tile instead changes nothing)
    const int n0 = blockIdx.x * kB16BN, m0 = blockIdx.y * BM;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int ktaps = K1 ? 1 : a.ktaps;
    const int P = (ktaps - 1) >> 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int lp = rperm(lr);
    const __bf16* Wb = reinterpret_cast<const __bf16*>(a.W);
    const __bf16* Xp = reinterpret_cast<const __bf16*>(a.Xp);

    const int xunits = a.Cpad / 32;                      // 128-byte units per plane row
    const int nchunks = MX4 ? xunits - xunits / 4 : xunits;      // units the loop walks (mx4: C/64 of fp16 channels + C/128 cross units; the weight image holds exactly these)
    const int niter = nchunks * ktaps;
    const int jrow = lane >> 3, jslot = lane & 7;        // this lane's (row, physical slot) inside a 1-KB DMA instruction

    // A: wave w issues instructions q = w, w+4, ... (8 tile rows each).  The swizzle term ((r >> 1) & 7) of tile row
    // r = 8q + jrow is 4 (q & 1) + (jrow >> 1) and q & 1 == w & 1, so the logical slot this lane fetches is a constant.
    const int a_instr = K1 ? BM / 8 : (BM + 2 * P + 7) >> 3;
    const int sA = jslot ^ (jrow >> 1) ^ ((wave & 1) << 2);
    const int arow0 = m0 - P + wave * 8 + jrow;
    const int xrc = a.xp_row_chunks ? a.xp_row_chunks : xunits;      // chunks per plane row; grouped conv: this N tile's group starts at chunk cg
    const int cg = a.k_groups > 1 ? (n0 / (a.N / a.k_groups)) * nchunks : 0;
    const __bf16* a_src0 = Xp + (ptrdiff_t)arow0 * xrc * 64 + sA * 8 + cg * 64;      // dereferenced only when the row is in [0, R)
    const size_t a_qstride = (size_t)32 * xrc * 64;
    // LDS destinations of the DMA instructions as SCALARS (M0 = SGPR arithmetic; otherwise hipcc keeps the LDS pointer in a VGPR
    // and pays a v_readfirstlane + M0 hazard per instruction: the issue of the 4-6 instructions of a step took 13-30 % of it)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)smem_p);
    const unsigned ldsA = lds0 + wave * 1024, ldsB = lds0 + (K1 ? 2 : 1) * AROWS * 128 + wave * 1024;
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
    // B: instruction q = w + 4u fills LDS rows 8q + jrow = 64 (u >> 1) + 16 ((w >> 1) + 2 (u & 1)) + (8 (w & 1) + jrow),
    // i.e. n-tile nt = (w >> 1) + 2 (u & 1) of column half u >> 1, tile row jB; it receives weight row
    // 64 (u >> 1) + 4 rperm_inv(jB) + nt (see the header): one per-lane pointer + three uniform offsets.
    const int jB = (wave & 1) * 8 + jrow;
    const int sB = jslot ^ ((jB >> 1) & 7);
    const __bf16* b_src0 = Wb + ((size_t)(n0 + 4 * rperm_inv(jB) + (wave >> 1)) * niter) * 64 + sB * 8;
    const size_t b_o1 = (size_t)2 * niter * 64, b_o2 = (size_t)64 * niter * 64;
    auto dma_B = [&](int it, int buf) {
        const unsigned dst = ldsB + buf * (kB16BN * 128);
        const __bf16* src = b_src0 + (size_t)it * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            dma16(src + (u & 1) * b_o1 + (u >> 1) * b_o2, dst + u * 4096);
    };

    // split-K: this workgroup's share of the 32-channel chunks
    const int ks = (a.ksplit > 1) ? (int)blockIdx.z : 0;
    const int c_begin = (a.ksplit > 1) ? ks * (nchunks / a.ksplit) : 0;
    const int c_end = (a.ksplit > 1) ? c_begin + nchunks / a.ksplit : nchunks;
    const int it_end = c_end * ktaps;
    // mx4: the E8M0 scale bytes of the A tile's rows (32 per row: one per 16-channel slot of every cross unit, gemm_row4.h EPI 4) into LDS behind the operand
    // buffers; the first dma_barrier of the loop makes them visible.  The weight side has a scale per 16-byte slot of every weight row (gemm_mx.h: mx4_scale_image_bytes): 1 KB per
    // (N tile, cross unit, tap), fetched with the weight stage by ONE more LDS-DMA piece (wave 0) into a double buffer behind the row scales.
    unsigned char* Sc = reinterpret_cast<unsigned char*>(smem_p) + pl_lds_bytes<BM, K1>();
    constexpr int sc_ld = kMx4ScLd;                     // bytes per row of the A tile in LDS: 8 per cross unit, C = 384 (the launcher checks; the global records are 32 apart).
                                                        // A compile-time stride: the eight reads of a step are then one base address + immediates (as a run-time stride they cost
                                                        // registers the kernel does not have: 256 VGPRs + 28 bytes of scratch per lane, 25 % slower)
    const unsigned char* Sb = Sc + AROWS * sc_ld;
    const int it_cross0 = (xunits >> 1) * ktaps;      // first cross-unit step
    const unsigned char* wsb_tile = MX4 ? a.w_rowscale + (size_t)blockIdx.x * (xunits / 4) * ktaps * 1024 + lane * 16 : nullptr;
    const unsigned ldsSc = lds0 + (unsigned)pl_lds_bytes<BM, K1>();
    const unsigned sc_lane = ldsSc + (unsigned)((wm * (BM / 2) + lp) * sc_ld + lg * 2);       // this lane's row of m-tile 0, tap 0, cross unit 0
    const unsigned ldsSb = lds0 + (unsigned)pl_lds_bytes<BM, K1>() + (unsigned)(AROWS * sc_ld);
    auto dma_S = [&](int it_next) {      // the scale block of step it_next (a cross-unit step) into buffer it_next & 1
        if (MX4 && wave == 0 && it_next >= it_cross0) dma16(wsb_tile + (size_t)(it_next - it_cross0) * 1024, ldsSb + (it_next & 1) * 1024);
    };
    if constexpr (MX4) {
        static_assert(!MX4 || AROWS * kMx4ScLd <= kMx4RowScaleLds, "row-scale records of the A tile");
        constexpr int ncu = kMx4ScLd / 8;              // cross units per row: 8 scale bytes each
        for (int t = tid; t < AROWS * ncu; t += 256) {      // 8 bytes (one cross unit of one row) per thread and turn
            const int rt = t / ncu, u = t - rt * ncu, row = m0 - P + rt;
            *reinterpret_cast<uint2*>(Sc + rt * sc_ld + u * 8) =
                (row >= 0 && row < a.R) ? *reinterpret_cast<const uint2*>(a.x_rowscale + (size_t)row * 32 + u * 8) : uint2{0x7f7f7f7fu, 0x7f7f7f7fu};
        }
        dma_S(c_begin * ktaps);
    }
    dma_A(c_begin, K1 ? (c_begin & 1) : 0);
    dma_B(c_begin * ktaps, (c_begin * ktaps) & 1);
    // The accumulators start at bias + residual (loaded while the first tiles are in flight), so the epilogue has no
    // loads on its critical path: acc[mt][nt][r] belongs to row (mt, r), channel col + nt.
    const int col = n0 + wn * 64 + 4 * lr;    // this lane's four consecutive output channels
    f32x4 acc[MT][4];
    {
        f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias && col < a.N && ks == 0) bv = *reinterpret_cast<const f32x4*>(a.bias + col);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (BM / 2) + mt * 16 + rperm(lg * 4 + r);
                f32x4 v = bv;
                v += load4_or_zero(a.resid + (size_t)row * a.ldr + col, a.resid != nullptr && ks == 0 && row < a.R && col < a.N);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt][r] = v[nt];
            }
    }
    int it = c_begin * ktaps;
#ifdef FS2_GEMM_TIMING
    long long tprev = __builtin_readcyclecounter();
    const long long t_begin = tprev, r_begin = __builtin_amdgcn_s_memrealtime();      // shader cycles vs the constant 100-MHz counter
#endif
    // the k-loop over chunks [cb, ce) x taps; KIND (compile time) picks the MFMA body: 0 = split arithmetic per NSPLIT / ARITH, 1 = mx units
    // of fp16 channels, 2 = mx units of e4m3 channels, 3 = mx4 cross units (e2m1, both terms).  (Two instantiations for ARITH = 2 instead of a branch inside one loop: with both
    // bodies in one loop hipcc ran out of registers -- 108 spilled at BM = 256.)
    auto k_loop = [&](auto kind_tag, const int cb, const int ce) {
        constexpr int KIND = decltype(kind_tag)::value;
        constexpr bool kNeedB1 = KIND != 0 || NSPLIT == 3;      // second 16-byte piece of the B rows (lo half / second k-half)
        for (int chunk = cb; chunk < ce; ++chunk) {
            for (int tap = 0; tap < ktaps; ++tap, ++it) {
                dma_barrier();     // DMA of step `it` landed; every wave is done with step it-1
                FS2_GT(0)
#ifndef FS2_PROBE_DMA      // (ablations for tools/probes/mx_conv_probe.hip: 1 = weight stages only on even steps, 2 = none, 3 = no A refills)
#define FS2_PROBE_DMA 0
#endif
                if (it + 1 < it_end && !(FS2_PROBE_DMA == 1 && (it & 1)) && FS2_PROBE_DMA != 2) {
                    dma_B(it + 1, (it + 1) & 1);
                    if (K1) dma_A(it + 1, (it + 1) & 1);
                    if constexpr (MX4) dma_S(it + 1);
                }
                FS2_GT(1)
                const char* As = As0 + (K1 ? (it & 1) : 0) * (AROWS * 128);
                const char* Bs = Bs0 + (it & 1) * (kB16BN * 128);
                // a fragment = the two 16-byte pieces slot lg | slot 4 + lg of a row: split arithmetic: hi | lo of k 8 lg .. + 7; mx unit of fp16
                // channels: k 8 lg .. + 7 | 32 + 8 lg ..; mx unit of e4m3 channels: registers 0-3 | 4-7 of the scaled MFMA = k 16 lg .. + 15 |
                // 64 + 16 lg .. (measured in tools/probes/mx_probe.hip)
                bf16x8_t b0[4], b1[4];
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n = wn * 64 + nt * 16 + lp;
                    b0[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + swz(n, lg));
                    if (kNeedB1) b1[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + swz(n, 4 + lg));
                }
                // (KIND 3: this lane's row-scale words of the step, m-tile 0 -- as a 32-bit LDS address: through the generic pointer the compiler carries it in
                //  64 bits, three register pairs the cross loop does not have, and the reload of the spilled one waits on vmcnt(0), i.e. on the stage's LDS-DMA)
                const unsigned scp = sc_lane + (unsigned)__builtin_amdgcn_readfirstlane(tap * sc_ld + (chunk - (xunits >> 1)) * 8);
                int2 sbq = int2{0, 0};
                if constexpr (KIND == 3) sbq = *reinterpret_cast<const int2*>(Sb + (it & 1) * 1024 + (lg * 128 + wn * 64 + 4 * lr) * 2);
                if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(1);
                // A fragments: the swizzle term ((row >> 1) & 7) of row wm BM/2 + 16 mt + lp + tap does not depend on mt, so the two pieces of
                // m-tile mt sit at ONE per-step base address + mt * 2048 (an immediate offset of the LDS read) -- hipcc does not see this and
                // spent ~9 VALU instructions per m-tile and step on the addresses (PMC: 2.2 VALU per MFMA in the mx loop)
                const int rb = wm * (BM / 2) + lp + tap;
                const char* ap0 = As + (rb << 7) + ((lg ^ ((rb >> 1) & 7)) << 4);
                const char* ap1 = As + (rb << 7) + (((4 + lg) ^ ((rb >> 1) & 7)) << 4);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8_t a0 = *reinterpret_cast<const bf16x8_t*>(ap0 + mt * 2048);
                    if constexpr (KIND == 3) {
                        const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(ap1 + mt * 2048);
                        const int sa = *reinterpret_cast<lds_u16_t*>(scp + (unsigned)(mt * 16 * sc_ld));      // this lane's row of m-tile mt at this tap: byte 0 = slot lg, byte 1 = slot 4 + lg of this cross unit
                        // (sbq: this lane's eight weight-scale bytes of the step -- [n-tile 0: slot lg | slot 4 + lg][n-tile 1: ..] | [n-tile 2 ..][n-tile 3 ..] -- read once per step below)
                        const v4i_t z4 = v4i_t{0, 0, 0, 0};
                        const v8i_t av0 = __builtin_shufflevector(__builtin_bit_cast(v4i_t, a0), z4, 0, 1, 2, 3, 4, 5, 6, 7);
                        const v8i_t av1 = __builtin_shufflevector(__builtin_bit_cast(v4i_t, a1), z4, 0, 1, 2, 3, 4, 5, 6, 7);
#define FS2_MX4_PAIR(NT_)                                                                                                                              \
                        acc[mt][NT_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av0, __builtin_shufflevector(__builtin_bit_cast(v4i_t, b0[NT_]), z4, 0, 1, 2, 3, 4, 5, 6, 7), \
                                                                                       acc[mt][NT_], 4, 4, 0, sa, 2 * ((NT_) & 1), (NT_) < 2 ? sbq.x : sbq.y);
                        FS2_MX4_PAIR(0) FS2_MX4_PAIR(1) FS2_MX4_PAIR(2) FS2_MX4_PAIR(3)
#undef FS2_MX4_PAIR
#define FS2_MX4_PAIR(NT_)                                                                                                                              \
                        acc[mt][NT_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av1, __builtin_shufflevector(__builtin_bit_cast(v4i_t, b1[NT_]), z4, 0, 1, 2, 3, 4, 5, 6, 7), \
                                                                                       acc[mt][NT_], 4, 4, 1, sa, 2 * ((NT_) & 1) + 1, (NT_) < 2 ? sbq.x : sbq.y);
                        FS2_MX4_PAIR(0) FS2_MX4_PAIR(1) FS2_MX4_PAIR(2) FS2_MX4_PAIR(3)
#undef FS2_MX4_PAIR
                    } else if constexpr (KIND == 2) {
                        const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(ap1 + mt * 2048);
                        const v8i_t av = __builtin_shufflevector(__builtin_bit_cast(v4i_t, a0), __builtin_bit_cast(v4i_t, a1), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) {
                            const v8i_t bv = __builtin_shufflevector(__builtin_bit_cast(v4i_t, b0[nt]), __builtin_bit_cast(v4i_t, b1[nt]), 0, 1, 2, 3, 4, 5, 6, 7);
                            acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[mt][nt], 0, 0, 0, a.mx_scale, 0, a.mx_scale_b);
                        }
                    } else if constexpr (KIND == 1) {
                        const bf16x8_t a1 = *reinterpret_cast<const bf16x8_t*>(ap1 + mt * 2048);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<true>(a0, b0[nt], acc[mt][nt]);
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<true>(a1, b1[nt], acc[mt][nt]);
                    } else {
                        if (NSPLIT >= 2) {
                            const bf16x8_t al = *reinterpret_cast<const bf16x8_t*>(ap1 + mt * 2048);
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(al, b0[nt], acc[mt][nt]);
                        }
                        if (NSPLIT == 3) {
#pragma unroll
                            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(a0, b1[nt], acc[mt][nt]);
                        }
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<F16>(a0, b0[nt], acc[mt][nt]);
                    }
                }
                if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(0);
                FS2_GT(2)
                if (!K1 && tap == ktaps - 1 && chunk + 1 < c_end && FS2_PROBE_DMA != 3) {
                    __syncthreads();              // every wave has read its last fragments of this chunk's A tile
                    dma_A(chunk + 1, 0);
                    FS2_GT(3)
                }
            }
        }
    };
    if constexpr (ARITH == 3) {
        const int nmain = xunits >> 1;         // units of fp16 channels
        if (c_begin < nmain) k_loop(std::integral_constant<int, 1>{}, c_begin, c_end < nmain ? c_end : nmain);
        if (c_end > nmain) k_loop(std::integral_constant<int, 3>{}, c_begin > nmain ? c_begin : nmain, c_end);
    } else if constexpr (ARITH == 2) {
#ifndef FS2_MX_SKIP      // (tools/probes/mx_conv_probe.hip: 1 = no fp16 units, 2 = no e4m3 units)
#define FS2_MX_SKIP 0
#endif
        // (split-K: this workgroup's units [c_begin, c_end) may lie in either half)
        const int half = nchunks >> 1;
        if (FS2_MX_SKIP != 1 && c_begin < half) k_loop(std::integral_constant<int, 1>{}, c_begin, c_end < half ? c_end : half);
        if (FS2_MX_SKIP == 1) { it = half * ktaps; __syncthreads(); dma_A(half, 0); dma_B(it, it & 1); }
        if (FS2_MX_SKIP != 2 && c_end > half) k_loop(std::integral_constant<int, 2>{}, c_begin > half ? c_begin : half, c_end);
    } else {
        k_loop(std::integral_constant<int, 0>{}, c_begin, c_end);
    }
#ifdef FS2_GEMM_TIMING
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        g_gemm_phase[6] = __builtin_readcyclecounter() - t_begin;
        g_gemm_phase[7] = __builtin_amdgcn_s_memrealtime() - r_begin;
    }
#endif
    if constexpr (K1 && BM <= 128) {
        if (a.qk_hi != nullptr && n0 < 2 * a.att_D) {
            // fused QKV epilogue, Q | K tiles: row-major split-bf16 operands straight from the registers (8 + 8 bytes per
            // 4 channels, 128 contiguous bytes per row and plane); gap rows and rows beyond R are written as zeros
            const float sc = (n0 < a.att_D) ? a.q_scale : 1.f;
            const int* __restrict__ rpos = a.row_pos;
            __bf16* qkh = reinterpret_cast<__bf16*>(a.qk_hi);
            __bf16* qkl = reinterpret_cast<__bf16*>(a.qk_lo);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * (BM / 2) + mt * 16 + rperm(lg * 4 + r);
                    if (row >= a.Rvt) continue;
                    const bool ok = row < a.R && (rpos == nullptr || rpos[row] >= 0);
                    const float f = ok ? sc : 0.f;
                    uint2 hi, lo;
                    split4(f32x4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]} * f, hi, lo);
                    const size_t off = (size_t)row * 2 * a.att_D + col;
                    *reinterpret_cast<uint2*>(qkh + off) = hi;
                    *reinterpret_cast<uint2*>(qkl + off) = lo;
                }
            return;
        }
        if (a.qk_hi != nullptr) {             // V tile: (+bias) -> LDS -> V^T planes (8 consecutive keys per 16-byte store)
            float* tile = reinterpret_cast<float*>(smem_p);
            __syncthreads();                  // the operand buffers are dead: reuse them for the output tile
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                {
                    const int trow = wm * (BM / 2) + mt * 16 + rperm(lg * 4 + r);
                    *reinterpret_cast<f32x4*>(tile + trow * kQkvLd + ((wn * 64 + 4 * lr) ^ qkv_tile_swz(trow))) =
                        f32x4{acc[mt][0][r], acc[mt][1][r], acc[mt][2][r], acc[mt][3][r]};
                }
            __syncthreads();
            vt_tile_store<BM>(a, tile, m0, n0, tid);
            return;
        }
    }
    if (ks > 0) {                // split-K partial: raw sums into this split's buffer (same row stride as Y)
        GemmArgs p = a;
        p.Y = a.kpart + (size_t)(ks - 1) * a.kpart_stride;
        p.Yp = nullptr; p.relu_pre = 0; p.act_post = 0;
        pl_epilogue<MT>(p, acc, m0 + wm * (BM / 2), col, lg);
        return;
    }
    pl_epilogue<MT>(a, acc, m0 + wm * (BM / 2), col, lg);
}

}  // namespace fs2

namespace fs2 {

// ---------------------------------------------------------------------------------------------------------------
// gemm_row8_bf16: row-complete k = 1 GEMM (N = 128 NB = 256 or 384 outputs per row) with the LayerNorm epilogue fused.
//
// Measured on MI355X (FS2_PROBE experiments): an LDS-DMA round trip under load takes ~1.2 us, so a
// double-buffered k = 1 GEMM moves (bytes in flight per CU) / 1.2 us no matter how fast its MFMAs are, and the 64/128-row
// x 128-column tiles above spend 25-50 % of their time in the epilogue plus a second HBM-bound pass (ln_rows).  This
// kernel takes 128 rows x ALL N columns per workgroup: 2/3 of the operand bytes per flop of a 128 x 128 tile, the whole
// row in registers at the end, so bias + residual + LayerNorm + activation + positional encoding happen in the epilogue
// and the result leaves once, as fp32 and as planes (no fp32 round trip of the pre-LN tensor, no second launch).
//   * 8 waves as 4(M) x 2(N); wave tile 32 rows x 64 NB columns (MT = 2, NT = 4 NB); 16 NB accumulator registers x 2.
//   * one workgroup per CU (2 x (16 + 16 NB) KB of LDS), 2 waves per SIMD.
//   * B rows are DMA'd permuted as in gemm_pl_bf16: n-tile nt of lane lr is channel 64 NB wn + 64 (nt >> 2) + 4 lr + (nt & 3).
//   * LayerNorm statistics: two passes (mean, then centred sum of squares) over the 64 NB values a wave holds per row
//     (16-lane reduction), completed across the two N-waves through 2 KB of LDS.
// The MFMAs of one k-step of the 8-wave row-complete kernels (wave tile MT x NT 16x16 tiles; A fragments already in registers).
// B fragments are read one pair of n-tiles AHEAD of the MFMAs that use them (register double buffer, order pinned with
// sched_group_barrier): 3 LDS waits per k-step instead of 17.  Measured at c3: no change (dec.qkv 0.105 -> 0.102 ms, ffn2+LN
// 0.095 -> 0.095, out+LN 0.072 -> 0.069): with two waves per SIMD the LDS stalls were already hidden.  Ablations of
// gemm_row8_bf16 at c3 (232 workgroups in lockstep, N = 384): without any MFMA the kernel takes the same time; a k-step costs
// 1.3-1.6 us = the round trip of the ONE 64-KB stage that 160 KB of LDS lets a workgroup keep in flight (48 KB of it the
// weight slice every CU fetches from L2 at the same moment); the bias/residual/LayerNorm/store phase moves 137 MB in ~25 us.
template <int NSPLIT, int MT, int NT>
__device__ __forceinline__ void row8_mfma_step(const char* Bs, int nrow, int lg, const bf16x8_t (&ah)[MT], const bf16x8_t (&al)[MT], f32x4 (&acc)[MT][NT]) {
    constexpr int kReads = NSPLIT == 3 ? 4 : 2, kMfma = 2 * MT * NSPLIT;
    // LDS row nrow + 16 n of n-tile n: the swizzle term ((row >> 1) & 7) does not depend on n, so the hi / lo pieces of every n-tile sit at
    // one base address + n * 2048 (an immediate offset) -- spelled out because hipcc recomputed the swizzle per read (PMC: 3 VALU per MFMA)
    const char* bp0 = Bs + swz(nrow, lg);
    const char* bp1 = Bs + swz(nrow, 4 + lg);
    bf16x8_t bh[2][2], bl[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        bh[0][u] = *reinterpret_cast<const bf16x8_t*>(bp0 + u * 2048);
        if (NSPLIT == 3) bl[0][u] = *reinterpret_cast<const bf16x8_t*>(bp1 + u * 2048);
    }
    // hipcc's scheduler otherwise sinks every read to just before its first use, whatever the source order: pin the order
    // [first B reads] ([next B reads] [this group's MFMAs])*   (the A reads sit in the caller, before s_setprio)
    __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
#pragma unroll
    for (int n2 = 0; n2 < NT; n2 += 2) {
        const int cb = (n2 >> 1) & 1, nb = cb ^ 1;
        if (n2 + 2 < NT) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bh[nb][u] = *reinterpret_cast<const bf16x8_t*>(bp0 + (n2 + 2 + u) * 2048);
                if (NSPLIT == 3) bl[nb][u] = *reinterpret_cast<const bf16x8_t*>(bp1 + (n2 + 2 + u) * 2048);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kMfma, 0);
        if (NSPLIT == 3) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][n2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bh[cb][u], acc[mt][n2 + u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][n2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bl[cb][u], acc[mt][n2 + u], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt][n2 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mt], bh[cb][u], acc[mt][n2 + u], 0, 0, 0);
    }
}

// MT = 16-row m-tiles per wave: the workgroup takes 64 MT rows (128 / 192).  With one workgroup per CU a launch of T 128-row tiles
// takes ceil(T / 256) rounds -- 286 tiles (c3 at 7.87 frames per phoneme) ran as 256 + 30 and cost two rounds; the launcher picks the
// smallest tile height that keeps the number of rounds (191 tiles of 192 rows: one round).  Results do not depend on MT.
template <int NB, int MT = 2> constexpr size_t row8_lds_bytes() { return 2 * (size_t)(64 * MT + 128 * NB) * 128; }

template <int NSPLIT, int NB, int MT = 2>
__global__ __launch_bounds__(512, 1) void gemm_row8_bf16(GemmArgs a) {
    constexpr int NT = 4 * NB, BM = 64 * MT, BN = 128 * NB, RW = 16 * MT;      // RW: rows per wave
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave index as a SCALAR: every LDS-DMA destination (M0) is then SGPR arithmetic instead of a v_readfirstlane per instruction
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int lr = lane & 15, lg = lane >> 4;
    const int lp = rperm(lr);
    const __bf16* Wb = reinterpret_cast<const __bf16*>(a.W);
    const __bf16* Xp = reinterpret_cast<const __bf16*>(a.Xp);

    const int niter = a.Cpad / 32;
    const int jrow = lane >> 3, jslot = lane & 7;
    // A: 8 MT one-KB instructions per stage, wave w issues q = w + 8i, i < MT (tile rows 8q + jrow)
    const int sA = jslot ^ (jrow >> 1) ^ ((wave & 1) << 2);
    const int arow0 = m0 + wave * 8 + jrow;
    const __bf16* a_src0 = Xp + (size_t)arow0 * niter * 64 + sA * 8;
    const size_t a_qstride = (size_t)64 * niter * 64;
    // B: 16 NB instructions per stage, wave w issues q = w + 8u (u < 2 NB): LDS rows 8q + jrow = 16 ((w >> 1) + 4u) + jB,
    // i.e. 64-column group u, n-tile (w >> 1) of it, tile row jB -> weight row 64u + 4 rperm_inv(jB) + (w >> 1)
    const int jB = (wave & 1) * 8 + jrow;
    const int sB = jslot ^ ((jB >> 1) & 7);
    const __bf16* b_src0 = Wb + ((size_t)(4 * rperm_inv(jB) + (wave >> 1)) * niter) * 64 + sB * 8;
    const size_t b_ustride = (size_t)64 * niter * 64;
    auto dma_stage = [&](int it, int buf) {
        char* as = smem_r + buf * STAGE + wave * 1024;
        const __bf16* asrc = a_src0 + (size_t)it * 64;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bool ok = arow0 + 64 * i < a.R;
            const void* sp = ok ? static_cast<const void*>(asrc + i * a_qstride) : static_cast<const void*>(g_zero16);
            __builtin_amdgcn_global_load_lds(sp, (lds_void_t*)(as + i * 8192), 16, 0, 0);
        }
        char* bs = smem_r + buf * STAGE + BM * 128 + wave * 1024;
        const __bf16* bsrc = b_src0 + (size_t)it * 64;
#pragma unroll
        for (int u = 0; u < 2 * NB; ++u)
            __builtin_amdgcn_global_load_lds(bsrc + u * b_ustride, (lds_void_t*)(bs + u * 8192), 16, 0, 0);
    };

    dma_stage(0, 0);
    // accumulators start at bias + residual (loaded under the first DMA round trip): acc[mt][4g + j][r] is channel
    // col0 + 64 g + j of tile row (mt, r)
    const int col0 = wn * (64 * NB) + 4 * lr;
    // this lane's rows: row(mt, r) = rowb + 16 mt + rp[r].  (Arrays of row indices, positions, means and reciprocal deviations used to live
    // through the epilogue: 20 MT registers beside 16 MT NB accumulators -- the 192-row form spilled 158 of them and lost to the 128-row form
    // whenever it did not save a whole round of workgroups.)
    const int rowb = m0 + wm * RW;
    int rp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rp[r] = rperm(lg * 4 + r);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + col0 + 64 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rowb + mt * 16 + rp[r];
                f32x4 v = bv;
                v += load4_or_zero(a.resid + (size_t)row * a.ldr + col0 + 64 * g, a.resid != nullptr && row < a.R);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt][4 * g + j][r] = v[j];
            }
    }
    for (int it = 0; it < niter; ++it) {
        dma_barrier();
        if (it + 1 < niter) dma_stage(it + 1, (it + 1) & 1);
        const char* As = smem_r + (it & 1) * STAGE;
        const char* Bs = As + BM * 128;
        bf16x8_t ah[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {      // (one base address + mt * 2048: the swizzle term does not depend on mt)
            ah[mt] = *reinterpret_cast<const bf16x8_t*>(As + swz(wm * RW + lp, lg) + mt * 2048);
            if (NSPLIT == 3) al[mt] = *reinterpret_cast<const bf16x8_t*>(As + swz(wm * RW + lp, 4 + lg) + mt * 2048);
        }
        if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(1);
        row8_mfma_step<NSPLIT, MT, NT>(Bs, wn * (64 * NB) + lp, lg, ah, al, acc);
        if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(0);
    }

    // ---- epilogue: rows stay in registers.  LayerNorm statistics in two passes (mean, then centred sum of squares) over the 64 NB values a wave
    // holds per row, completed across the two N-waves through LDS; the accumulators are then normalised IN PLACE, so neither the means nor the
    // reciprocal deviations outlive that loop.
    const int* __restrict__ rpos = a.row_pos;
    float* __restrict__ Y = a.Y;
    void* __restrict__ Yp = a.Yp;
    const bool relu_first = a.relu_pre != 0;
    float* red = reinterpret_cast<float*>(smem_r);      // [2 passes][8 waves][RW rows]
    if (relu_first) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][n][r] = fmaxf(acc[mt][n][r], 0.f);
    }
    if (a.ln_g) {
        float rsum[MT][4], mean[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) s += acc[mt][n][r];
                rsum[mt][r] = wave16_sum(s);
            }
        __syncthreads();                                // operand buffers are dead
        if (lr == 0)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * RW + mt * 16 + lg * 4 + r] = rsum[mt][r];
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mean[mt][r] = (rsum[mt][r] + red[(wave ^ 1) * RW + mt * 16 + lg * 4 + r]) / (float)a.N;
                float q = 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) { const float d = acc[mt][n][r] - mean[mt][r]; q += d * d; }
                rsum[mt][r] = wave16_sum(q);
            }
        if (lr == 0)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[8 * RW + wave * RW + mt * 16 + lg * 4 + r] = rsum[mt][r];
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float rstd = 1.f / sqrtf((rsum[mt][r] + red[8 * RW + (wave ^ 1) * RW + mt * 16 + lg * 4 + r]) / (float)a.N + a.ln_eps);
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[mt][n][r] = (acc[mt][n][r] - mean[mt][r]) * rstd;
            }
    }
    const float alpha = (a.pe && a.pe_alpha) ? a.pe_alpha[0] : 1.f;
    // the two forms (with / without the positional-encoding add) are separate straight-line bodies: no load sits behind a
    // per-lane branch (see load4_or_zero), and the common form carries no positional-encoding loads at all
    auto emit = [&](auto pe_tag) {
        constexpr bool PE = decltype(pe_tag)::value;
        f32x4 gam[NB], bet[NB];
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            gam[g] = f32x4{1.f, 1.f, 1.f, 1.f}; bet[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.ln_g) { gam[g] = *reinterpret_cast<const f32x4*>(a.ln_g + col0 + 64 * g); bet[g] = *reinterpret_cast<const f32x4*>(a.ln_b + col0 + 64 * g); }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int pos[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {           // (the four loads go out together, as address selects: see load4_or_zero)
                const int row = rowb + mt * 16 + rp[r];
                const int pv = loadi_or_zero(rpos + row, rpos != nullptr && row < a.R);
                pos[r] = row < a.R ? pv : -1;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rowb + mt * 16 + rp[r];
                const bool live = pos[r] >= 0;
#pragma unroll
                for (int g = 0; g < NB; ++g) {
                    const int col = col0 + 64 * g;
                    f32x4 pe4 = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (PE) pe4 = load4_or_zero(a.pe + (size_t)(live ? pos[r] : 0) * a.pe_ld + col, live);
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = acc[mt][4 * g + j][r];
                        if (a.ln_g) t = t * gam[g][j] + bet[g][j];
                        t = apply_act(t, a.act_post);
                        if (PE) t = t * a.x_scale + alpha * pe4[j];
                        v[j] = live ? t : 0.f;
                    }
                    if (row < a.R) {
                        if (Y) *reinterpret_cast<f32x4*>(Y + (size_t)row * a.ldy + col) = v;
                        if (Yp) store_planes4m(Yp, row, a.yp_chunks, col, v, a.yp_f16, a.yp_scale);
                    }
                }
            }
        }
    };
    if (a.pe) emit(std::true_type{}); else emit(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------
// gemm_row8c_bf16: the row-complete structure for a k-tap convolution (N = 128 NB outputs per row) with ReLU -> LayerNorm ->
// activation (and the predictors' scalar head Linear(N, 1)) in the epilogue: the pitch / energy predictor layers at frame level
// (conv k = 3, 256 -> 256, reference variance_predictor.py:46-51).  They used to run as 128 x 128 conv tiles that wrote the pre-LN
// tensor, followed by an HBM-bound row pass (ln_rows): four 20-us passes plus four 30-MB scratch round trips per step at c3.
//   * 8 waves as 4(M) x 2(N), wave tile 32 rows x 64 NB columns, as gemm_row8_bf16; one workgroup per CU.
//   * A: one tile of 128 + halo rows per 32-channel chunk, shared by the taps (single buffer, refilled behind a barrier at the
//     end of a chunk, exactly as in gemm_pl_bf16's conv form); B: one 128 NB x 128 B tile per (chunk, tap), double-buffered.
template <int NB, int MT = 2> constexpr size_t row8c_lds_bytes() { return (size_t)(64 * MT + kMaxHalo) * 128 + 2 * (size_t)(128 * NB) * 128; }

// GROUPS = 2 (NB = 4: N = 512 = two stacked 256-channel layers over ONE input, the first layer of the pitch and the energy predictor): every
// N-wave holds one group's 256 columns, so ReLU / LayerNorm statistics stay inside the wave.  grid.y = a.k_groups > 1 (the second layer of
// the two predictors as one launch): workgroup (x, g) computes group g's N outputs from chunks [g Cpad/32, ..) of the plane rows, with the
// weights, bias, LayerNorm parameters and scalar head of group g (all stacked along N).
template <int NSPLIT, int NB, int MT = 2, int GROUPS = 1>
__global__ __launch_bounds__(512, 1) void gemm_row8c_bf16(GemmArgs a) {
    constexpr int NT = 4 * NB, BM = 64 * MT, BN = 128 * NB, RW = 16 * MT;
    static_assert(GROUPS == 1 || (GROUPS == 2 && NB == 4), "two groups = two N-waves of 256 columns");
    constexpr int AROWS = BM + kMaxHalo;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* As = smem_c;
    char* Bs0 = smem_c + AROWS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int ktaps = a.ktaps, P = (ktaps - 1) >> 1;
    const int lr = lane & 15, lg = lane >> 4;
    const int lp = rperm(lr);
    const int nchunks = a.Cpad / 32;
    const int niter = nchunks * ktaps;
    const int kg = a.k_groups > 1 ? (int)blockIdx.y : 0;             // input / parameter group of this workgroup
    const int xrc = a.xp_row_chunks ? a.xp_row_chunks : nchunks;
    const __bf16* Wb = reinterpret_cast<const __bf16*>(a.W) + (size_t)kg * BN * niter * 64;
    const __bf16* Xp = reinterpret_cast<const __bf16*>(a.Xp) + (size_t)kg * nchunks * 64;
    if (kg) { a.bias = a.bias ? a.bias + kg * BN : nullptr; a.ln_g = a.ln_g ? a.ln_g + kg * BN : nullptr; a.ln_b = a.ln_b ? a.ln_b + kg * BN : nullptr;
              a.dot_w = a.dot_w ? a.dot_w + kg * BN : nullptr; a.dot_b = a.dot_b ? a.dot_b + kg : nullptr; a.dot_out = a.dot_out ? a.dot_out + (size_t)kg * a.dot_gstride : nullptr; }
    const int jrow = lane >> 3, jslot = lane & 7;
    // A: instruction q = w, w + 8, w + 16 fills tile rows 8q + jrow (q & 1 == w & 1, so the swizzle term is a per-lane constant)
    const int a_instr = (BM + 2 * P + 7) >> 3;
    const int sA = jslot ^ (jrow >> 1) ^ ((wave & 1) << 2);
    const int arow0 = m0 - P + wave * 8 + jrow;
    const __bf16* a_src0 = Xp + (ptrdiff_t)arow0 * xrc * 64 + sA * 8;
    const size_t a_qstride = (size_t)64 * xrc * 64;
    auto dma_A = [&](int ch) {
        char* dst = As + wave * 1024;
        const __bf16* src = a_src0 + (size_t)ch * 64;
        int row = arow0;
        for (int q = wave; q < a_instr; q += 8) {
            const bool ok = row >= 0 && row < a.R;
            const void* sp = ok ? static_cast<const void*>(src) : static_cast<const void*>(g_zero16);
            __builtin_amdgcn_global_load_lds(sp, (lds_void_t*)dst, 16, 0, 0);
            dst += 8192; src += a_qstride; row += 64;
        }
    };
    // B: as gemm_row8_bf16 (weight row 64u + 4 rperm_inv(jB) + (w >> 1) -> LDS rows 16 ((w >> 1) + 4u) + jB), image step it = chunk * ktaps + tap
    const int jB = (wave & 1) * 8 + jrow;
    const int sB = jslot ^ ((jB >> 1) & 7);
    const __bf16* b_src0 = Wb + ((size_t)(4 * rperm_inv(jB) + (wave >> 1)) * niter) * 64 + sB * 8;
    const size_t b_ustride = (size_t)64 * niter * 64;
    auto dma_B = [&](int it, int buf) {
        char* bs = Bs0 + buf * (BN * 128) + wave * 1024;
        const __bf16* bsrc = b_src0 + (size_t)it * 64;
#pragma unroll
        for (int u = 0; u < 2 * NB; ++u)
            __builtin_amdgcn_global_load_lds(bsrc + u * b_ustride, (lds_void_t*)(bs + u * 8192), 16, 0, 0);
    };

    dma_A(0);
    dma_B(0, 0);
    const int col0 = wn * (64 * NB) + 4 * lr;
    const int rowb = m0 + wm * RW;        // this lane's rows: row(mt, r) = rowb + 16 mt + rp[r] (no arrays of row state: see gemm_row8_bf16)
    int rp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rp[r] = rperm(lg * 4 + r);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + col0 + 64 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rowb + mt * 16 + rp[r];
                f32x4 v = bv;
                v += load4_or_zero(a.resid + (size_t)row * a.ldr + col0 + 64 * g, a.resid != nullptr && row < a.R);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[mt][4 * g + j][r] = v[j];
            }
    }
    int it = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        for (int tap = 0; tap < ktaps; ++tap, ++it) {
            dma_barrier();                       // DMA of step `it` landed; every wave is done with step it - 1
            if (it + 1 < niter) dma_B(it + 1, (it + 1) & 1);
            const char* Bs = Bs0 + (it & 1) * (BN * 128);
            bf16x8_t ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {      // (one base address + mt * 2048: the swizzle term does not depend on mt)
                ah[mt] = *reinterpret_cast<const bf16x8_t*>(As + swz(wm * RW + lp + tap, lg) + mt * 2048);
                if (NSPLIT == 3) al[mt] = *reinterpret_cast<const bf16x8_t*>(As + swz(wm * RW + lp + tap, 4 + lg) + mt * 2048);
            }
            if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(1);
            row8_mfma_step<NSPLIT, MT, NT>(Bs, wn * (64 * NB) + lp, lg, ah, al, acc);
            if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(0);
            if (tap == ktaps - 1 && chunk + 1 < nchunks) {
                __syncthreads();                 // every wave has read its last fragments of this chunk's A tile
                dma_A(chunk + 1);
            }
        }
    }

    // ---- epilogue: rows stay in registers, normalised in place (as gemm_row8_bf16) + the scalar head
    const int* __restrict__ rpos = a.row_pos;
    float* __restrict__ Y = a.Y;
    void* __restrict__ Yp = a.Yp;
    const bool relu_first = a.relu_pre != 0;
    float* red = reinterpret_cast<float*>(smem_c);      // [3 passes][8 waves][RW rows]
    if (relu_first) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][n][r] = fmaxf(acc[mt][n][r], 0.f);
    }
    __syncthreads();                                    // operand buffers are dead
    const int n_ln = GROUPS == 2 ? a.N / 2 : a.N;      // columns one LayerNorm runs over
    if (a.ln_g) {
        float rsum[MT][4], mean[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s = 0.f;
#pragma unroll
                for (int n = 0; n < NT; ++n) s += acc[mt][n][r];
                rsum[mt][r] = wave16_sum(s);
            }
        if constexpr (GROUPS == 2) {      // the wave holds the whole group: no exchange
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mean[mt][r] = rsum[mt][r] / (float)n_ln;
                    float q = 0.f;
#pragma unroll
                    for (int n = 0; n < NT; ++n) { const float d = acc[mt][n][r] - mean[mt][r]; q += d * d; }
                    const float rstd = 1.f / sqrtf(wave16_sum(q) / (float)n_ln + a.ln_eps);
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[mt][n][r] = (acc[mt][n][r] - mean[mt][r]) * rstd;
                }
        } else {
            if (lr == 0)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave * RW + mt * 16 + lg * 4 + r] = rsum[mt][r];
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mean[mt][r] = (rsum[mt][r] + red[(wave ^ 1) * RW + mt * 16 + lg * 4 + r]) / (float)a.N;
                    float q = 0.f;
#pragma unroll
                    for (int n = 0; n < NT; ++n) { const float d = acc[mt][n][r] - mean[mt][r]; q += d * d; }
                    rsum[mt][r] = wave16_sum(q);
                }
            if (lr == 0)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[8 * RW + wave * RW + mt * 16 + lg * 4 + r] = rsum[mt][r];
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float rstd = 1.f / sqrtf((rsum[mt][r] + red[8 * RW + (wave ^ 1) * RW + mt * 16 + lg * 4 + r]) / (float)a.N + a.ln_eps);
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[mt][n][r] = (acc[mt][n][r] - mean[mt][r]) * rstd;
                }
        }
    }
    float dsum[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dsum[mt][r] = 0.f;
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        const int col = col0 + 64 * g;
        f32x4 gam = f32x4{1.f, 1.f, 1.f, 1.f}, bet = f32x4{0.f, 0.f, 0.f, 0.f}, dw = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.ln_g) { gam = *reinterpret_cast<const f32x4*>(a.ln_g + col); bet = *reinterpret_cast<const f32x4*>(a.ln_b + col); }
        if (a.dot_w) dw = *reinterpret_cast<const f32x4*>(a.dot_w + col);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int pos[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {           // (re-read per 64-column group: four L1-resident loads instead of 4 MT registers held throughout)
                const int row = rowb + mt * 16 + rp[r];
                const int pv = loadi_or_zero(rpos + row, rpos != nullptr && row < a.R);
                pos[r] = row < a.R ? pv : -1;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rowb + mt * 16 + rp[r];
                const bool live = pos[r] >= 0;      // (selects, not branches: see gemm_row8_bf16)
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = acc[mt][4 * g + j][r];
                    if (a.ln_g) t = t * gam[j] + bet[j];
                    t = apply_act(t, a.act_post);
                    v[j] = live ? t : 0.f;
                    dsum[mt][r] += live ? t * dw[j] : 0.f;
                }
                if (row < a.R) {
                    if (Y) *reinterpret_cast<f32x4*>(Y + (size_t)row * a.ldy + col) = v;
                    if (Yp) store_planes4m(Yp, row, a.yp_chunks, col + a.yp_col_off, v, a.yp_f16, a.yp_scale);
                }
            }
        }
    }
    if (a.dot_w) {      // scalar head: dot_out[row] = v . dot_w + dot_b, summed over the 16 lanes of a row group and the two N-waves
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dsum[mt][r] = wave16_sum(dsum[mt][r]);
        if (lr == 0)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[16 * RW + wave * RW + mt * 16 + lg * 4 + r] = dsum[mt][r];
        __syncthreads();
        if (wn == 0 && lr == 0) {
            const float db = a.dot_b ? a.dot_b[0] : 0.f;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rowb + mt * 16 + rp[r];
                    const bool live = row < a.R && loadi_or_zero(rpos + row, rpos != nullptr && row < a.R) >= 0;
                    if (row < a.R) a.dot_out[row] = live ? dsum[mt][r] + red[16 * RW + (wave ^ 1) * RW + mt * 16 + lg * 4 + r] + db : 0.f;
                }
        }
    }
}

// gemm_qkv8_bf16: the fused QKV projection on the row8 structure (8 waves, 128 rows x 128 NB columns per pass, one workgroup per
// CU).  N = 3 D with D = 128 NB: the workgroup makes three passes over its 128 rows - Q, K, V - re-streaming the A tile from L2;
// per MFMA it issues 1/2.25 of the LDS-DMA instructions and meets 1/3 of the barriers of the 64 x 128 tiles of gemm_pl_bf16, whose
// k-step was 31 % DMA issue + 30 % barrier (tools/probes/gemm_probe.hip).  Q (x log2e / sqrt(d_k)) and K leave straight from the
// registers as row-major split-bf16 planes [Rvt][2D]; V goes through LDS in three 128-column passes and leaves as V^T [D][Rvt]
// (8 consecutive keys per 16-byte store).  Rows that are gaps or beyond R are written as zeros.
template <int NB, int MT = 2> constexpr size_t qkv8_lds_bytes() {      // operand stages, or the [BM][132] fp32 tile of the V pass
    return row8_lds_bytes<NB, MT>() > (size_t)64 * MT * kQkvLd * 4 ? row8_lds_bytes<NB, MT>() : (size_t)64 * MT * kQkvLd * 4;
}

template <int NSPLIT, int NB, int MT = 2, bool APART = false>
__global__ __launch_bounds__(512, 1) void gemm_qkv8_bf16(GemmArgs a) {
    constexpr int NT = 4 * NB, BM = 64 * MT, BN = 128 * NB, RW = 16 * MT;
    constexpr int STAGE = (BM + BN) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave index as a SCALAR: every LDS-DMA destination (M0) is then SGPR arithmetic instead of a v_readfirstlane per instruction
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int lr = lane & 15, lg = lane >> 4;
    const int lp = rperm(lr);
    const __bf16* Wb = reinterpret_cast<const __bf16*>(a.W);
    const __bf16* Xp = reinterpret_cast<const __bf16*>(a.Xp);

    const int niter = a.Cpad / 32;
    const int jrow = lane >> 3, jslot = lane & 7;
    // A: 16 one-KB instructions per stage, wave w issues q = w and w + 8 (tile rows 8q + jrow)
    const int sA = jslot ^ (jrow >> 1) ^ ((wave & 1) << 2);
    const int arow0 = m0 + wave * 8 + jrow;
    const __bf16* a_src0 = Xp + (size_t)arow0 * niter * 64 + sA * 8;
    const size_t a_qstride = (size_t)64 * niter * 64;
    // B: 16 NB instructions per stage, wave w issues q = w + 8u (u < 2 NB): LDS rows 8q + jrow = 16 ((w >> 1) + 4u) + jB,
    // i.e. 64-column group u, n-tile (w >> 1) of it, tile row jB -> weight row 64u + 4 rperm_inv(jB) + (w >> 1)
    const int jB = (wave & 1) * 8 + jrow;
    const int sB = jslot ^ ((jB >> 1) & 7);
    const __bf16* b_src0 = Wb + ((size_t)(4 * rperm_inv(jB) + (wave >> 1)) * niter) * 64 + sB * 8;
    const size_t b_ustride = (size_t)64 * niter * 64;
    auto dma_stage = [&](int it, int buf) {
        char* as = smem_r + buf * STAGE + wave * 1024;
        const __bf16* asrc = a_src0 + (size_t)it * 64;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bool ok = arow0 + 64 * i < a.R;
            const void* sp = ok ? static_cast<const void*>(asrc + i * a_qstride) : static_cast<const void*>(g_zero16);
            __builtin_amdgcn_global_load_lds(sp, (lds_void_t*)(as + i * 8192), 16, 0, 0);
        }
        char* bs = smem_r + buf * STAGE + BM * 128 + wave * 1024;
        const __bf16* bsrc = b_src0 + (size_t)it * 64;
#pragma unroll
        for (int u = 0; u < 2 * NB; ++u)
            __builtin_amdgcn_global_load_lds(bsrc + u * b_ustride, (lds_void_t*)(bs + u * 8192), 16, 0, 0);
    };

    const int col0 = wn * (64 * NB) + 4 * lr;
    // this lane's rows: row(mt, r) = rowb + 16 mt + rp[r]; their validity as ONE bit mask (bit 4 mt + r) -- kept as arrays of row indices and
    // flags they cost 8 MT registers through all three k-loops (the 192-row form spilled 63 registers)
    const int rowb = m0 + wm * RW;
    int rp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rp[r] = rperm(lg * 4 + r);
    unsigned vmask = 0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = rowb + mt * 16 + rp[r];
            const bool ok = row < a.R && loadi_or_zero(a.row_pos + row, a.row_pos != nullptr && row < a.R) >= 0;
            vmask |= ok ? (1u << (mt * 4 + r)) : 0u;
        }
    __bf16* qkh = reinterpret_cast<__bf16*>(a.qk_hi);
    __bf16* qkl = reinterpret_cast<__bf16*>(a.qk_lo);
    // APART: this workgroup makes one pass (blockIdx.y) only -- a third of a tile as the unit of work (fs2_runtime.hip: qkv8_plan).
    // (A template parameter: with run-time loop bounds the whole-tile form ran 24 % slower.)
    const __bf16* b_src_blk = b_src0 + (APART ? (size_t)blockIdx.y * BN * niter * 64 : (size_t)0);
    for (int nbi = 0; nbi < (APART ? 1 : 3); ++nbi, b_src_blk += (size_t)BN * niter * 64) {
        const int nb = APART ? (int)blockIdx.y : nbi;
        __syncthreads();                 // every wave is done with the previous pass's operand buffers / LDS tile
        b_src0 = b_src_blk;
        dma_stage(0, 0);
        f32x4 acc[MT][NT];
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + nb * BN + col0 + 64 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[mt][4 * g + j][r] = bv[j];
        }
        for (int it = 0; it < niter; ++it) {
            dma_barrier();
            if (it + 1 < niter) dma_stage(it + 1, (it + 1) & 1);
            const char* As = smem_r + (it & 1) * STAGE;
            const char* Bs = As + BM * 128;
            bf16x8_t ah[MT], al[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {      // (one base address + mt * 2048: the swizzle term does not depend on mt)
                ah[mt] = *reinterpret_cast<const bf16x8_t*>(As + swz(wm * RW + lp, lg) + mt * 2048);
                if (NSPLIT == 3) al[mt] = *reinterpret_cast<const bf16x8_t*>(As + swz(wm * RW + lp, 4 + lg) + mt * 2048);
            }
            if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(1);
            row8_mfma_step<NSPLIT, MT, NT>(Bs, wn * (64 * NB) + lp, lg, ah, al, acc);
            if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(0);
        }

        if (nb < 2) {                    // Q | K: 8 + 8 bytes of hi / lo per 4 channels, 128 contiguous bytes per row and plane
            const float sc = (nb == 0) ? a.q_scale : 1.f;
#pragma unroll
            for (int g = 0; g < NB; ++g)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rowb + mt * 16 + rp[r];
                        if (row >= a.Rvt) continue;
                        const float f = ((vmask >> (mt * 4 + r)) & 1u) ? sc : 0.f;
                        uint2 hi, lo;
                        split4(f32x4{acc[mt][4 * g][r], acc[mt][4 * g + 1][r], acc[mt][4 * g + 2][r], acc[mt][4 * g + 3][r]} * f, hi, lo);
                        const size_t off = (size_t)row * 2 * BN + nb * BN + col0 + 64 * g;
                        *reinterpret_cast<uint2*>(qkh + off) = hi;
                        *reinterpret_cast<uint2*>(qkl + off) = lo;
                    }
        } else {                         // V: three 128-column passes through a [128][132] fp32 tile in LDS -> V^T planes
            float* tile = reinterpret_cast<float*>(smem_r);
            __bf16* vth = reinterpret_cast<__bf16*>(a.vt_hi);
            __bf16* vtl = reinterpret_cast<__bf16*>(a.vt_lo);
            for (int pass = 0; pass < NB; ++pass) {
                __syncthreads();         // operand buffers / previous pass's tile are dead
#pragma unroll
                for (int g = 0; g < NB; ++g) {
                    const int G = wn * NB + g;           // 64-column group of the block
                    if ((G >> 1) != pass) continue;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const f32x4 v = f32x4{acc[mt][4 * g][r], acc[mt][4 * g + 1][r], acc[mt][4 * g + 2][r], acc[mt][4 * g + 3][r]};
                            const int trow = wm * RW + mt * 16 + rp[r];
                            *reinterpret_cast<f32x4*>(tile + trow * kQkvLd + (((G & 1) * 64 + 4 * lr) ^ qkv_tile_swz(trow))) =
                                ((vmask >> (mt * 4 + r)) & 1u) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                }
                __syncthreads();
                // column c of the pass, rows 8j .. 8j+7: 4 consecutive lanes share a column (64-byte V^T segments)
#pragma unroll
                for (int u = 0; u < 2 * MT; ++u) {
                    const int idx = tid + u * 512;
                    const int c = (idx >> 2) & 127, j = ((idx >> 9) << 2) | (idx & 3);
                    const int row = m0 + 8 * j;
                    if (row >= a.Rvt) continue;
                    const float* t = tile + (8 * j) * kQkvLd + (c ^ qkv_tile_swz(8 * j));      // (rows 8j .. 8j + 7 share one swizzle term)
                    const SplitPair sp = split8(make_float4(t[0], t[kQkvLd], t[2 * kQkvLd], t[3 * kQkvLd]),
                                                make_float4(t[4 * kQkvLd], t[5 * kQkvLd], t[6 * kQkvLd], t[7 * kQkvLd]));
                    const size_t off = (size_t)(pass * 128 + c) * a.Rvt + row;
                    *reinterpret_cast<uint4*>(vth + off) = sp.hi;
                    *reinterpret_cast<uint4*>(vtl + off) = sp.lo;
                }
            }
        }
    }
}

}  // namespace fs2
