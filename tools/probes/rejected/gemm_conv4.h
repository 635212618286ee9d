// REJECTED (round 6, measured; kept for tools/probes/conv4_probe.hip -- not part of libfs2_hip.so).  Bit-identical to the shipped kernel at every tile height,
// 4 % faster at c3's row count (fewer, fuller rounds of workgroups) and 5-13 % SLOWER at steady state (c4 / c5-shard row counts): per k-step 2,560 cycles for
// 1,536 cycles of MFMA issue -- LDS-DMA issue 550, fragment reads 350, barrier 160 -- and 15 % of a workgroup's time in a prologue and an epilogue that
// nothing overlaps.  profiles/r06_conv4_probe.txt, DESIGN.md section 3.
//
// gemm_conv4_mx4: the decoder's FFN convolution (k = 9, 384 -> 1024 + ReLU; reference core/modules.py:237-248) in the mx4 arithmetic of
// gemm_planes.h (ARITH = 3) on the one-wavefront-per-SIMD structure of gemm_row4.h: same operands (mx4 planes + their per-block scale bytes, the mx4
// weight image + its scale image), same per-accumulator MFMA order, so the results are bit-identical to gemm_pl_bf16<1, 256, false, 3>.
//
// Why (round 6).  gemm_pl_bf16 runs two 4-wave workgroups per CU (tile 256 x 128, 128 accumulator registers per wave) and lets the two overlap each
// other's barrier waits, DMA issue and fragment reads statistically: the matrix pipe is busy ~0.57 of the time, and every experiment on that loop since
// round 3 was negative.  What that structure cannot change is the LDS-DMA volume per MFMA: a 256 x 128 tile fetches 16 KB of weights
// per 64 MFMAs of a wave, 12 B per CU and cycle at today's pace against the ~19 B per cycle a CU accepts -- at full MFMA rate it would need 18.  Here a
// workgroup is 4 waves stacked along M, wave tile 16 MT rows x 128 columns (8 n-tiles: 32 MT literal accumulator registers), workgroup tile
// BM = 64 MT rows x 128 columns: the same 16-KB weight stage serves 2 MT / 8 as many MFMAs (MT = 6: 1.5 x), the A tile (BM + 8 rows x 64 channels,
// re-used by the nine taps) is double-buffered, the weight stages sit in a ring of THREE so that a stage has a whole k-step to land, and the
// k-step is pipelined by hand as in gemm_row4_bf16:
//   * a k-step = (unit, tap) = 4 groups (pairs of n-tiles) of 4 MT MFMAs; the B fragments of pair p + 1 are requested before the MFMAs of pair p;
//   * the 4 weight pieces (+ the scale piece of a cross-unit stage: wave 0) of stage it + 2 are dealt out between the MFMAs of groups 0-2;
//   * in front of group 3: s_waitcnt vmcnt(pieces issued in this step) -- LDS-DMA lands in issue order, so everything older (stage it + 1, the A
//     pieces) is in LDS -- and the step's ONE barrier; behind it the fragments (and scale words) of step it + 1 are requested and up to two pieces of
//     the NEXT unit's A tile are issued, under the MFMAs of group 3.
// Units 0-5 of a row hold fp16 channels (KIND 1: two v_mfma_f32_16x16x32_f16 per fragment pair), units 6-8 are cross units (KIND 3: two
// v_mfma_scale_f32_16x16x128_f8f6f4 with cbsz = blgp = 4; scale bytes as in gemm_planes.h).  C = 384, k = 9, N a multiple of 128: the launcher checks.
#pragma once
#include "gemm_row4.h"

namespace fs2 {

typedef __attribute__((address_space(3))) const bf16x8_t lds_b128_t;
typedef int v2i_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const v2i_t lds_i2_t;

template <int MT> constexpr int conv4_arows() { return 64 * MT + 8; }
constexpr int kConv4BStage = 128 * 128;
template <int MT> constexpr size_t conv4_lds_bytes() { return 2 * (size_t)conv4_arows<MT>() * 128 + 3 * (size_t)kConv4BStage + (size_t)conv4_arows<MT>() * kMx4ScLd + 3 * 1024; }

// acc[T] += A.B on a cross unit: block-scaled e2m1 x e2m1, K = 128 (16 cycles); the operands are ONE 16-byte piece each (the assembler's 128-bit
// operand form of the fp4 formats).  BLK 0: scale byte 0 of sa, byte 2 JB of sb; BLK 1: byte 1 of sa, byte 2 JB + 1 of sb (op_sel | op_sel_hi << 1).
template <int T, int BLK, int JB>
__device__ __forceinline__ void acc_mfma_mx4(const v4i_t& fa, const v4i_t& fb, int sa, int sb) {
#define X(t, r0, r1, r2, r3) if constexpr (T == t) { \
        if constexpr (BLK == 0 && JB == 0) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 a[" #r0 ":" #r3 "], %0, %1, a[" #r0 ":" #r3 "], %2, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : : "v"(fa), "v"(fb), "v"(sa), "v"(sb) : "a" #r0, "a" #r1, "a" #r2, "a" #r3); \
        else if constexpr (BLK == 0) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 a[" #r0 ":" #r3 "], %0, %1, a[" #r0 ":" #r3 "], %2, %3 op_sel_hi:[0,1,0] cbsz:4 blgp:4" : : "v"(fa), "v"(fb), "v"(sa), "v"(sb) : "a" #r0, "a" #r1, "a" #r2, "a" #r3); \
        else if constexpr (JB == 0) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 a[" #r0 ":" #r3 "], %0, %1, a[" #r0 ":" #r3 "], %2, %3 op_sel:[1,1,0] op_sel_hi:[0,0,0] cbsz:4 blgp:4" : : "v"(fa), "v"(fb), "v"(sa), "v"(sb) : "a" #r0, "a" #r1, "a" #r2, "a" #r3); \
        else asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 a[" #r0 ":" #r3 "], %0, %1, a[" #r0 ":" #r3 "], %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,0] cbsz:4 blgp:4" : : "v"(fa), "v"(fb), "v"(sa), "v"(sb) : "a" #r0, "a" #r1, "a" #r2, "a" #r3); }
    FS2_ACC_TUPLES(X)
#undef X
}

// this wave's LDS-DMA except its newest N instructions has landed, its LDS reads have returned; then the workgroup meets
template <int N> __device__ __forceinline__ void conv4_wait_barrier() {
    static_assert(N == 0 || (N >= 3 && N <= 6), "pieces a wave issues in front of a step's barrier: 3 weight pieces + up to two A pieces + the scale piece");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Phase stamps for tools/probes/conv4_probe.hip (compiled only with -DFS2_CONV4_TIMING): wave 0 of workgroup (0, 0); [0] cycles in front of the barrier waits,
// summed over the k-steps (s_waitcnt + s_barrier), [1] the whole k-loop, [2] prologue (entry -> first fragments requested), [3] epilogue.
#ifndef FS2_CONV4_ABL      // ablations for tools/probes/conv4_probe.hip (wrong results, the same instruction stream otherwise): 1 no LDS-DMA inside the loop,
#define FS2_CONV4_ABL 0    // 2 no barrier waits, 4 no fragment reads inside the loop, 8 no MFMAs
#endif
#ifdef FS2_CONV4_TIMING
__device__ long long g_conv4_phase[8];
#define FS2_C4T(stmt) stmt
#else
#define FS2_C4T(stmt)
#endif

template <int MT>
__global__ __launch_bounds__(256, 1) void gemm_conv4_mx4(GemmArgs a) {
    constexpr int NT = 8, NP = 4, BM = 64 * MT, RW = 16 * MT, AR = conv4_arows<MT>(), NQ = AR / 8;
    constexpr int KT = 9, XU = 12, NMAIN = 6, NUNITS = 9, NITER = NUNITS * KT, IT_CROSS0 = NMAIN * KT;      // C = 384: 12 units per plane row, 6 + 3 walked
    constexpr int ASLOT = AR * 128;
    static_assert(MT * NT <= 64 && AR % 8 == 0 && NITER % 2 == 1 && IT_CROSS0 % 2 == 0, "accumulators a[0 : 32 MT); the step pairs below");
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    FS2_C4T(const long long t_entry = __builtin_readcyclecounter(); long long t_wait = 0;)
    const int n0 = blockIdx.x * 128, m0 = blockIdx.y * BM;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int lr = lane & 15, lg = lane >> 4;
    const int lp = rperm(lr);
    const int jrow = lane >> 3, jslot = lane & 7;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)smem_c);
    const unsigned ldsA = lds0, ldsB = lds0 + 2 * ASLOT, ldsSc = ldsB + 3 * kConv4BStage, ldsSb = ldsSc + AR * kMx4ScLd;

    // ---- LDS-DMA pieces (1 KB: 8 rows x 128 B; lane = (row jrow, physical slot jslot))
    // A: piece i of wave w is instruction q = w + 4 i < NQ: tile rows 8 q + jrow = plane rows m0 - 4 + 8 q + jrow (zeros outside [0, R)); q & 1 == w & 1, so
    // the logical slot this lane fetches is a constant.
    const int xrc = a.xp_row_chunks ? a.xp_row_chunks : XU;
    const size_t arow_bytes = (size_t)xrc * 128;
    const int sA = jslot ^ (jrow >> 1) ^ ((wave & 1) << 2);
    const int arow0 = m0 - 4 + wave * 8 + jrow;
    const char* a_lane = reinterpret_cast<const char*>(a.Xp) + (ptrdiff_t)arow0 * (ptrdiff_t)arow_bytes + sA * 16;      // dereferenced only when the row is in [0, R)
    auto piece_A = [&](int i, int chunk) __attribute__((always_inline)) {      // piece i of this wave for unit `chunk` -> A slot chunk & 1 (the caller checks wave + 4 i < NQ)
        const int row = arow0 + 32 * i;
        const char* src = a_lane + (size_t)i * 32 * arow_bytes + (size_t)chunk * 128;
        const void* sp = (row >= 0 && row < a.R) ? static_cast<const void*>(src) : static_cast<const void*>(g_zero16);
        dma16(sp, ldsA + (chunk & 1) * ASLOT + (wave + 4 * i) * 1024);
    };
    // B: piece u < 4 of wave w is instruction q = w + 4 u: LDS rows 8 q + jrow = n-tile T = q >> 1, tile row jB; that row holds weight row
    // n0 + 64 (T >> 2) + 4 rperm_inv(jB) + (T & 3) (four consecutive channels per lane: gemm_planes.h).  Offsets relative to the tile's first weight row.
    const int jB = (wave & 1) * 8 + jrow;
    const int sB = jslot ^ ((jB >> 1) & 7);
    unsigned b_off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int T = (wave >> 1) + 2 * u;
        b_off[u] = (unsigned)(64 * (T >> 2) + 4 * rperm_inv(jB) + (T & 3)) * (unsigned)(NITER * 128) + (unsigned)(sB * 16);
    }
    gchar_t* bbase0 = uniform_ptr(reinterpret_cast<const char*>(a.W) + (size_t)n0 * NITER * 128);      // (a.W: the mx4 weight image, as gemm_pl_bf16 receives it)
    auto piece_B = [&](auto u_tag, int it, int slot) __attribute__((always_inline)) {
        constexpr int u = decltype(u_tag)::value;
        dma16_so(bbase0 + (size_t)it * 128, b_off[u], ldsB + slot * kConv4BStage + (wave + 4 * u) * 1024);
    };
    // the weight block scales of a cross-unit stage: 1 KB per (N tile, cross unit, tap) (gemm_mx.h: mx4_scale_image_bytes), wave 0
    const unsigned char* wsb_lane = a.w_rowscale + (size_t)blockIdx.x * (XU / 4) * KT * 1024 + lane * 16;
    auto piece_S = [&](int it, int slot) __attribute__((always_inline)) { dma16(wsb_lane + (size_t)(it - IT_CROSS0) * 1024, ldsSb + slot * 1024); };

    // ---- prologue: unit 0's A tile, stages 0 and 1, the row-scale bytes of the A tile's rows (24 per row: gemm_row4.h EPI 4 writes 32)
#pragma unroll
    for (int i = 0; i < (NQ + 3) / 4; ++i)
        if (wave + 4 * i < NQ) piece_A(i, 0);
    for_seq_i<0, 4>([&](auto u_tag) __attribute__((always_inline)) { piece_B(u_tag, 0, 0); });
    for_seq_i<0, 4>([&](auto u_tag) __attribute__((always_inline)) { piece_B(u_tag, 1, 1); });
    {
        unsigned char* Sc = reinterpret_cast<unsigned char*>(smem_c) + 2 * ASLOT + 3 * kConv4BStage;
        constexpr int ncu = kMx4ScLd / 8;
        for (int t = tid; t < AR * ncu; t += 256) {
            const int rt = t / ncu, u = t - rt * ncu, row = m0 - 4 + rt;
            *reinterpret_cast<uint2*>(Sc + rt * kMx4ScLd + u * 8) =
                (row >= 0 && row < a.R) ? *reinterpret_cast<const uint2*>(a.x_rowscale + (size_t)row * 32 + u * 8) : uint2{0x7f7f7f7fu, 0x7f7f7f7fu};
        }
    }
    // this lane's rows: row(mt, r) = m0 + wave RW + 16 mt + rperm(4 lg + r); valid (inside R, not a gap row) as ONE bit mask, bit 4 mt + r
    int rp4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rp4[r] = rperm(lg * 4 + r);
    unsigned vmask = 0;
    {
        const int* __restrict__ rpos = a.row_pos;
        int flag[MT][4];                  // (all requests first: one round trip, not 4 MT of them)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wave * RW + mt * 16 + rp4[r];
                flag[mt][r] = loadi_or_zero(rpos + row, rpos != nullptr && row < a.R);
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wave * RW + mt * 16 + rp4[r];
                vmask |= (row < a.R && flag[mt][r] >= 0) ? (1u << (mt * 4 + r)) : 0u;
            }
    }
    // accumulators start at the bias: tuple (mt, 4 g + j) register r = tile row (mt, rperm(4 lg + r)), channel n0 + 64 g + 4 lr + j
    {
        f32x4 bv[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) bv[g] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + n0 + 64 * g + 4 * lr) : f32x4{0.f, 0.f, 0.f, 0.f};
        for_seq_i<0, MT * NT>([&](auto t_tag) __attribute__((always_inline)) {
            constexpr int T = decltype(t_tag)::value, n = T % NT;
            const float v = bv[n >> 2][n & 3];
            acc_set<T>(f32x4{v, v, v, v});
        });
    }

    // ---- fragments.  A: tile row wave RW + 16 mt + lp + tap, pieces slot lg | slot 4 + lg; the swizzle term does not depend on mt (rows 16 apart).
    struct AFrag { bf16x8_t h[MT], l[MT]; };
    struct BPair { bf16x8_t h[2], l[2]; };
    struct Scales { int sa[MT]; v2i_t sb[2]; };
    const int rb0 = wave * RW + lp;
    auto load_A = [&](AFrag& f, int chunk, int tap) __attribute__((always_inline)) {
        const int rb = rb0 + tap;
        const unsigned sw = (unsigned)(rb >> 1) & 7u;
        const unsigned base = ldsA + (unsigned)((chunk & 1) * ASLOT) + ((unsigned)rb << 7);
        const unsigned p0 = base + (((unsigned)lg ^ sw) << 4), p1 = base + (((unsigned)(4 + lg) ^ sw) << 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f.h[mt] = *reinterpret_cast<lds_b128_t*>(p0 + (unsigned)(mt * 2048));
            f.l[mt] = *reinterpret_cast<lds_b128_t*>(p1 + (unsigned)(mt * 2048));
        }
    };
    const unsigned bl0 = ldsB + ((unsigned)lp << 7) + (((unsigned)lg ^ ((unsigned)(lp >> 1) & 7u)) << 4);
    const unsigned bl1 = ldsB + ((unsigned)lp << 7) + (((unsigned)(4 + lg) ^ ((unsigned)(lp >> 1) & 7u)) << 4);
    auto load_B = [&](BPair& f, int slot, int n2) __attribute__((always_inline)) {
        const unsigned o = (unsigned)(slot * kConv4BStage + n2 * 2048);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f.h[u] = *reinterpret_cast<lds_b128_t*>(bl0 + o + (unsigned)(u * 2048));
            f.l[u] = *reinterpret_cast<lds_b128_t*>(bl1 + o + (unsigned)(u * 2048));
        }
    };
    // scale words of a cross-unit step: activations -- this lane's row of m-tile mt at this tap, bytes [slot lg | slot 4 + lg] of the cross unit; weights -- this
    // lane's four channels 64 g + 4 lr .. + 3 of the step's scale block, bytes [slot lg | slot 4 + lg] each
    const unsigned sc_lane = ldsSc + (unsigned)(rb0 * kMx4ScLd + lg * 2);
    const unsigned sb_lane = ldsSb + (unsigned)((lg * 128 + 4 * lr) * 2);
    auto load_S = [&](Scales& s, int chunk, int tap, int slot) __attribute__((always_inline)) {
        const unsigned p = sc_lane + (unsigned)(tap * kMx4ScLd + (chunk - NMAIN) * 8);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) s.sa[mt] = *reinterpret_cast<lds_u16_t*>(p + (unsigned)(mt * 16 * kMx4ScLd));
#pragma unroll
        for (int g = 0; g < 2; ++g) s.sb[g] = *reinterpret_cast<lds_i2_t*>(sb_lane + (unsigned)(slot * 1024 + g * 128));
    };

    // the MFMAs of one group (n-tiles n2, n2 + 1), flat index m: blk = piece pair (slot lg | slot 4 + lg), u, mt -- per accumulator: piece 0 then piece 1, as gemm_pl_bf16
    auto mfma_group = [&](auto kind_tag, const AFrag& fa, const BPair& fb, const Scales& sc, auto n2_tag, auto&& between) __attribute__((always_inline)) {
        constexpr int n2 = decltype(n2_tag)::value, KIND = decltype(kind_tag)::value;
        for_seq_i<0, 4 * MT>([&](auto m_tag) __attribute__((always_inline)) {
            constexpr int m = decltype(m_tag)::value;
            constexpr int blk = m / (2 * MT), u = (m % (2 * MT)) / MT, mt = m % MT;
            constexpr int n = n2 + u, T = mt * NT + n, g = n >> 2, j = n & 3;
            if constexpr ((FS2_CONV4_ABL & 8) != 0) {
            } else if constexpr (KIND == 3) {
                const int sbw = j < 2 ? sc.sb[g][0] : sc.sb[g][1];
                if constexpr (blk == 0) acc_mfma_mx4<T, 0, (j & 1)>(__builtin_bit_cast(v4i_t, fa.h[mt]), __builtin_bit_cast(v4i_t, fb.h[u]), sc.sa[mt], sbw);
                else acc_mfma_mx4<T, 1, (j & 1)>(__builtin_bit_cast(v4i_t, fa.l[mt]), __builtin_bit_cast(v4i_t, fb.l[u]), sc.sa[mt], sbw);
            } else {
                if constexpr (blk == 0) acc_mfma_f16<T>(fa.h[mt], fb.h[u]);
                else acc_mfma_f16<T>(fa.l[mt], fb.l[u]);
            }
            between(m_tag);
        });
    };

    // ---- first fragments
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    AFrag fa0, fa1;
    BPair fb0, fb1;
    Scales sc0, sc1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) sc0.sa[mt] = sc1.sa[mt] = 0;
    sc0.sb[0] = sc0.sb[1] = sc1.sb[0] = sc1.sb[1] = v2i_t{0, 0};
    load_A(fa0, 0, 0);
    load_B(fb0, 0, 0);

    // one k-step `it` = (chunk, tap), weight stage in ring slot it % 3: fragments in (fc, fb0 = first pair, sc_c); leaves those of step it + 1 in (fn, fb0, sc_n).
    // Compile time: KIND of this step, MORE1 / MORE2 = steps it + 1 / it + 2 exist, KN3 = step it + 1 is a cross-unit step.
    // Nothing but MFMAs may come in bursts: the four waves of a workgroup leave the barrier together, the LDS pipe and the vector-memory path are shared by the
    // four SIMDs, instruction issue is in order -- a burst of fragment reads (or pieces) of all four waves queues up and the matrix pipes idle behind it
    // (ablations of the first build, tools/probes/conv4_probe.hip: reads in bursts +520, pieces in one group +500, barrier +350 cycles on a 1,536-cycle step).
    // So every group of 4 MT MFMAs carries its share of the step's other work, one instruction at a time, evenly spaced:
    //   group p < 3: B pair p + 1 (4 reads), a third of step it + 1's A fragments (group 2: + its row-scale words), and group 3: B pair 0 and the weight-scale
    //                words of step it + 1, behind the barrier;
    //   every group p: LDS-DMA piece p of weight stage it + 2 into ring slot (it + 2) % 3 (free since the previous barrier); groups 0 / 2: the two A pieces of
    //                the step (tile of unit chunk + 1, pieces tap and tap + 7 of this wave, taps 0-6), group 1: the scale piece of stage it + 2 (wave 0);
    //   one barrier per step, in front of group 3, behind s_waitcnt vmcnt(pieces this wave issued in groups 0-2): LDS-DMA lands in issue order, so stage
    //                it + 1 is in LDS, and every wave holds stage `it` in registers.
    // Step it + 1's A fragments can be fetched in front of the barrier because its A tile landed a step ago (a piece has landed one barrier after the step of
    // its issue).  Through registers instead (global_load_dwordx4 in step t, ds_write_b128 in step t + 1) the pieces cost MORE: 710 cycles per step against
    // 550 (the writes wait for loads one step old), measured and dropped.
    constexpr int RAQ = (2 * MT + 2) / 3;                          // A-fragment reads per group
    int pend = 0;
    auto k_step = [&](auto kind_tag, AFrag& fc, AFrag& fn, Scales& sc_c, Scales& sc_n, int it, int chunk, int tap,
                      auto more1_tag, auto more2_tag, auto kn3_tag) __attribute__((always_inline)) {
        constexpr bool MORE1 = decltype(more1_tag)::value, MORE2 = decltype(more2_tag)::value, KN3 = decltype(kn3_tag)::value;
        constexpr bool DMA = !(FS2_CONV4_ABL & 1), RD = !(FS2_CONV4_ABL & 4);
        using KTAG = decltype(kind_tag);
        const int cur = it % 3, nxt = (it + 1) % 3, nn = (it + 2) % 3;
        const int tap1 = tap == KT - 1 ? 0 : tap + 1, chunk1 = tap == KT - 1 ? chunk + 1 : chunk;
        // addresses of step it + 1's fragments / scale words, of this step's B rows
        const int rbn = rb0 + tap1;
        const unsigned swn = (unsigned)(rbn >> 1) & 7u;
        const unsigned an = ldsA + (unsigned)((chunk1 & 1) * ASLOT) + ((unsigned)rbn << 7);
        const unsigned an0 = an + (((unsigned)lg ^ swn) << 4), an1 = an + (((unsigned)(4 + lg) ^ swn) << 4);
        const unsigned san = sc_lane + (unsigned)(tap1 * kMx4ScLd + (chunk1 - NMAIN) * 8);
        const unsigned bcur = (unsigned)(cur * kConv4BStage), bnxt = (unsigned)(nxt * kConv4BStage);
        const bool a_on = MORE1 && tap < KT - 2 && chunk + 1 < NUNITS;
        const bool a0 = a_on && wave + 4 * tap < NQ, a1 = a_on && wave + 4 * (tap + 7) < NQ;
        const bool s_on = MORE2 && it + 2 >= IT_CROSS0 && wave == 0;
        for_seq_i<0, NP>([&](auto p_tag) __attribute__((always_inline)) {
            constexpr int p = decltype(p_tag)::value;
            BPair& fthis = (p & 1) ? fb1 : fb0;
            BPair& fnext = (p & 1) ? fb0 : fb1;
            using N2 = std::integral_constant<int, 2 * p>;
            constexpr int ra_lo = p < 3 ? p * RAQ : 0, ra_hi = p < 3 ? (((p + 1) * RAQ < 2 * MT) ? (p + 1) * RAQ : 2 * MT) : 0;
            constexpr int n_ra = (MORE1 && ra_hi > ra_lo) ? ra_hi - ra_lo : 0;
            constexpr int n_rb = (p < 3 || MORE1) ? 4 : 0;
            constexpr int n_sa = (p == 2 && MORE1 && KN3) ? MT : 0, n_sb = (p == 3 && MORE1 && KN3) ? 2 : 0;
            constexpr int n_rd = n_rb + n_ra + n_sa + n_sb;
            constexpr int n_pc = (MORE2 ? 1 : 0) + (p < 3 && (p == 1 ? MORE2 : MORE1) ? 1 : 0);      // the weight piece, the group's extra piece
            constexpr int n_act = n_rd + n_pc, gm = 4 * MT;
            static_assert(n_act < gm, "one action per MFMA slot");
            // the list alternates: piece actions are spread between the reads (piece action c sits at list position pos(c))
            auto action = [&](auto k_tag) __attribute__((always_inline)) {
                constexpr int k = decltype(k_tag)::value;
                // piece action c (0 .. n_pc - 1) takes list position (c + 1) * n_act / (n_pc + 1); the reads fill the rest in order
                constexpr auto is_piece = [](int kk) constexpr { for (int c = 0; c < n_pc; ++c) if (((c + 1) * n_act) / (n_pc + 1) == kk) return c; return -1; };
                constexpr int c = n_pc > 0 ? is_piece(k) : -1;
                if constexpr (c >= 0) {
                    if constexpr (DMA) {
                        if constexpr (MORE2 && c == 0) piece_B(std::integral_constant<int, p>{}, it + 2, nn);
                        else if constexpr (p == 0) { if (a0) piece_A(tap, chunk + 1); }
                        else if constexpr (p == 2) { if (a1) piece_A(tap + 7, chunk + 1); }
                        else if constexpr (p == 1) { if (s_on) piece_S(it + 2, nn); }
                    }
                } else {
                    constexpr auto n_before = [](int kk) constexpr { int n = 0; for (int c2 = 0; c2 < n_pc; ++c2) if (((c2 + 1) * n_act) / (n_pc + 1) < kk) ++n; return n; };
                    constexpr int r = k - n_before(k);              // index among the reads
                    if constexpr (r < n_rb) {                                         // B fragment read
                        constexpr int u = r >> 1, half = r & 1;
                        if constexpr (RD) {
                            const unsigned o = (p < 3 ? bcur + (unsigned)((2 * (p + 1) + u) * 2048) : bnxt + (unsigned)(u * 2048));
                            if constexpr (half == 0) fnext.h[u] = *reinterpret_cast<lds_b128_t*>(bl0 + o);
                            else fnext.l[u] = *reinterpret_cast<lds_b128_t*>(bl1 + o);
                        }
                    } else if constexpr (r < n_rb + n_ra) {                           // A fragment read of step it + 1
                        constexpr int ai = ra_lo + (r - n_rb), mt = ai >> 1, half = ai & 1;
                        if constexpr (RD) {
                            if constexpr (half == 0) fn.h[mt] = *reinterpret_cast<lds_b128_t*>(an0 + (unsigned)(mt * 2048));
                            else fn.l[mt] = *reinterpret_cast<lds_b128_t*>(an1 + (unsigned)(mt * 2048));
                        }
                    } else if constexpr (r < n_rb + n_ra + n_sa) {                    // row-scale word of m-tile mt
                        constexpr int mt = r - n_rb - n_ra;
                        if constexpr (RD) sc_n.sa[mt] = *reinterpret_cast<lds_u16_t*>(san + (unsigned)(mt * 16 * kMx4ScLd));
                    } else {                                                          // weight-scale words of column block g
                        constexpr int g = r - n_rb - n_ra - n_sa;
                        if constexpr (RD) sc_n.sb[g] = *reinterpret_cast<lds_i2_t*>(sb_lane + (unsigned)(nxt * 1024 + g * 128));
                    }
                }
            };
            if constexpr (p == 3 && MORE1) {
                FS2_C4T(asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const long long tw0 = __builtin_readcyclecounter();)
                if (FS2_CONV4_ABL & 2) {}
                else if (pend == 0) conv4_wait_barrier<0>();
                else if (pend == 3) conv4_wait_barrier<3>();
                else if (pend == 4) conv4_wait_barrier<4>();
                else if (pend == 5) conv4_wait_barrier<5>();
                else conv4_wait_barrier<6>();
                FS2_C4T(t_wait += __builtin_readcyclecounter() - tw0;)
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(KTAG{}, fc, fthis, sc_c, N2{}, [&](auto m_tag) __attribute__((always_inline)) {
                constexpr int m = decltype(m_tag)::value;
                for_seq_i<0, n_act>([&](auto k_tag) __attribute__((always_inline)) {
                    constexpr int k = decltype(k_tag)::value;
                    constexpr int slot = ((k + 1) * gm) / (n_act + 1) - 1;
                    if constexpr (slot == m) {
                        __builtin_amdgcn_sched_barrier(0);
                        action(k_tag);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                });
            });
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (p == 2) pend = DMA ? (MORE2 ? 3 : 0) + (a0 ? 1 : 0) + (a1 ? 1 : 0) + (s_on ? 1 : 0) : 0;
        });
    };
    using K1 = std::integral_constant<int, 1>;
    using K3 = std::integral_constant<int, 3>;
    using TT = std::true_type;
    using FF = std::false_type;
    int it = 0, chunk = 0, tap = 0;
    FS2_C4T(const long long t_loop = __builtin_readcyclecounter();)
    auto advance = [&]() __attribute__((always_inline)) { ++it; if (++tap == KT) { tap = 0; ++chunk; } };
    // steps come in pairs (fa0 -> fa1 -> fa0); IT_CROSS0 is even, NITER odd
    for (; it < IT_CROSS0 - 2;) {
        k_step(K1{}, fa0, fa1, sc0, sc1, it, chunk, tap, TT{}, TT{}, FF{}); advance();
        k_step(K1{}, fa1, fa0, sc1, sc0, it, chunk, tap, TT{}, TT{}, FF{}); advance();
    }
    k_step(K1{}, fa0, fa1, sc0, sc1, it, chunk, tap, TT{}, TT{}, FF{}); advance();
    k_step(K1{}, fa1, fa0, sc1, sc0, it, chunk, tap, TT{}, TT{}, TT{}); advance();      // step 54 is a cross-unit step
    for (; it < NITER - 3;) {
        k_step(K3{}, fa0, fa1, sc0, sc1, it, chunk, tap, TT{}, TT{}, TT{}); advance();
        k_step(K3{}, fa1, fa0, sc1, sc0, it, chunk, tap, TT{}, TT{}, TT{}); advance();
    }
    k_step(K3{}, fa0, fa1, sc0, sc1, it, chunk, tap, TT{}, TT{}, TT{}); advance();      // it = NITER - 3: stage NITER - 1 is the last
    k_step(K3{}, fa1, fa0, sc1, sc0, it, chunk, tap, TT{}, FF{}, TT{}); advance();
    k_step(K3{}, fa0, fa1, sc0, sc1, it, chunk, tap, FF{}, FF{}, FF{});

    // ---- epilogue: ReLU -> mx planes of the hidden layer (fp16 | e4m3 residual | e4m3 copy: common.h store_planes4_mx), gap rows -> 0.  gemm_pl_bf16's
    // epilogue (pl_epilogue) fetches a row flag per accumulator row and waits for it -- 48 round trips that its second workgroup per CU hides and this
    // structure does not (measured: 87 k of a workgroup's 323 k cycles); here the flags were fetched in the prologue, as one bit mask per lane.
    acc_drain();
    FS2_C4T(const long long t_epi = __builtin_readcyclecounter();)
    {
        const int rowb = m0 + wave * RW;
        const float ysc = a.yp_scale;
        const int ych = a.yp_chunks;
        for_seq_i<0, MT>([&](auto mt_tag) __attribute__((always_inline)) {
            constexpr int mt = decltype(mt_tag)::value;
            for_seq_i<0, 2>([&](auto g_tag) __attribute__((always_inline)) {
                constexpr int g = decltype(g_tag)::value;
                f32x4 x[4];
                for_seq_i<0, 4>([&](auto j_tag) __attribute__((always_inline)) { x[decltype(j_tag)::value] = acc_get<mt * NT + 4 * g + decltype(j_tag)::value>(); });
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rowb + mt * 16 + rp4[r];
                    if (row >= a.R) continue;
                    const bool ok = (vmask >> (mt * 4 + r)) & 1u;
                    store_planes4_mx(a.Yp, (size_t)row, ych, n0 + 64 * g + 4 * lr,
                                     f32x4{ok ? fmaxf(x[0][r], 0.f) : 0.f, ok ? fmaxf(x[1][r], 0.f) : 0.f, ok ? fmaxf(x[2][r], 0.f) : 0.f, ok ? fmaxf(x[3][r], 0.f) : 0.f}, ysc);
                }
            });
        });
    }
    FS2_C4T(if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { g_conv4_phase[0] = t_wait; g_conv4_phase[1] = t_epi - t_loop; g_conv4_phase[2] = t_loop - t_entry; g_conv4_phase[3] = __builtin_readcyclecounter() - t_epi; })
}

}  // namespace fs2
