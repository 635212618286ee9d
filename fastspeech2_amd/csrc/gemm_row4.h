// gemm_row4_bf16: the row-complete k = 1 GEMM + LayerNorm of the decoder (out-proj + LN1, FFN2 + LN2, the decoder input layer;
// reference core/encoder.py:60-69,118-125, core/modules.py:248, core/attention.py:71-74) with ONE wavefront per SIMD and the whole
// 512-entry register file: 4 waves as 2(M) x 2(N), wave tile 16 MT rows x 64 NB columns, workgroup tile BM = 32 MT rows x all
// N = 128 NB columns.  Same operands, LDS images, lane -> row / column permutations and per-accumulator MFMA order as
// gemm_row8_bf16 (gemm_planes.h), so the results are bit-identical to it for every tile height.
//
// Why a second structure.  gemm_row8_bf16 holds 128 rows per workgroup in 8 waves x 96 accumulators; its
// 192-row form (144 accumulators, 256 registers: 42 spilled) was the only way to run c3's 36.6 k rows in one round of workgroups
// and cost 1.4-1.9 x a 128-row tile.  With 512 registers per wave a 160-row tile is 240 accumulators and nothing spills: c3 runs
// 229 workgroups of 160 rows (one round on 89 % of the CUs) instead of 191 of 192 rows on 75 %, the c5 shard 2 rounds instead
// of 3.  The launcher picks the tile height that minimises rounds x height (fs2_runtime.hip: row4_mt).
//
// The accumulators never appear as compiler values.  Left to hipcc (f32x4 acc[MT][NT] and the MFMA builtins under
// __launch_bounds__(256, 1)) the 192 / 240 accumulators were split between the two register files, copied with v_accvgpr_* around
// every group of MFMAs and spilled to scratch INSIDE the k-loop (127 - 821 registers spilled; the experience of attn_w32.h once more).
// Here accumulator tuple T = mt NT + n IS a[4 T : 4 T + 3], named literally in the MFMA / v_accvgpr statements below and listed
// as their clobbers; the compiler keeps only the fragments and addresses (< 256 architectural registers, no AGPR of its own).
// tools/probes/audit_rows.py checks the generated ISA after every edit: no compiler v_accvgpr_*, no scratch.
//
// A wave that is alone on its SIMD has nobody to cover its stalls, so the k-loop is software-pipelined by hand:
//   * a k-step = NT / 2 groups of 6 MT MFMAs (one pair of n-tiles each); the B fragments of pair p + 1 are requested before the
//     MFMAs of pair p (register double buffer);
//   * the step's ONE barrier sits in front of its LAST group: behind it (every wave has fetched all fragments of stage `it`, and
//     stage it + 1 has landed everywhere) the A fragments and the first B pair of step it + 1 are requested and the LDS-DMA of
//     stage it + 2 starts -- all of it under the last group's MFMAs, so the matrix pipe never drains at a step boundary;
//   * the 4 NB + MT one-KB LDS-DMA pieces a wave issues per stage are dealt out between MFMAs (a piece costs its wave ~60 cycles
//     of issue, which only the MFMAs already in the pipe cover): SCHED = number of groups of the NEXT step they are spread over beside
//     the last group of this one (0: one burst behind the barrier; 2: the default); a piece is `s_mov m0` + ONE VMEM instruction -- uniform
//     64-bit stage base in SGPRs + a per-lane 32-bit byte offset computed once (as address arithmetic per piece it was 5 VALU
//     instructions each, ~400 issue cycles of a 3,700-cycle step).
#pragma once
#include <type_traits>
#include "gemm_planes.h"

namespace fs2 {

// compile-time loop: f(integral_constant<int, LO>) ... f(integral_constant<int, HI - 1>)
template <int LO, int HI, class F>
__device__ __forceinline__ void for_seq_i(F&& f) {
    if constexpr (LO < HI) { f(std::integral_constant<int, LO>{}); for_seq_i<LO + 1, HI>(f); }
}

// accumulator tuple T = a[4 T : 4 T + 3] (X-macro: tuple index, its four registers)
#define FS2_ACC_TUPLES(X) \
    X(0, 0, 1, 2, 3) X(1, 4, 5, 6, 7) X(2, 8, 9, 10, 11) X(3, 12, 13, 14, 15) \
    X(4, 16, 17, 18, 19) X(5, 20, 21, 22, 23) X(6, 24, 25, 26, 27) X(7, 28, 29, 30, 31) \
    X(8, 32, 33, 34, 35) X(9, 36, 37, 38, 39) X(10, 40, 41, 42, 43) X(11, 44, 45, 46, 47) \
    X(12, 48, 49, 50, 51) X(13, 52, 53, 54, 55) X(14, 56, 57, 58, 59) X(15, 60, 61, 62, 63) \
    X(16, 64, 65, 66, 67) X(17, 68, 69, 70, 71) X(18, 72, 73, 74, 75) X(19, 76, 77, 78, 79) \
    X(20, 80, 81, 82, 83) X(21, 84, 85, 86, 87) X(22, 88, 89, 90, 91) X(23, 92, 93, 94, 95) \
    X(24, 96, 97, 98, 99) X(25, 100, 101, 102, 103) X(26, 104, 105, 106, 107) X(27, 108, 109, 110, 111) \
    X(28, 112, 113, 114, 115) X(29, 116, 117, 118, 119) X(30, 120, 121, 122, 123) X(31, 124, 125, 126, 127) \
    X(32, 128, 129, 130, 131) X(33, 132, 133, 134, 135) X(34, 136, 137, 138, 139) X(35, 140, 141, 142, 143) \
    X(36, 144, 145, 146, 147) X(37, 148, 149, 150, 151) X(38, 152, 153, 154, 155) X(39, 156, 157, 158, 159) \
    X(40, 160, 161, 162, 163) X(41, 164, 165, 166, 167) X(42, 168, 169, 170, 171) X(43, 172, 173, 174, 175) \
    X(44, 176, 177, 178, 179) X(45, 180, 181, 182, 183) X(46, 184, 185, 186, 187) X(47, 188, 189, 190, 191) \
    X(48, 192, 193, 194, 195) X(49, 196, 197, 198, 199) X(50, 200, 201, 202, 203) X(51, 204, 205, 206, 207) \
    X(52, 208, 209, 210, 211) X(53, 212, 213, 214, 215) X(54, 216, 217, 218, 219) X(55, 220, 221, 222, 223) \
    X(56, 224, 225, 226, 227) X(57, 228, 229, 230, 231) X(58, 232, 233, 234, 235) X(59, 236, 237, 238, 239) \
    X(60, 240, 241, 242, 243) X(61, 244, 245, 246, 247) X(62, 248, 249, 250, 251) X(63, 252, 253, 254, 255)

// acc[T] += A.B   (v_mfma_f32_16x16x32_bf16, fragments in architectural registers)
template <int T>
__device__ __forceinline__ void acc_mfma(const bf16x8_t& fa, const bf16x8_t& fb) {
#define X(t, r0, r1, r2, r3) if constexpr (T == t) asm volatile("v_mfma_f32_16x16x32_bf16 a[" #r0 ":" #r3 "], %0, %1, a[" #r0 ":" #r3 "]" : : "v"(fa), "v"(fb) : "a" #r0, "a" #r1, "a" #r2, "a" #r3);
    FS2_ACC_TUPLES(X)
#undef X
}
// the same on fp16 fragments (an mx unit of 64 fp16 channels: gemm_mx.h)
template <int T>
__device__ __forceinline__ void acc_mfma_f16(const bf16x8_t& fa, const bf16x8_t& fb) {
#define X(t, r0, r1, r2, r3) if constexpr (T == t) asm volatile("v_mfma_f32_16x16x32_f16 a[" #r0 ":" #r3 "], %0, %1, a[" #r0 ":" #r3 "]" : : "v"(fa), "v"(fb) : "a" #r0, "a" #r1, "a" #r2, "a" #r3);
    FS2_ACC_TUPLES(X)
#undef X
}
// acc[T] += A.B on an mx unit of 128 e4m3 channels: ONE block-scaled K = 128 MFMA (32 cycles); operands = both 16-byte pieces of the row, the
// E8M0 scale bytes (one constant per tensor, x 0x01010101) in architectural registers
template <int T>
__device__ __forceinline__ void acc_mfma_mx(const v8i_t& fa, const v8i_t& fb, int sa, int sb) {
#define X(t, r0, r1, r2, r3) if constexpr (T == t) asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 a[" #r0 ":" #r3 "], %0, %1, a[" #r0 ":" #r3 "], %2, %3 op_sel_hi:[0,0,0]" : : "v"(fa), "v"(fb), "v"(sa), "v"(sb) : "a" #r0, "a" #r1, "a" #r2, "a" #r3);
    FS2_ACC_TUPLES(X)
#undef X
}
// acc[T] = v
template <int T>
__device__ __forceinline__ void acc_set(const f32x4& v) {
#define X(t, r0, r1, r2, r3) if constexpr (T == t) asm volatile("v_accvgpr_write_b32 a" #r0 ", %0\n\tv_accvgpr_write_b32 a" #r1 ", %1\n\tv_accvgpr_write_b32 a" #r2 ", %2\n\tv_accvgpr_write_b32 a" #r3 ", %3" : : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]) : "a" #r0, "a" #r1, "a" #r2, "a" #r3);
    FS2_ACC_TUPLES(X)
#undef X
}
// v = acc[T]   (the caller has drained the matrix pipe: acc_drain)
template <int T>
__device__ __forceinline__ f32x4 acc_get() {
    float x0, x1, x2, x3;
#define X(t, r0, r1, r2, r3) if constexpr (T == t) asm volatile("v_accvgpr_read_b32 %0, a" #r0 "\n\tv_accvgpr_read_b32 %1, a" #r1 "\n\tv_accvgpr_read_b32 %2, a" #r2 "\n\tv_accvgpr_read_b32 %3, a" #r3 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3));
    FS2_ACC_TUPLES(X)
#undef X
    return f32x4{x0, x1, x2, x3};
}
__device__ __forceinline__ void acc_drain() { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory"); }      // 20 states: the last 8-pass MFMA's result is readable

template <int NB, int MT> constexpr size_t row4_lds_bytes() { return 2 * (size_t)(32 * MT + 128 * NB) * 128; }

// Phase stamps for tools/probes/row_probe.hip (compiled only with -DFS2_ROW_TIMING): shader cycles of wave 0 of a few workgroups at
// [0] entry, [1] first stage + residual landed, [2] k-loop done, [3] LayerNorm statistics done, [4] stores issued; [5] = s_memrealtime span.
#ifdef FS2_ROW_TIMING
__device__ long long g_row_phase[8][8];
#define FS2_RT(i) { if (threadIdx.x == 0 && (blockIdx.x & 31) == 0 && (blockIdx.x >> 5) < 8) g_row_phase[blockIdx.x >> 5][i] = __builtin_readcyclecounter(); }
#else
#define FS2_RT(i)
#endif

// EPI (compile time: the epilogue is unrolled over the accumulator tuples, and every launch-uniform choice left to run time would sit in that
// instruction stream 12 MT times -- the first build, with gemm_row8_bf16's run-time switches, was 400 KB of code): 0 = LayerNorm -> fp32 rows +
// split-bf16 planes (FFN2 + LN2; out-proj + LN1 in bf16x3 mode), 1 = LayerNorm -> fp32 rows + mx planes (out-proj + LN1 in mix_mx mode),
// 2 = LayerNorm -> ReLU -> x_scale v + alpha pe -> fp32 rows + split-bf16 planes (the decoder input layer).  Everything else stays on gemm_row8_bf16
// (fs2_runtime.hip: use_row4).  EPI = 3: one pass (blockIdx.y: Q, K or V) of the fused QKV projection -- no LayerNorm, no residual; Q (x log2e / sqrt(d_k)) and K leave
// as row-major split-bf16 planes [Rvt][2 D], V through a [BM][132] fp32 tile in LDS as V^T planes [D][Rvt] (8 consecutive keys per 16-byte store): gemm_qkv8_bf16's
// outputs, bit for bit, in 160-row tiles (687 pass-workgroups at c3 instead of 858 of 128 rows).
// ARITH = 2 (FFN2 + LN2 in mix_mx mode): the "mx" arithmetic of gemm_mx.h on this GEMM -- the A planes are mx planes (the hidden
// layer leaves FFN1's epilogue that way, with the static scale 2^kh of its a-priori bound), the weight image is the mx image of w_2: the first half of
// the 128-byte units are 64 fp16 channels (two fp16 MFMAs per fragment pair), the second half 128 e4m3 channels (ONE block-scaled MFMA): 32 matrix-pipe
// cycles per accumulator and unit instead of 48.  Same DMA pieces, ring, barriers and fragment reads; simulated first (tools/arith_sim_ffn2.py).
// RES (round 6, VERDICT r05 item 2: these launches are HBM-bound and half their algorithmic bytes were a second copy of the residual stream): where the
// residual comes from and what leaves.  0 = rounds 1-5: residual from fp32 rows (a.resid, or none), result as fp32 rows AND planes (kept for
// tools/probes/row_probe.hip, which holds it against gemm_row8_bf16 bit for bit; the library no longer instantiates it for EPI 0-2).  1 / 2 / 3 =
// PLANES ONLY: the result leaves as planes and nothing else (the next GEMM's A operand and the next LayerNorm-fused launch's residual are the same
// bytes), and the residual is what the producing launch's planes hold -- 1: split-bf16 planes (hi + lo, exact in fp32: 16-17 significant bits of the
// value the fp32 row held), 2: mx planes (fp16(x) + e4m3((x - fp16(x)) 2^(ka+11)) 2^-(ka+11): 12 of the row's 16 bytes per 4 channels are read,
// ~15 significant bits), 3: no residual (the decoder input layer).  Simulated first: tools/arith_sim_residual.py (mel +1.9e-5 / +2.8e-5 at c2).
template <int NSPLIT, int NB, int MT, int EPI = 0, int SCHED = 2, int ARITH = 0, int RES = 0>
__global__ __launch_bounds__(256, 1) void gemm_row4_bf16(GemmArgs a) {
    static_assert(ARITH == 0 || (ARITH == 2 && NSPLIT == 3), "split-bf16 (0) or mx (2) operands");
    static_assert(RES >= 0 && RES <= 3 && (EPI != 3 || RES == 0), "residual source / output form");
    static_assert(EPI != 4 || RES != 0, "EPI 4 (LayerNorm -> mx4 planes + one scale byte per row, the A operand of gemm_pl_bf16<.., ARITH = 3>) exists planes-only");
    constexpr int NT = 4 * NB, NP = NT / 2, BM = 32 * MT, BN = 128 * NB, RW = 16 * MT;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int PIECES = MT + 4 * NB;                       // one-KB LDS-DMA pieces per wave and stage: A pieces first (they come from HBM), then B
    // pieces [cut(0), cut(1)) are issued behind the barrier (last group of a step), [cut(g + 1), cut(g + 2)) in group g < SCHED of the next step
    static_assert(NP >= 3 && NP % 2 == 0 && SCHED >= 0 && SCHED <= NP - 2, "an even number of n-tile pairs per wave (N = 256 or 384)");
    auto cut = [](int i) constexpr { return (PIECES * i) / (SCHED + 1); };
    extern __shared__ __attribute__((aligned(16))) char smem_q[];
    FS2_RT(0)
    const int tid = threadIdx.x, lane = tid & 63;
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
    const int nbp = EPI == 3 ? (int)blockIdx.y : 0;           // QKV pass of this workgroup: its 128 NB weight rows / bias entries
    if constexpr (EPI == 3) {
        Wb += (size_t)nbp * BN * niter * 64;
        a.bias = a.bias ? a.bias + nbp * BN : nullptr;
        a.resid = nullptr;
    }

    // A: piece i of wave w fills tile rows 32 i + 8 w + jrow (piece index q = w + 4 i, q & 1 == w & 1: the swizzle term is a per-lane constant).
    // Rows beyond R (the last tile) repeat row R - 1: their results are never stored, and rows of a GEMM do not interact.
    const int sA = jslot ^ (jrow >> 1) ^ ((wave & 1) << 2);
    const int arow0 = m0 + wave * 8 + jrow;
    unsigned a_off[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_off[i] = (unsigned)(min(arow0 + 32 * i, max(a.R - 1, m0)) - m0) * (unsigned)(niter * 128) + (unsigned)(sA * 16);      // (m0 >= R: a QKV tile of padding rows)
    // B: piece u of wave w (q = w + 4 u) fills LDS rows 32 u + 8 w + jrow = n-tile 2 u + (w >> 1), tile row jB; that row belongs to
    // weight row 64 (u >> 1) + 4 rperm_inv(jB) + 2 (u & 1) + (w >> 1) (gemm_planes.h: four consecutive channels per lane)
    const int jB = (wave & 1) * 8 + jrow;
    const int sB = jslot ^ ((jB >> 1) & 7);
    unsigned b_off[4 * NB];
#pragma unroll
    for (int u = 0; u < 4 * NB; ++u) b_off[u] = (unsigned)(64 * (u >> 1) + 4 * rperm_inv(jB) + 2 * (u & 1) + (wave >> 1)) * (unsigned)(niter * 128) + (unsigned)(sB * 16);
    // (offsets relative to the TILE's first row, < 1 MB, so that nothing depends on how the SGPR-base addressing form extends its 32-bit VGPR offset:
    //  as offsets from the start of the planes they pass 2^31 on a 630 k-row launch with 4-KB rows -- c5 unsharded)
    gchar_t* abase0 = uniform_ptr(reinterpret_cast<const char*>(Xp) + (size_t)m0 * niter * 128);
    gchar_t* bbase0 = uniform_ptr(Wb);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)smem_q);
    const unsigned ldsw = lds0 + wave * 1024;
    // piece k (compile time) of stage `it` into ring slot `buf`: k-step `it` of either operand starts 128 it bytes into the rows
    auto piece = [&](auto k_tag, int it, int buf) __attribute__((always_inline)) {
        constexpr int k = decltype(k_tag)::value;
        if constexpr (k < MT) dma16_so(abase0 + (size_t)it * 128, a_off[k], ldsw + buf * STAGE + k * 4096);
        else dma16_so(bbase0 + (size_t)it * 128, b_off[k - MT], ldsw + buf * STAGE + BM * 128 + (k - MT) * 4096);
    };
    auto pieces = [&](auto lo_tag, auto hi_tag, int it, int buf) __attribute__((always_inline)) {
        constexpr int lo = decltype(lo_tag)::value, hi = decltype(hi_tag)::value;
        for_seq_i<lo, hi>([&](auto k_tag) __attribute__((always_inline)) { piece(k_tag, it, buf); });
    };
    using I0 = std::integral_constant<int, 0>;
    using IPL = std::integral_constant<int, cut(1)>;
    using IPA = std::integral_constant<int, PIECES>;

    pieces(I0{}, IPA{}, 0, 0);
    // accumulators start at bias + residual (loaded under the first DMA round trip): acc[mt][4 g + j][r] is channel col0 + 64 g + j of tile row (mt, r)
    const int col0 = wn * (64 * NB) + 4 * lr;
    const int rowb = m0 + wm * RW;
    int rp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rp[r] = rperm(lg * 4 + r);
    // (residual rows of m-tile mt + 1 are requested before those of m-tile mt go into the accumulators: one exposed round trip, 24 NB registers)
    static_assert(MT * NT <= 64, "the accumulators are a[0 : 4 MT NT)");
    // (planes as the residual: the raw words are held -- hi | lo of four channels (RES 1), fp16 x 4 | e4m3 x 4 (RES 2): no more registers than the
    //  fp32 form -- and converted where they go into the accumulators, so that the loads of m-tile mt + 1 are still in flight behind those of m-tile mt)
    f32x4 rv[2][NB][4];
    u32x4 rraw[2][NB][4];
    const char* rpl = reinterpret_cast<const char*>(a.residp);
    // RES 1: plane_byte(row, chunks, col0 + 64 g) = row chunks 128 + this lane's constant + 256 g;  RES 2: row 4 C + 2 (col0 + 64 g) [fp16] and + 2 C + col0 + 64 g [e4m3 residual]
    const size_t rrow_bytes = (size_t)a.residp_chunks * 128;
    const unsigned rlane = RES == 1 ? (unsigned)plane_byte(0, a.residp_chunks, col0) : (unsigned)(2 * col0);
    const size_t roff8 = a.residp_mx == 2 ? 3 * rrow_bytes / 4 : rrow_bytes / 2;      // the e4m3 residual words of the row: mx planes at 2 C, mx4 planes at 3 C
    auto load_resid = [&](auto mt_tag) __attribute__((always_inline)) {
        constexpr int mt = decltype(mt_tag)::value;
#pragma unroll
        for (int g = 0; g < NB; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rowb + mt * 16 + rp[r];
                if constexpr (RES == 0) {
                    rv[mt & 1][g][r] = load4_or_zero(a.resid + (size_t)row * a.ldr + col0 + 64 * g, a.resid != nullptr && row < a.R);
                } else if constexpr (RES == 1) {
                    const char* q = rpl + (size_t)row * rrow_bytes + rlane + 256 * g;
                    const uint2 hi = load8_or_zero(q, row < a.R), lo = load8_or_zero(q + 64, row < a.R);
                    rraw[mt & 1][g][r] = u32x4{hi.x, hi.y, lo.x, lo.y};
                } else if constexpr (RES == 2) {
                    const char* q = rpl + (size_t)row * rrow_bytes;
                    const uint2 hf = load8_or_zero(q + rlane + 128 * g, row < a.R);
                    const unsigned e8 = (unsigned)loadi_or_zero(reinterpret_cast<const int*>(q + roff8 + col0 + 64 * g), row < a.R);
                    rraw[mt & 1][g][r] = u32x4{hf.x, hf.y, e8, 0u};
                }
            }
    };
    const float rscale = a.residp_scale;
    // the four channels of one raw residual word group as fp32 (exact: see RES above)
    auto resid4 = [&](const u32x4& w) __attribute__((always_inline)) {
        if constexpr (RES == 1) {
            return f32x4{__uint_as_float(w[0] << 16) + __uint_as_float(w[2] << 16), __uint_as_float(w[0] & 0xffff0000u) + __uint_as_float(w[2] & 0xffff0000u),
                         __uint_as_float(w[1] << 16) + __uint_as_float(w[3] << 16), __uint_as_float(w[1] & 0xffff0000u) + __uint_as_float(w[3] & 0xffff0000u)};
        } else {
            const uint2 hw = uint2{w[0], w[1]};
            const f16x4_t hf = *reinterpret_cast<const f16x4_t*>(&hw);
            return f32x4{(float)hf[0] + __builtin_amdgcn_cvt_f32_fp8((int)w[2], 0) * rscale, (float)hf[1] + __builtin_amdgcn_cvt_f32_fp8((int)w[2], 1) * rscale,
                         (float)hf[2] + __builtin_amdgcn_cvt_f32_fp8((int)w[2], 2) * rscale, (float)hf[3] + __builtin_amdgcn_cvt_f32_fp8((int)w[2], 3) * rscale};
        }
    };
    f32x4 bv[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) bv[g] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + col0 + 64 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr bool HAS_RES = EPI != 3 && RES != 3;
    if constexpr (HAS_RES) load_resid(I0{});
    for_seq_i<0, MT>([&](auto mt_tag) __attribute__((always_inline)) {
        constexpr int mt = decltype(mt_tag)::value;
        if constexpr (mt + 1 < MT && HAS_RES) load_resid(std::integral_constant<int, mt + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        for_seq_i<0, NB>([&](auto g_tag) __attribute__((always_inline)) {
            constexpr int g = decltype(g_tag)::value;
            f32x4 v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (!HAS_RES) v[r] = bv[g];
                else if constexpr (RES == 0) v[r] = bv[g] + rv[mt & 1][g][r];
                else v[r] = bv[g] + resid4(rraw[mt & 1][g][r]);
            }
            for_seq_i<0, 4>([&](auto j_tag) __attribute__((always_inline)) {      // tuple (mt, 4 g + j): register r = row r, channel col0 + 64 g + j
                constexpr int j = decltype(j_tag)::value;
                acc_set<mt * NT + 4 * g + j>(f32x4{v[0][j], v[1][j], v[2][j], v[3][j]});
            });
        });
        __builtin_amdgcn_sched_barrier(0);
    });

    // fragment addresses: one base per operand half, everything else is an immediate (the swizzle term does not depend on the m- / n-tile)
    const char* ap0 = smem_q + swz(wm * RW + lp, lg);
    const char* ap1 = smem_q + swz(wm * RW + lp, 4 + lg);
    const char* bp0 = smem_q + BM * 128 + swz(wn * (64 * NB) + lp, lg);
    const char* bp1 = smem_q + BM * 128 + swz(wn * (64 * NB) + lp, 4 + lg);
    struct AFrag { bf16x8_t h[MT], l[MT]; };
    struct BPair { bf16x8_t h[2], l[2]; };
    auto load_A = [&](AFrag& f, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f.h[mt] = *reinterpret_cast<const bf16x8_t*>(ap0 + buf * STAGE + mt * 2048);
            if (NSPLIT == 3) f.l[mt] = *reinterpret_cast<const bf16x8_t*>(ap1 + buf * STAGE + mt * 2048);
        }
    };
    auto load_B = [&](BPair& f, int buf, int n2) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f.h[u] = *reinterpret_cast<const bf16x8_t*>(bp0 + buf * STAGE + (n2 + u) * 2048);
            if (NSPLIT == 3) f.l[u] = *reinterpret_cast<const bf16x8_t*>(bp1 + buf * STAGE + (n2 + u) * 2048);
        }
    };
    // the MFMAs of one group (n-tiles n2, n2 + 1), flat index m = 0 .. in gemm_row8_bf16's order (per accumulator: lo.hi, hi.lo, hi.hi); KIND 1 / 2: an mx
    // unit of fp16 / of e4m3 channels (per accumulator: piece 0 then piece 1 / one scaled MFMA).  After MFMA number m the callback `between(m)` may issue
    // something else (a DMA piece).  asm volatile: issued in program order.
    auto group_mfmas = [](int kind) constexpr { return kind == 0 ? (NSPLIT == 3 ? 6 : 2) * MT : (kind == 1 ? 4 * MT : 2 * MT); };
    const int mx_sa = a.mx_scale, mx_sb = a.mx_scale_b;
    auto mfma_group = [&](auto kind_tag, const AFrag& fa, const BPair& fb, auto n2_tag, auto&& between) __attribute__((always_inline)) {
        constexpr int n2 = decltype(n2_tag)::value, KIND = decltype(kind_tag)::value;
        for_seq_i<0, group_mfmas(KIND)>([&](auto m_tag) __attribute__((always_inline)) {
            constexpr int m = decltype(m_tag)::value;
            constexpr int blk = m / (2 * MT), u = (m % (2 * MT)) / MT, mt = m % MT;
            constexpr int T = mt * NT + n2 + u;
            if constexpr (KIND == 2) {
                acc_mfma_mx<T>(__builtin_shufflevector(__builtin_bit_cast(v4i_t, fa.h[mt]), __builtin_bit_cast(v4i_t, fa.l[mt]), 0, 1, 2, 3, 4, 5, 6, 7),
                               __builtin_shufflevector(__builtin_bit_cast(v4i_t, fb.h[u]), __builtin_bit_cast(v4i_t, fb.l[u]), 0, 1, 2, 3, 4, 5, 6, 7), mx_sa, mx_sb);
            } else if constexpr (KIND == 1) {
                if constexpr (blk == 0) acc_mfma_f16<T>(fa.h[mt], fb.h[u]);
                else acc_mfma_f16<T>(fa.l[mt], fb.l[u]);
            } else if constexpr (NSPLIT == 3) {
                if constexpr (blk == 0) acc_mfma<T>(fa.l[mt], fb.h[u]);
                else if constexpr (blk == 1) acc_mfma<T>(fa.h[mt], fb.l[u]);
                else acc_mfma<T>(fa.h[mt], fb.h[u]);
            } else {
                acc_mfma<T>(fa.h[mt], fb.h[u]);
            }
            between(m_tag);
        });
    };
    // deal the pieces [first, first + cnt) of stage `it` (ring slot buf) out evenly over a group's MFMAs: piece j goes behind MFMA number
    // floor((j + 1) gm / (cnt + 1)) - 1.  `between` callback of mfma_group.
    auto deal = [&](auto kind_tag, auto m_tag, auto first_tag, auto cnt_tag, int it, int buf) __attribute__((always_inline)) {
        constexpr int first = decltype(first_tag)::value, cnt = decltype(cnt_tag)::value, m = decltype(m_tag)::value;
        constexpr int gm = group_mfmas(decltype(kind_tag)::value);
        for_seq_i<0, cnt>([&](auto j_tag) __attribute__((always_inline)) {
            constexpr int j = decltype(j_tag)::value;
            constexpr int raw = ((j + 1) * gm) / (cnt + 1) - 1, slot = raw < 0 ? 0 : (raw > gm - 1 ? gm - 1 : raw);
            if constexpr (slot == m) {
                __builtin_amdgcn_sched_barrier(0);
                piece(std::integral_constant<int, first + j>{}, it, buf);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    auto nothing = [](auto) __attribute__((always_inline)) {};

    // ---- prologue of the pipeline: stage 0 landed; request the first fragments; start stage 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    FS2_RT(1)
    AFrag fa0, fa1;
    BPair fb0, fb1;
    load_A(fa0, 0);
    load_B(fb0, 0, 0);
    pieces(I0{}, IPL{}, 1, 1);

    // one k-step: fragments of step `it` in (fc: A, fb0: first B pair); leaves those of step it + 1 in (fn, fb0).  MORE1 / MORE2 (compile
    // time): stages it + 1 / it + 2 exist -- the loop's last two steps are instantiations of their own, so no DMA piece sits behind a branch.
    auto k_step = [&](auto kind_tag, AFrag& fc, AFrag& fn, int it, auto more1_tag, auto more2_tag) __attribute__((always_inline)) {
        constexpr bool MORE1 = decltype(more1_tag)::value, MORE2 = decltype(more2_tag)::value;
        using KT = decltype(kind_tag);
        const int cur = it & 1;
        for_seq_i<0, NP>([&](auto p_tag) __attribute__((always_inline)) {
            constexpr int p = decltype(p_tag)::value;
            BPair& fthis = (p & 1) ? fb1 : fb0;
            BPair& fnext = (p & 1) ? fb0 : fb1;
            using N2 = std::integral_constant<int, 2 * p>;
            if constexpr (p + 1 < NP) {
                load_B(fnext, cur, 2 * (p + 1));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (p < SCHED && MORE1 && (cut(p + 2) > cut(p + 1)))
                    mfma_group(KT{}, fc, fthis, N2{}, [&](auto m_tag) __attribute__((always_inline)) {
                        deal(KT{}, m_tag, std::integral_constant<int, cut(p + 1)>{}, std::integral_constant<int, cut(p + 2) - cut(p + 1)>{}, it + 1, cur ^ 1); });
                else mfma_group(KT{}, fc, fthis, N2{}, nothing);
            } else {
                // every fragment of stage `it` is in registers (the last pair was requested a group ago); stage it + 1 must have landed everywhere
                if constexpr (MORE1) {
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    load_A(fn, cur ^ 1);
                    load_B(fnext, cur ^ 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MORE2) mfma_group(KT{}, fc, fthis, N2{}, [&](auto m_tag) __attribute__((always_inline)) { deal(KT{}, m_tag, I0{}, IPL{}, it + 2, cur); });
                else mfma_group(KT{}, fc, fthis, N2{}, nothing);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using KL = std::integral_constant<int, ARITH == 2 ? 2 : 0>;      // the kind of the last steps
    int it = 0;
    if constexpr (ARITH == 2) {
        // (niter is a multiple of 4, >= 4: the launcher checks; units [0, niter / 2) hold fp16 channels, the rest e4m3 channels)
        for (; it < niter / 2; it += 2) {
            k_step(K1{}, fa0, fa1, it, std::true_type{}, std::true_type{});
            k_step(K1{}, fa1, fa0, it + 1, std::true_type{}, std::true_type{});
        }
    }
    // (niter is even and >= 2: the launcher checks)
    for (; it + 2 < niter; it += 2) {
        k_step(KL{}, fa0, fa1, it, std::true_type{}, std::true_type{});
        k_step(KL{}, fa1, fa0, it + 1, std::true_type{}, std::true_type{});
    }
    k_step(KL{}, fa0, fa1, it, std::true_type{}, std::false_type{});
    k_step(KL{}, fa1, fa0, it + 1, std::false_type{}, std::false_type{});
    FS2_RT(2)

    // ---- epilogue (gemm_row8_bf16's arithmetic): the rows stay in the accumulator file and are read three times -- row sums, centred sums of
    // squares (completed across the two N-waves through LDS), then normalise + affine (+ ReLU + positional encoding) on the way out
    acc_drain();
    const int* __restrict__ rpos = a.row_pos;
    if constexpr (EPI == 3) {
        // this lane's rows: row(mt, r) = rowb + 16 mt + rp[r]; their validity as ONE bit mask (bit 4 mt + r); gaps and rows beyond R leave as zeros
        unsigned vmask = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rowb + mt * 16 + rp[r];
                const bool ok = row < a.R && loadi_or_zero(rpos + row, rpos != nullptr && row < a.R) >= 0;
                vmask |= ok ? (1u << (mt * 4 + r)) : 0u;
            }
        if (nbp < 2) {                   // Q | K: 8 + 8 bytes of hi / lo per 4 channels, 128 contiguous bytes per row and plane
            __bf16* qkh = reinterpret_cast<__bf16*>(a.qk_hi);
            __bf16* qkl = reinterpret_cast<__bf16*>(a.qk_lo);
            const float sc = (nbp == 0) ? a.q_scale : 1.f;
            for_seq_i<0, NB>([&](auto g_tag) __attribute__((always_inline)) {
                constexpr int g = decltype(g_tag)::value;
                for_seq_i<0, MT>([&](auto mt_tag) __attribute__((always_inline)) {
                    constexpr int mt = decltype(mt_tag)::value;
                    f32x4 x[4];
                    for_seq_i<0, 4>([&](auto j_tag) __attribute__((always_inline)) { x[decltype(j_tag)::value] = acc_get<mt * NT + 4 * g + decltype(j_tag)::value>(); });
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rowb + mt * 16 + rp[r];
                        if (row >= a.Rvt) continue;
                        const float f = ((vmask >> (mt * 4 + r)) & 1u) ? sc : 0.f;
                        uint2 hi, lo;
                        split4(f32x4{x[0][r], x[1][r], x[2][r], x[3][r]} * f, hi, lo);
                        const size_t off = (size_t)row * 2 * BN + nbp * BN + col0 + 64 * g;
                        *reinterpret_cast<uint2*>(qkh + off) = hi;
                        *reinterpret_cast<uint2*>(qkl + off) = lo;
                    }
                });
            });
        } else {                         // V: NB 128-column passes through a [BM][132] fp32 tile in LDS -> V^T planes (as gemm_qkv8_bf16)
            float* tile = reinterpret_cast<float*>(smem_q);
            __bf16* vth = reinterpret_cast<__bf16*>(a.vt_hi);
            __bf16* vtl = reinterpret_cast<__bf16*>(a.vt_lo);
            for_seq_i<0, NB>([&](auto pass_tag) __attribute__((always_inline)) {
                constexpr int pass = decltype(pass_tag)::value;
                __syncthreads();         // operand buffers / previous pass's tile are dead
                for_seq_i<0, NB>([&](auto g_tag) __attribute__((always_inline)) {
                    constexpr int g = decltype(g_tag)::value;
                    const int G = wn * NB + g;           // 64-column group of the block (wave-uniform)
                    if ((G >> 1) == pass) {
                        for_seq_i<0, MT>([&](auto mt_tag) __attribute__((always_inline)) {
                            constexpr int mt = decltype(mt_tag)::value;
                            f32x4 x[4];
                            for_seq_i<0, 4>([&](auto j_tag) __attribute__((always_inline)) { x[decltype(j_tag)::value] = acc_get<mt * NT + 4 * g + decltype(j_tag)::value>(); });
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int trow = wm * RW + mt * 16 + rp[r];
                                *reinterpret_cast<f32x4*>(tile + trow * kQkvLd + (((G & 1) * 64 + 4 * lr) ^ qkv_tile_swz(trow))) =
                                    ((vmask >> (mt * 4 + r)) & 1u) ? f32x4{x[0][r], x[1][r], x[2][r], x[3][r]} : f32x4{0.f, 0.f, 0.f, 0.f};
                            }
                        });
                    }
                });
                __syncthreads();
                // column c of the pass, rows 8 j .. 8 j + 7: 4 consecutive lanes share a column (64-byte V^T segments)
#pragma unroll
                for (int u = 0; u < BM / 16; ++u) {
                    const int idx = tid + u * 256;
                    const int c = (idx >> 2) & 127, j = ((idx >> 9) << 2) | (idx & 3);
                    const int row = m0 + 8 * j;
                    if (row >= a.Rvt) continue;
                    const float* t = tile + (8 * j) * kQkvLd + (c ^ qkv_tile_swz(8 * j));      // (rows 8 j .. 8 j + 7 share one swizzle term)
                    const SplitPair sp = split8(make_float4(t[0], t[kQkvLd], t[2 * kQkvLd], t[3 * kQkvLd]),
                                                make_float4(t[4 * kQkvLd], t[5 * kQkvLd], t[6 * kQkvLd], t[7 * kQkvLd]));
                    const size_t off = (size_t)(pass * 128 + c) * a.Rvt + row;
                    *reinterpret_cast<uint4*>(vth + off) = sp.hi;
                    *reinterpret_cast<uint4*>(vtl + off) = sp.lo;
                }
            });
        }
        return;
    }
    float* __restrict__ Y = a.Y;
    void* __restrict__ Yp = a.Yp;
    float* red = reinterpret_cast<float*>(smem_q);      // [2 passes][4 waves][RW rows]
    float mean[MT][4], rstd[MT][4];
    {
        float rsum[MT][4];
        for_seq_i<0, MT>([&](auto mt_tag) __attribute__((always_inline)) {
            constexpr int mt = decltype(mt_tag)::value;
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
            for_seq_i<0, NT>([&](auto n_tag) __attribute__((always_inline)) { s += acc_get<mt * NT + decltype(n_tag)::value>(); });
#pragma unroll
            for (int r = 0; r < 4; ++r) rsum[mt][r] = wave16_sum(s[r]);
        });
        __syncthreads();                                // operand buffers are dead
        if (lr == 0)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * RW + mt * 16 + lg * 4 + r] = rsum[mt][r];
        __syncthreads();
        for_seq_i<0, MT>([&](auto mt_tag) __attribute__((always_inline)) {
            constexpr int mt = decltype(mt_tag)::value;
#pragma unroll
            for (int r = 0; r < 4; ++r) mean[mt][r] = (rsum[mt][r] + red[(wave ^ 1) * RW + mt * 16 + lg * 4 + r]) / (float)a.N;
            f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
            for_seq_i<0, NT>([&](auto n_tag) __attribute__((always_inline)) {
                const f32x4 x = acc_get<mt * NT + decltype(n_tag)::value>();
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = x[r] - mean[mt][r]; q[r] += d * d; }
            });
#pragma unroll
            for (int r = 0; r < 4; ++r) rsum[mt][r] = wave16_sum(q[r]);
        });
        if (lr == 0)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[4 * RW + wave * RW + mt * 16 + lg * 4 + r] = rsum[mt][r];
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                rstd[mt][r] = 1.f / sqrtf((rsum[mt][r] + red[4 * RW + (wave ^ 1) * RW + mt * 16 + lg * 4 + r]) / (float)a.N + a.ln_eps);
    }
    FS2_RT(3)
    constexpr bool PE = EPI == 2;
    const float alpha = (PE && a.pe_alpha) ? a.pe_alpha[0] : 1.f;
    f32x4 gam[NB], bet[NB];
#pragma unroll
    for (int g = 0; g < NB; ++g) { gam[g] = *reinterpret_cast<const f32x4*>(a.ln_g + col0 + 64 * g); bet[g] = *reinterpret_cast<const f32x4*>(a.ln_b + col0 + 64 * g); }
    // EPI 4 (mx4 planes): e2m1 has two exponent bits, so the cross-term copies carry a scale per (row, 16-channel slot) -- the block a lane of the consuming scaled
    // MFMA holds -- taken from the slot's own largest |LayerNorm output|: the slot's 16 channels are this lane's four and those of its three neighbours (lr ^ 1, lr ^ 2),
    // two shuffles inside the store pass below (the first build took ONE scale per row: a fourth pass over the accumulators and a cross-wave reduction, and channels
    // whose gamma differ 80-fold shared it: 5.6e-4 on the mel under such weights, simulated 1.8e-4 per slot).  The byte (2^-11 of the residuals' pre-scale folded
    // in) goes to a.yp_rowscale[row][32]: byte 8 u + 2 (s & 3) + (s >> 2) for slot s of cross unit u, so that a lane of gemm_pl_bf16<.., ARITH = 3> reads the scales of
    // its two MFMAs of a step (slots lg and 4 + lg) as one 16-bit word.
    for_seq_i<0, MT>([&](auto mt_tag) __attribute__((always_inline)) {
        constexpr int mt = decltype(mt_tag)::value;
        int pos[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {           // (the four loads go out together, as address selects: see load4_or_zero)
            const int row = rowb + mt * 16 + rp[r];
            const int pv = loadi_or_zero(rpos + row, rpos != nullptr && row < a.R);
            pos[r] = row < a.R ? pv : -1;
        }
        // (a wave alone on its SIMD has nobody to cover a load's latency: the positional-encoding rows of the whole m-tile are requested here, in
        //  one batch behind the four positions, not one by one in front of their use -- the first build ran this epilogue 13 % behind gemm_row8_bf16's)
        f32x4 pe4[NB][4];
        if constexpr (PE) {
#pragma unroll
            for (int g = 0; g < NB; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) pe4[g][r] = load4_or_zero(a.pe + (size_t)(pos[r] >= 0 ? pos[r] : 0) * a.pe_ld + col0 + 64 * g, pos[r] >= 0);
        }
        for_seq_i<0, NB>([&](auto g_tag) __attribute__((always_inline)) {
            constexpr int g = decltype(g_tag)::value;
            const int col = col0 + 64 * g;
            f32x4 x[4];      // x[j][r]: channel col + j of row r
            for_seq_i<0, 4>([&](auto j_tag) __attribute__((always_inline)) { x[decltype(j_tag)::value] = acc_get<mt * NT + 4 * g + decltype(j_tag)::value>(); });
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = rowb + mt * 16 + rp[r];
                const bool live = pos[r] >= 0;
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = (x[j][r] - mean[mt][r]) * rstd[mt][r];
                    t = t * gam[g][j] + bet[g][j];
                    if constexpr (PE) { t = fmaxf(t, 0.f); t = t * a.x_scale + alpha * pe4[g][r][j]; }
                    v[j] = live ? t : 0.f;
                }
                int eb4 = 0;
                if constexpr (EPI == 4) {      // (unconditional: every lane of the quad takes part in the shuffles; a quad shares its row)
                    float m4 = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
                    m4 = fmaxf(m4, __shfl_xor(m4, 1));
                    m4 = fmaxf(m4, __shfl_xor(m4, 2));
                    eb4 = mx4_scale_byte(m4);
                }
                if (row < a.R) {
                    if constexpr (RES == 0) *reinterpret_cast<f32x4*>(Y + (size_t)row * a.ldy + col) = v;
                    if constexpr (EPI == 4) {
                        store_planes4_mx4(Yp, row, a.yp_chunks, col, v, a.yp_scale, mx4_inv_scale(eb4));
                        if ((lr & 3) == 0) a.yp_rowscale[(size_t)row * 32 + (col >> 7) * 8 + ((col >> 4) & 3) * 2 + ((col >> 6) & 1)] = (unsigned char)(eb4 - 11);
                    }
                    else if constexpr (EPI == 1) store_planes4_mx(Yp, row, a.yp_chunks, col, v, a.yp_scale);
                    else store_planes4(Yp, row, a.yp_chunks, col, v);
                }
            }
        });
    });
    FS2_RT(4)
}

}  // namespace fs2
