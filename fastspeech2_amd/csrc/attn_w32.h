// attn_w32: split-bf16 flash-style self-attention with 32 queries per wavefront on v_mfma_f32_32x32x16_bf16, one wavefront per
// SIMD (the whole 512-entry register file), K / V^T tiles brought in by LDS-DMA into a ring of three.  Same arithmetic as
// attn_bf16<DK,3> (every operand x = hi + lo in bf16, lo*hi + hi*lo + hi*hi, fp32 accumulate, base-2 online softmax);
// replaces reference core/attention.py:55-70 for the launches whose grid fills the chip (fs2_runtime.hip: use_attn_w32 -- the
// decoder's at c2 .. c5, the encoder's at c4).
//
// Why a second kernel: attn_bf16 keeps 16 queries per wave, so every 1-KB fragment read from
// LDS feeds ONE 16-cycle MFMA, a tile costs two workgroup barriers, and 238 registers leave no room to overlap anything.  Here
//   * a workgroup = 4 waves x 32 queries = 128 queries of one (utterance, head); a fragment read feeds a 32-cycle MFMA;
//   * S^T = K.Q^T (A = K rows from LDS, B = Q^T from registers): lane (q = l&31, hi = l>>5) ends up with the scores of its own query
//     against 16 keys; the row r of the C layout is (r&3) + 8 (r>>2) + 4 hi, and LDS row rho of the K tile holds key
//     swap_bits_2_3(rho), so that register r of lane (q, hi) is key 16 (r>>3) + 8 hi + (r&7): registers 8m .. 8m+7 are exactly the
//     k-slots 8 hi .. 8 hi + 7 of the m-th 16-key MFMA of the next product, against V^T stored in natural key order;
//   * O^T = V^T.P^T (A = V^T rows from LDS, B = P^T from registers): column = the lane's own query, so the running max and the
//     normaliser are lane-local (attn_bf16 fetches them with ds_bpermute);
//   * the wave is alone on its SIMD, so nothing but its own instruction stream covers the softmax: the exponentials of tile t sit
//     between the MFMAs of Q.K^T of tile t + 1, the score sums of tile t + 1 between the MFMAs of P.V of tile t, the LDS-DMA pieces
//     of the tiles to come in the slots with the least other work (K pieces in Q.K^T, V^T pieces in P.V) -- one MFMA per "slot",
//     every slot closed by a scheduling fence; ONE workgroup barrier
//     per tile; rings of THREE tiles per operand, so that a DMA piece has two iterations to land (the closing wait of an iteration is
//     vmcnt(this iteration's own pieces): an LDS-DMA round trip under load is ~2,500 cycles, half a tile);
//   * NO rescale of O and no running maximum inside the loop: the reference maximum m of a row is the maximum of its FIRST tile; the
//     S^T accumulator chain starts from -m instead of 0 (a 16-register tuple set once), so scores arrive relative to it and go straight
//     into v_exp.  P = 2^(s - m) may exceed 1 -- bf16 has fp32's exponent range and hi + lo is relative, so that costs the split
//     arithmetic nothing -- as long as a row's probabilities of one tile sum to less than 2^60 (O and the normaliser then stay far
//     inside fp32).  A wave that meets a larger sum -- a score 41 nats above everything in the row's first 32 keys -- finishes the
//     tile, stops computing, keeps feeding the DMA ring, and afterwards redoes its 32 rows in a plain fp32 two-pass loop
//     (attn_w32_rows_slow).  Reason: a single compiler-visible VALU use of the O accumulators inside the loop makes hipcc treat
//     them as either-file values and copy all 96 registers into and out of the accumulator file around every P.V phase.
// Prologue and epilogue: the wave's 32 Q rows arrive as one more K-layout tile by LDS-DMA, staged in the ring slots the first K / V^T
// tiles do not use yet, and are read into the accumulator file once; the context leaves as split-bf16 planes staged through the (then
// idle) ring memory -- behind `s_waitcnt vmcnt(0)` + a barrier: pieces of the redundant last fetch may still be in flight -- and is
// stored as whole rows (fp32 output, the operator tests' form, goes straight from the registers).
// Register files (the MFMAs are inline asm so that the file of every operand is ours to choose; left alone, hipcc's allocator
// shuffled 1,700 v_accvgpr_* per kernel and spilled): O^T (96 registers at d_k = 192) and the Q fragments (96) live in the
// accumulator half, everything the VALU touches in the architectural half (<= 170).  hipcc pads no hazards around asm (cdna guide
// section 5.7): an accumulate chain needs none; a VALU-written operand gets `s_nop 1` ahead of the MFMA that reads it; every other
// reader of an MFMA result sits behind at least six other MFMAs or an explicit 16-state drain.  tools/probes/audit_w32.py checks the
// generated ISA for all of this (no compiler v_accvgpr_* on O^T's registers, no scratch in the loop, the pads) after every edit.
// LDS image (per ring slot): K tile = 32 rows x [hi DK | lo DK] bf16 (row stride a multiple of 256 B), 16-byte slot s of row rho
// stored at slot (s & ~15) | ((s & 15) ^ (rho & 15)); V^T tile = DK rows x [hi 32 keys | lo 32 keys] = 128 B, slot s of row n at
// s ^ ((n >> 1) & 7).  Each ds_read_b128 lane group ({0-3,12-15,20-27}, ...) then covers 16 distinct 16-byte bank slots: rows with
// 16 distinct values of rho & 15 resp. n & 15.  The swizzle costs nothing: LDS-DMA writes lane j of an instruction to LDS byte
// 16 j of its 1-KB piece, and the lane chooses which global 16 bytes go there.
#pragma once
#include <type_traits>
#include <utility>
#include "attn_bf16.h"

namespace fs2 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DK>
constexpr size_t attn_w32_lds_bytes() { return (size_t)3 * (32 * DK * 4 + DK * 128); }      // K ring + V^T ring of three tiles each

template <class F, int... I>
__device__ __forceinline__ void for_seq(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

// O^T never appears as a compiler value: n-tile N IS a[16 N : 16 N + 15], written literally in the statements below and listed as
// their clobbers (which keeps every value that lives across the tile loop -- the Q fragments -- out of a[0 : 16 NT) and makes the
// kernel descriptor allocate the range).  With O as "+a" operands hipcc kept two copies of every accumulator tuple and moved 64-96
// registers between them per P.V phase; pinned with "{a[..]}" it kept O in the architectural file and copied it in and out.
// AUDIT after every edit (tools/probes/audit_w32.py): no compiler v_accvgpr_* touching a0 .. a(16 NT - 1), no scratch.
template <int N>
__device__ __forceinline__ void mfma_o0(const bf16x8_t& z_v) {                             // a[16N..] = 0 (padded: z_v is VALU-written)
    if constexpr (N == 0) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[0:15], %0, %0, 0" : : "v"(z_v) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    else if constexpr (N == 1) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %0, %0, 0" : : "v"(z_v) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    else if constexpr (N == 2) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[32:47], %0, %0, 0" : : "v"(z_v) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    else if constexpr (N == 3) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[48:63], %0, %0, 0" : : "v"(z_v) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    else if constexpr (N == 4) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[64:79], %0, %0, 0" : : "v"(z_v) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    else if constexpr (N == 5) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[80:95], %0, %0, 0" : : "v"(z_v) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    else if constexpr (N == 6) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[96:111], %0, %0, 0" : : "v"(z_v) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    else if constexpr (N == 7) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 a[112:127], %0, %0, 0" : : "v"(z_v) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
}
template <int N>
__device__ __forceinline__ void read_o(float (&e)[16]) {                                    // a[16N..] -> architectural registers
    if constexpr (N == 0) asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\tv_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7\n\tv_accvgpr_read_b32 %8, a8\n\tv_accvgpr_read_b32 %9, a9\n\tv_accvgpr_read_b32 %10, a10\n\tv_accvgpr_read_b32 %11, a11\n\tv_accvgpr_read_b32 %12, a12\n\tv_accvgpr_read_b32 %13, a13\n\tv_accvgpr_read_b32 %14, a14\n\tv_accvgpr_read_b32 %15, a15" : "=v"(e[0]), "=v"(e[1]), "=v"(e[2]), "=v"(e[3]), "=v"(e[4]), "=v"(e[5]), "=v"(e[6]), "=v"(e[7]), "=v"(e[8]), "=v"(e[9]), "=v"(e[10]), "=v"(e[11]), "=v"(e[12]), "=v"(e[13]), "=v"(e[14]), "=v"(e[15]));
    else if constexpr (N == 1) asm volatile("v_accvgpr_read_b32 %0, a16\n\tv_accvgpr_read_b32 %1, a17\n\tv_accvgpr_read_b32 %2, a18\n\tv_accvgpr_read_b32 %3, a19\n\tv_accvgpr_read_b32 %4, a20\n\tv_accvgpr_read_b32 %5, a21\n\tv_accvgpr_read_b32 %6, a22\n\tv_accvgpr_read_b32 %7, a23\n\tv_accvgpr_read_b32 %8, a24\n\tv_accvgpr_read_b32 %9, a25\n\tv_accvgpr_read_b32 %10, a26\n\tv_accvgpr_read_b32 %11, a27\n\tv_accvgpr_read_b32 %12, a28\n\tv_accvgpr_read_b32 %13, a29\n\tv_accvgpr_read_b32 %14, a30\n\tv_accvgpr_read_b32 %15, a31" : "=v"(e[0]), "=v"(e[1]), "=v"(e[2]), "=v"(e[3]), "=v"(e[4]), "=v"(e[5]), "=v"(e[6]), "=v"(e[7]), "=v"(e[8]), "=v"(e[9]), "=v"(e[10]), "=v"(e[11]), "=v"(e[12]), "=v"(e[13]), "=v"(e[14]), "=v"(e[15]));
    else if constexpr (N == 2) asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35\n\tv_accvgpr_read_b32 %4, a36\n\tv_accvgpr_read_b32 %5, a37\n\tv_accvgpr_read_b32 %6, a38\n\tv_accvgpr_read_b32 %7, a39\n\tv_accvgpr_read_b32 %8, a40\n\tv_accvgpr_read_b32 %9, a41\n\tv_accvgpr_read_b32 %10, a42\n\tv_accvgpr_read_b32 %11, a43\n\tv_accvgpr_read_b32 %12, a44\n\tv_accvgpr_read_b32 %13, a45\n\tv_accvgpr_read_b32 %14, a46\n\tv_accvgpr_read_b32 %15, a47" : "=v"(e[0]), "=v"(e[1]), "=v"(e[2]), "=v"(e[3]), "=v"(e[4]), "=v"(e[5]), "=v"(e[6]), "=v"(e[7]), "=v"(e[8]), "=v"(e[9]), "=v"(e[10]), "=v"(e[11]), "=v"(e[12]), "=v"(e[13]), "=v"(e[14]), "=v"(e[15]));
    else if constexpr (N == 3) asm volatile("v_accvgpr_read_b32 %0, a48\n\tv_accvgpr_read_b32 %1, a49\n\tv_accvgpr_read_b32 %2, a50\n\tv_accvgpr_read_b32 %3, a51\n\tv_accvgpr_read_b32 %4, a52\n\tv_accvgpr_read_b32 %5, a53\n\tv_accvgpr_read_b32 %6, a54\n\tv_accvgpr_read_b32 %7, a55\n\tv_accvgpr_read_b32 %8, a56\n\tv_accvgpr_read_b32 %9, a57\n\tv_accvgpr_read_b32 %10, a58\n\tv_accvgpr_read_b32 %11, a59\n\tv_accvgpr_read_b32 %12, a60\n\tv_accvgpr_read_b32 %13, a61\n\tv_accvgpr_read_b32 %14, a62\n\tv_accvgpr_read_b32 %15, a63" : "=v"(e[0]), "=v"(e[1]), "=v"(e[2]), "=v"(e[3]), "=v"(e[4]), "=v"(e[5]), "=v"(e[6]), "=v"(e[7]), "=v"(e[8]), "=v"(e[9]), "=v"(e[10]), "=v"(e[11]), "=v"(e[12]), "=v"(e[13]), "=v"(e[14]), "=v"(e[15]));
    else if constexpr (N == 4) asm volatile("v_accvgpr_read_b32 %0, a64\n\tv_accvgpr_read_b32 %1, a65\n\tv_accvgpr_read_b32 %2, a66\n\tv_accvgpr_read_b32 %3, a67\n\tv_accvgpr_read_b32 %4, a68\n\tv_accvgpr_read_b32 %5, a69\n\tv_accvgpr_read_b32 %6, a70\n\tv_accvgpr_read_b32 %7, a71\n\tv_accvgpr_read_b32 %8, a72\n\tv_accvgpr_read_b32 %9, a73\n\tv_accvgpr_read_b32 %10, a74\n\tv_accvgpr_read_b32 %11, a75\n\tv_accvgpr_read_b32 %12, a76\n\tv_accvgpr_read_b32 %13, a77\n\tv_accvgpr_read_b32 %14, a78\n\tv_accvgpr_read_b32 %15, a79" : "=v"(e[0]), "=v"(e[1]), "=v"(e[2]), "=v"(e[3]), "=v"(e[4]), "=v"(e[5]), "=v"(e[6]), "=v"(e[7]), "=v"(e[8]), "=v"(e[9]), "=v"(e[10]), "=v"(e[11]), "=v"(e[12]), "=v"(e[13]), "=v"(e[14]), "=v"(e[15]));
    else if constexpr (N == 5) asm volatile("v_accvgpr_read_b32 %0, a80\n\tv_accvgpr_read_b32 %1, a81\n\tv_accvgpr_read_b32 %2, a82\n\tv_accvgpr_read_b32 %3, a83\n\tv_accvgpr_read_b32 %4, a84\n\tv_accvgpr_read_b32 %5, a85\n\tv_accvgpr_read_b32 %6, a86\n\tv_accvgpr_read_b32 %7, a87\n\tv_accvgpr_read_b32 %8, a88\n\tv_accvgpr_read_b32 %9, a89\n\tv_accvgpr_read_b32 %10, a90\n\tv_accvgpr_read_b32 %11, a91\n\tv_accvgpr_read_b32 %12, a92\n\tv_accvgpr_read_b32 %13, a93\n\tv_accvgpr_read_b32 %14, a94\n\tv_accvgpr_read_b32 %15, a95" : "=v"(e[0]), "=v"(e[1]), "=v"(e[2]), "=v"(e[3]), "=v"(e[4]), "=v"(e[5]), "=v"(e[6]), "=v"(e[7]), "=v"(e[8]), "=v"(e[9]), "=v"(e[10]), "=v"(e[11]), "=v"(e[12]), "=v"(e[13]), "=v"(e[14]), "=v"(e[15]));
    else if constexpr (N == 6) asm volatile("v_accvgpr_read_b32 %0, a96\n\tv_accvgpr_read_b32 %1, a97\n\tv_accvgpr_read_b32 %2, a98\n\tv_accvgpr_read_b32 %3, a99\n\tv_accvgpr_read_b32 %4, a100\n\tv_accvgpr_read_b32 %5, a101\n\tv_accvgpr_read_b32 %6, a102\n\tv_accvgpr_read_b32 %7, a103\n\tv_accvgpr_read_b32 %8, a104\n\tv_accvgpr_read_b32 %9, a105\n\tv_accvgpr_read_b32 %10, a106\n\tv_accvgpr_read_b32 %11, a107\n\tv_accvgpr_read_b32 %12, a108\n\tv_accvgpr_read_b32 %13, a109\n\tv_accvgpr_read_b32 %14, a110\n\tv_accvgpr_read_b32 %15, a111" : "=v"(e[0]), "=v"(e[1]), "=v"(e[2]), "=v"(e[3]), "=v"(e[4]), "=v"(e[5]), "=v"(e[6]), "=v"(e[7]), "=v"(e[8]), "=v"(e[9]), "=v"(e[10]), "=v"(e[11]), "=v"(e[12]), "=v"(e[13]), "=v"(e[14]), "=v"(e[15]));
    else if constexpr (N == 7) asm volatile("v_accvgpr_read_b32 %0, a112\n\tv_accvgpr_read_b32 %1, a113\n\tv_accvgpr_read_b32 %2, a114\n\tv_accvgpr_read_b32 %3, a115\n\tv_accvgpr_read_b32 %4, a116\n\tv_accvgpr_read_b32 %5, a117\n\tv_accvgpr_read_b32 %6, a118\n\tv_accvgpr_read_b32 %7, a119\n\tv_accvgpr_read_b32 %8, a120\n\tv_accvgpr_read_b32 %9, a121\n\tv_accvgpr_read_b32 %10, a122\n\tv_accvgpr_read_b32 %11, a123\n\tv_accvgpr_read_b32 %12, a124\n\tv_accvgpr_read_b32 %13, a125\n\tv_accvgpr_read_b32 %14, a126\n\tv_accvgpr_read_b32 %15, a127" : "=v"(e[0]), "=v"(e[1]), "=v"(e[2]), "=v"(e[3]), "=v"(e[4]), "=v"(e[5]), "=v"(e[6]), "=v"(e[7]), "=v"(e[8]), "=v"(e[9]), "=v"(e[10]), "=v"(e[11]), "=v"(e[12]), "=v"(e[13]), "=v"(e[14]), "=v"(e[15]));
}
// Single MFMAs of the S^T chains (accumulator in the architectural file, B operand from the accumulator file).  KIND 0: acc += A.B;
// 1: acc = A.B + cm (the chain am starts from the row's negated reference maximum); 2: acc = A.B.
template <int KIND>
__device__ __forceinline__ void mfma_s(f32x16& acc, const bf16x8_t& a_v, const bf16x8_t& b_a, const f32x16& cm) {
    if constexpr (KIND == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a_v), "a"(b_a), "v"(cm));
    else if constexpr (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a_v), "a"(b_a));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a_v), "a"(b_a));
}
// O^T n-tile N (a[16 N : 16 N + 15]) += A.B; PAD: `s_nop 1` ahead (a VALU-written operand)
template <int N, bool PAD>
__device__ __forceinline__ void mfma_o(const bf16x8_t& a_v, const bf16x8_t& b_v) {
    if constexpr (N == 0) {
        if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" : : "v"(a_v), "v"(b_v) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" : : "v"(a_v), "v"(b_v) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15");
    }
    else if constexpr (N == 1) {
        if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" : : "v"(a_v), "v"(b_v) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" : : "v"(a_v), "v"(b_v) : "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31");
    }
    else if constexpr (N == 2) {
        if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" : : "v"(a_v), "v"(b_v) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" : : "v"(a_v), "v"(b_v) : "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
    }
    else if constexpr (N == 3) {
        if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" : : "v"(a_v), "v"(b_v) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" : : "v"(a_v), "v"(b_v) : "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63");
    }
    else if constexpr (N == 4) {
        if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" : : "v"(a_v), "v"(b_v) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[64:79], %0, %1, a[64:79]" : : "v"(a_v), "v"(b_v) : "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79");
    }
    else if constexpr (N == 5) {
        if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" : : "v"(a_v), "v"(b_v) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[80:95], %0, %1, a[80:95]" : : "v"(a_v), "v"(b_v) : "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95");
    }
    else if constexpr (N == 6) {
        if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" : : "v"(a_v), "v"(b_v) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[96:111], %0, %1, a[96:111]" : : "v"(a_v), "v"(b_v) : "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111");
    }
    else if constexpr (N == 7) {
        if constexpr (PAD) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" : : "v"(a_v), "v"(b_v) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
        else asm volatile("v_mfma_f32_32x32x16_bf16 a[112:127], %0, %1, a[112:127]" : : "v"(a_v), "v"(b_v) : "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127");
    }
}

// (Issuing the three MFMAs of a k-step / the six of a P.V group as ONE asm statement -- one compiler wait per block, MFMAs back to back --
//  measured 2-5 % slower than one statement per MFMA with the other work spread between them: profiles/r04_attn_w32_blocked_mfma_statements.txt.)
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 7\n\ts_nop 7" ::: "memory"); }      // 16 states: an 8-pass MFMA result is readable

// Ablations for tools/probes/attn_w32_probe.hip (timing only, results wrong): FS2_W32_ABL bit 6 (64): no V^T pieces; bit 7 (128): no K pieces; bit 4 (16): no closing vmcnt wait; bit 5 (32): no closing barrier; bit 0: no DMA pieces in the tile loop; bit 1: no
// softmax work in the blocks; bit 2: no closing wait / barrier; bit 3: no fragment reads inside the blocks (the first ones are reused);
// bit 8 (256): P.V at the MFMA count of an fp16 + e4m3 form: the V_lo . P_hi MFMAs are not issued -- 4 instead of 6 per group, what
// one fp16 MFMA + two block-scaled 8-bit cross terms at half rate would cost the matrix pipe; an optimistic bound (that form also reads e4m3 planes).
#ifndef FS2_W32_ABL
#define FS2_W32_ABL 0
#endif

// Phase timing for tools/probes/attn_w32_probe.hip (compiled only with -DFS2_W32_TIMING): cycles of wave 0 of workgroup (0, 0) per
// phase of the tile loop: [0] prep + phase A, [1] phase B (+ head), [2] DMA wait, [3] barrier, [4] tiles counted.
#ifdef FS2_W32_TIMING
__device__ long long g_w32_phase[16];
// (accumulated in scalar registers and written once at the end: a read-modify-write of global memory per stamp put a memory round trip
//  into every interval it opened)
#define FS2_WT(i) { const long long t_ = __builtin_readcyclecounter(); wstamp[i] += t_ - tprev; tprev = t_; }
#else
#define FS2_WT(i)
#endif

constexpr float kW32SumLimit = 1.15e18f;   // 2^60: a row whose probabilities (relative to its first tile's maximum) sum to more than this in one tile leaves the fast path

// The 32 rows of one wave again, from the operand planes in global memory, in plain fp32 (two passes: row maximum, then sums).
// Only for waves that left the fast path (kW32SumLimit); lane (q = l&31, hi) owns query q0 + q and the head channels
// [hi DK/2, (hi + 1) DK/2).  Same operands (hi + lo is exact in fp32), fp32 products: at least as accurate as the MFMA path.
template <int DK>
__device__ __noinline__ void attn_w32_rows_slow(const AttnB16Args a, int s0, int len, int klen, int q0, int h, int lane) {      // (a by value: a reference would move the kernel's own copy of the arguments into scratch memory)
    const int l31 = lane & 31, hi = lane >> 5;
    const int qrow = q0 + l31;
    if (lane == 0) {      // one count per wave (fs2_get_counter "attn_slow_path_waves"; fs2_decode_io.status[5])
        if (a.slow_count) atomicAdd(a.slow_count, 1);
        if (a.slow_count2) atomicAdd(a.slow_count2, 1);
    }
    if (qrow >= len) return;
    const __bf16* qh = a.qk_hi + (size_t)(s0 + qrow) * a.ldqk + (size_t)h * DK;
    const __bf16* ql = a.qk_lo + (size_t)(s0 + qrow) * a.ldqk + (size_t)h * DK;
    auto score = [&](int key) {
        const __bf16* kh = a.qk_hi + (size_t)(s0 + key) * a.ldqk + a.D + (size_t)h * DK;
        const __bf16* kl = a.qk_lo + (size_t)(s0 + key) * a.ldqk + a.D + (size_t)h * DK;
        float s = 0.f;
        for (int d = 0; d < DK; ++d) s = fmaf((float)qh[d] + (float)ql[d], (float)kh[d] + (float)kl[d], s);
        return s;
    };
    float m = -INFINITY;
    for (int key = 0; key < klen; ++key) m = fmaxf(m, score(key));
    constexpr int NC = DK / 2;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    float l = 0.f;
    const __bf16* vh = a.vt_hi + ((size_t)h * DK + (size_t)hi * NC) * a.Rvt + s0;
    const __bf16* vl = a.vt_lo + ((size_t)h * DK + (size_t)hi * NC) * a.Rvt + s0;
    for (int key = 0; key < klen; ++key) {
        const float p = __builtin_amdgcn_exp2f(score(key) - m);
        l += p;
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = fmaf(p, (float)vh[(size_t)c * a.Rvt + key] + (float)vl[(size_t)c * a.Rvt + key], acc[c]);
    }
    const float linv = (l > 0.f) ? 1.f / l : 0.f;
    const bool dead = a.mask_q && qrow >= klen;
    const size_t row = (size_t)(s0 + qrow);
#pragma unroll
    for (int c = 0; c < NC; c += 4) {
        const int col = h * DK + hi * NC + c;
        f32x4 v = f32x4{acc[c], acc[c + 1], acc[c + 2], acc[c + 3]} * linv;
        if (dead) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.ctx) *reinterpret_cast<f32x4*>(a.ctx + row * a.ldc + col) = v;
        if (a.ctxp) store_planes4(a.ctxp, row, a.ctxp_chunks, col, v);
    }
}

template <int DK>
__global__ __launch_bounds__(256, 1) void attn_w32(AttnB16Args a) {
    static_assert(DK % 64 == 0 && DK <= 256, "head dim: a multiple of 64 up to 256");
    extern __shared__ __attribute__((aligned(16))) char smem_w[];
    constexpr int KSL = DK / 8;           // 16-byte slots per plane of a K row
    constexpr int KROWB = DK * 4;         // bytes per K row [hi | lo]
    constexpr int KB = 32 * KROWB;        // bytes per K tile
    constexpr int VB = DK * 128;          // bytes per V^T tile
    constexpr int NKS = DK / 16;          // k-steps of Q.K^T
    constexpr int NT = DK / 32;           // 32-channel n-tiles of O^T
    constexpr int PW = DK / 32;           // 1-KB DMA pieces per wave and operand tile (KB / 1024 / 4 == VB / 1024 / 4)
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef FS2_W32_TIMING
    long long wstamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    long long tpro = __builtin_readcyclecounter();
#define FS2_WP(i) { const long long t_ = __builtin_readcyclecounter(); wstamp[i] += t_ - tpro; tpro = t_; }
#else
#define FS2_WP(i)
#endif
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    {
        const int nitems = a.nwork != nullptr ? *a.nwork : a.nitems;
        if ((int)blockIdx.x >= nitems) return;
    }
    // (everything the control flow hangs on as explicit scalars: loaded through the vector cache, hipcc otherwise keeps the tile loop's
    //  counter and conditions in vector registers and branches through EXEC masks)
    const int2 wk = a.work[blockIdx.x];
    const int b = __builtin_amdgcn_readfirstlane(wk.x), h = blockIdx.y;
    if (b < 0) return;                         // padding entry of the XCD-interleaved work list
    const int s0 = __builtin_amdgcn_readfirstlane(a.start[b]), len = __builtin_amdgcn_readfirstlane(a.len[b]);
    const int klen = __builtin_amdgcn_readfirstlane(a.klen[b]);
    const int q0 = __builtin_amdgcn_readfirstlane(wk.y) * kAttBlk + wave * 32;
    const bool wave_live = q0 < len;           // wave-uniform: a wave whose 32 queries lie beyond the utterance only feeds the DMA
    const int ntiles = (klen + 31) >> 5;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)smem_w);
    FS2_WP(8)

    // ---- DMA source offsets (loop invariant): piece p = 4u + wave of a tile, this lane's 16 bytes = physical slot 64 p + lane.
    // (ln: the lane index, passed in so that the slow paths can hand over an opaque copy -- otherwise hipcc hoists their whole address
    //  arithmetic out of the tile loop and keeps it in registers the loop needs)
    auto k_piece = [&](int u, int ln, int& key, int& byte_in_row) __attribute__((always_inline)) {        // -> key inside the tile, byte offset inside the row of the hi plane (lo plane added)
        const int G = (u * 4 + wave) * 64 + ln;
        const int rho = G / (2 * KSL), ps = G - rho * (2 * KSL);
        const int s = (ps & ~15) | ((ps & 15) ^ (rho & 15));
        const int plane = s >= KSL;
        key = (rho & 19) | ((rho & 4) << 1) | ((rho & 8) >> 1);
        byte_in_row = (s - plane * KSL) * 16 + plane * (int)a.qk_lo_bytes;
    };
    auto v_piece = [&](int u, int ln, int& n, int& j, int& plane) __attribute__((always_inline)) {        // -> head channel, 8-key group inside the tile, plane
        const int G = (u * 4 + wave) * 64 + ln;
        n = G >> 3;
        const int s = (G & 7) ^ ((n >> 1) & 7);
        j = s & 3; plane = s >> 2;
    };
    unsigned kgo[PW], vgo[PW];
#pragma unroll
    for (int u = 0; u < PW; ++u) {
        int key, bir, n, j, plane;
        k_piece(u, lane, key, bir);
        kgo[u] = (unsigned)(key * a.ldqk * 2 + bir);
        v_piece(u, lane, n, j, plane);
        vgo[u] = (unsigned)n * (unsigned)a.Rvt * 2u + (unsigned)j * 16u + (unsigned)plane * a.vt_lo_bytes;
    }
    gchar_t* kbase = uniform_ptr(reinterpret_cast<const char*>(a.qk_hi + a.D + (size_t)h * DK) + (size_t)s0 * a.ldqk * 2);
    gchar_t* vbase = uniform_ptr(reinterpret_cast<const char*>(a.vt_hi + (size_t)h * DK * a.Rvt + s0));

    // K tile kt -> ring slot buf.  Keys beyond klen (last tile): rows clamped to the last key (their scores are masked to -inf).
    auto issue_K = [&](int kt, int buf) __attribute__((always_inline)) {
        const int key0 = kt * 32;
        const unsigned dst = lds0 + buf * KB + wave * 1024;
        if (key0 + 32 <= klen) {
            gchar_t* base = uniform_ptr((const void*)(kbase + (size_t)key0 * a.ldqk * 2));
            asm volatile("s_nop 4" ::: "memory");      // v_readfirstlane -> SGPR base of a VMEM instruction: 5 states (hipcc pads nothing around asm)
#pragma unroll
            for (int u = 0; u < PW; ++u) dma16_so(base, kgo[u], dst + u * 4096);
        } else {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                int key, bir;
                k_piece(u, ln, key, bir);
                dma16((const void*)(kbase + (size_t)min(key0 + key, klen - 1) * a.ldqk * 2 + bir), dst + u * 4096);
            }
        }
    };
    // V^T tile kt -> ring slot buf.  8-key vectors that start beyond klen come from the zero vector (P = 0 times a non-finite V
    // would poison P.V, and the rows behind an utterance are not this call's data); vectors that straddle klen are masked in LDS by
    // the lane that fetched them (fix_V, after its vmcnt wait, ahead of the barrier).
    auto issue_V = [&](int kt, int buf) __attribute__((always_inline)) {
        const int key0 = kt * 32;
        const unsigned dst = lds0 + 3 * KB + buf * VB + wave * 1024;
        gchar_t* base = uniform_ptr((const void*)(vbase + (size_t)key0 * 2));
        if (key0 + 32 <= klen) {
            asm volatile("s_nop 4" ::: "memory");
#pragma unroll
            for (int u = 0; u < PW; ++u) dma16_so(base, vgo[u], dst + u * 4096);
        } else {
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                int n, j, plane;
                v_piece(u, ln, n, j, plane);
                const int nv = klen - (key0 + 8 * j);
                const unsigned go = (unsigned)n * (unsigned)a.Rvt * 2u + (unsigned)j * 16u + (unsigned)plane * a.vt_lo_bytes;
                const void* sp = (nv > 0) ? (const void*)(base + go) : static_cast<const void*>(g_zero16);
                dma16(sp, dst + u * 4096);
            }
        }
    };
    auto fix_V = [&](int kt, int buf) __attribute__((always_inline)) {         // call with this wave's DMA complete
        const int key0 = kt * 32;
        if (key0 + 32 <= klen || (klen & 7) == 0) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int u = 0; u < PW; ++u) {
            int n, j, plane;
            v_piece(u, ln, n, j, plane);
            const int nv = klen - (key0 + 8 * j);
            if (nv > 0 && nv < 8) {
                u32x4* p = reinterpret_cast<u32x4*>(smem_w + 3 * KB + buf * VB + (u * 4 + wave) * 1024 + ln * 16);
                u32x4 v = *p;
#pragma unroll
                for (int w = 0; w < 4; ++w) v[w] &= (nv > 2 * w + 1) ? 0xffffffffu : ((nv > 2 * w) ? 0x0000ffffu : 0u);
                *p = v;
            }
        }
    };

    // ---- fragment addresses: one lane-dependent base per even 16-byte column, everything else is an immediate
    // K: lane reads row rho = l31, logical slot s = plane KSL + 2c + hi; (s & 15) ^ (rho & 15) = E ^ g with E = (plane KSL + 2c) & 15 even, g = hi ^ (rho & 15)
    const char* kp[8];
    {
        const int g = hi ^ (l31 & 15);
#pragma unroll
        for (int e = 0; e < 8; ++e) kp[e] = smem_w + l31 * KROWB + ((((2 * e) ^ (g & 14)) | (g & 1)) << 4);
    }
    // V^T: lane reads row n = 32 nt + l31, logical slot s = 4 plane + 2m + hi; s ^ f = E ^ g with E = 4 plane + 2m, g = hi ^ ((l31 >> 1) & 7)
    const char* vp[4];
    {
        const int g = hi ^ ((l31 >> 1) & 7);
#pragma unroll
        for (int e = 0; e < 4; ++e) vp[e] = smem_w + 3 * KB + l31 * 128 + ((((2 * e) ^ (g & 6)) | (g & 1)) << 4);
    }

    FS2_WP(9)
    if (ntiles > 0) issue_K(0, 0);
    // Q fragments (B operand of S^T): row q0 + l31, d = 16c + 8 hi .. + 7.  The wave's 32 Q rows come in as ONE more tile of the K layout
    // (LDS-DMA, coalesced 384-byte plane rows; rows beyond the utterance repeat its last row: their results are never stored) into a
    // private region -- K slots 1, 2 and V^T slots 0, 1, free until the first barrier -- and the fragments are read from there.  (Loaded
    // straight from global memory, 24 x 16 bytes per lane at a 1.5-KB row stride, the Q loads were 10,500 of the 26,000 cycles a
    // workgroup spends outside its tile loop.)
    static_assert(KB == VB, "the Q staging region spans K and V^T ring slots of equal size");
    bf16x8_t qh[NKS], ql[NKS];
    {       // (every wave, also one whose rows all lie beyond the utterance: no second definition of the fragments for hipcc to merge)
        gchar_t* qbase = (gchar_t*)reinterpret_cast<const char*>(a.qk_hi + (size_t)h * DK);
#pragma unroll
        for (int pc = 0; pc < KSL; ++pc) {
            const int G = pc * 64 + lane;
            const int rho = G / (2 * KSL), ps = G - rho * (2 * KSL);
            const int sl = (ps & ~15) | ((ps & 15) ^ (rho & 15));
            const int plane = sl >= KSL;
            const int qr = min(q0 + rho, len - 1);
            dma16((const void*)(qbase + (size_t)(s0 + qr) * a.ldqk * 2 + (sl - plane * KSL) * 16 + (size_t)plane * a.qk_lo_bytes), lds0 + (1 + wave) * KB + pc * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int qoff = (1 + wave) * KB;
#pragma unroll
        for (int c = 0; c < NKS; ++c) {
            const int sh = 2 * c, sl = KSL + 2 * c;
            qh[c] = *reinterpret_cast<const bf16x8_t*>(kp[(sh & 15) >> 1] + qoff + (sh >> 4) * 256);
            ql[c] = *reinterpret_cast<const bf16x8_t*>(kp[(sl & 15) >> 1] + qoff + (sl >> 4) * 256);
            // from here on the fragments ARE accumulator-file values (every later use asks for one): without this hipcc keeps them in the
            // architectural file and copies each into a[0:3] -- O^T's registers, free in its eyes between two P.V statements -- at every use
            asm volatile("" : "+a"(qh[c]), "+a"(ql[c]));
        }
    }
    FS2_WP(10)
    {      // O^T = 0 (a[0 : 16 NT))
        const bf16x8_t z = __builtin_bit_cast(bf16x8_t, u32x4{0, 0, 0, 0});
        for_seq([&](auto n_tag) __attribute__((always_inline)) { mfma_o0<decltype(n_tag)::value>(z); }, std::make_integer_sequence<int, NT>{});
    }
    float l_run = 0.f;
    int bail = 0;                    // wave-uniform: this wave left the fast path (kW32SumLimit)
    f32x16 am, ac;                   // S^T accumulator chains: am the hi.hi products (started from -m), ac both cross terms (lo.hi, hi.lo)
    f32x16 negm;                     // all 16 registers = minus the row's reference maximum (the first tile's): the C operand am starts from
    float p[16];                     // log2-domain scores (relative to the reference maximum) of the tile whose exponentials come next, then its probabilities / their lo parts
    unsigned phw[8], plw[8];         // P^T fragments as packed bf16 pairs: word j = keys (2j, 2j+1) of this lane's 16; words 0-3 = first 16-key half
    float psum, psum_row = 0.f;
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    auto pack2 = [&](float x, float y) __attribute__((always_inline)) {
        return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{x, y}, bf16x2_t));
    };

    // ---- the softmax as micro-steps, so that the two MFMA phases can carry them a block at a time (a wave alone on its SIMD issues one
    // instruction per ~4-cycle window: every instruction beside the MFMAs counts).
    // tail: exponentials (the scores arrive relative to the row's reference maximum: the chains start from -m), row sum, P -> bf16 hi / lo
    // words.  34 steps: pair j = steps 4j .. 4j+3.
    constexpr int kTailSteps = 34;
    auto tail_step = [&](auto t_tag) __attribute__((always_inline)) {
        constexpr int T = decltype(t_tag)::value;
        if constexpr (T < 32) {
            constexpr int J = T >> 2, U = T & 3, R = 2 * J;
            if constexpr (U == 0) {
                if constexpr (J == 0) psum = 0.f;
                // (volatile asm: as plain intrinsics hipcc hoists all sixteen exponentials -- common to both copies of the tile body -- above
                //  the branch that selects the copy, where nothing covers them; the s_nop is the trans-result hazard hipcc cannot see)
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\ts_nop 0" : "+v"(p[R]), "+v"(p[R + 1]));
            } else if constexpr (U == 1) {
                psum += p[R];
                psum += p[R + 1];
                phw[J] = pack2(p[R], p[R + 1]);
                asm volatile("" : "+v"(phw[J]));      // (the conversion stays in this slot: with any branch further down hipcc sinks it to its first use, in front of phase B's first MFMA)
            } else if constexpr (U == 2) {
                p[R] -= __uint_as_float(phw[J] << 16);
                p[R + 1] -= __uint_as_float(phw[J] & 0xffff0000u);
            } else {
                plw[J] = pack2(p[R], p[R + 1]);
                asm volatile("" : "+v"(plw[J]));
            }
        } else if constexpr (T == 32) {
            auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(psum), __float_as_uint(psum), false, false);
            psum_row = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);      // the other 16 keys of this query live in lane l ^ 32
        } else {
            l_run += psum_row;
        }
    };
    // head: scores of the next tile from the finished chains.  8 steps (+ the key mask).  MASK: the tile (first key key0) holds keys beyond
    // klen -- only ever the last tile of an utterance, whose head runs outside the blocks (a branch inside them would split the stream).
    constexpr int kHeadSteps = 9;
    auto head_step = [&](auto t_tag, auto mask_tag, int key0) __attribute__((always_inline)) {
        constexpr int T = decltype(t_tag)::value;
        constexpr bool MASK = decltype(mask_tag)::value != 0;
        if constexpr (T < 8) {
            // (asm: left to itself hipcc merges the adds of all eight steps into one burst of packed adds and moves -- 16 instructions in
            //  one place -- and packed f32 VALU beside MFMAs is slower than the scalar form anyway)
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[2 * T]) : "v"(am[2 * T]), "v"(ac[2 * T]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(p[2 * T + 1]) : "v"(am[2 * T + 1]), "v"(ac[2 * T + 1]));
        } else {
            if constexpr (MASK) {            // (only the last tile of an utterance)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + 16 * (r >> 3) + 8 * hi + (r & 7);
                    if (key >= klen) p[r] = -INFINITY;
                }
            }
        }
    };
    // row maximum of p[] (first tile only: the row's reference maximum)
    auto row_max = [&]() __attribute__((always_inline)) {
        float t0 = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), t1 = fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7]));
        float t2 = fmaxf(fmaxf(p[8], p[9]), fmaxf(p[10], p[11])), t3 = fmaxf(fmaxf(p[12], p[13]), fmaxf(p[14], p[15]));
        float tmax = fmaxf(fmaxf(t0, t1), fmaxf(t2, t3));
        auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(tmax), __float_as_uint(tmax), false, false);
        return fmaxf(__uint_as_float(rr[0]), __uint_as_float(rr[1]));
    };
    // steps [S NSTEP / NSLOT, (S + 1) NSTEP / NSLOT) of a micro-step list go into slot S of NSLOT
    auto steps_of_slot = [&](auto s_tag, auto nslot_tag, auto nstep_tag, auto&& fn) __attribute__((always_inline)) {
        constexpr int S = decltype(s_tag)::value, NSLOT = decltype(nslot_tag)::value, NSTEP = decltype(nstep_tag)::value;
        constexpr int T0 = S * NSTEP / NSLOT, T1 = (S + 1) * NSTEP / NSLOT;
        for_seq([&](auto k_tag) __attribute__((always_inline)) { fn(std::integral_constant<int, T0 + decltype(k_tag)::value>{}); },
                std::make_integer_sequence<int, T1 - T0>{});
    };

    // ---- phase A: Q.K^T of the tile in ring slot BUF -> am, ac (3 NKS MFMAs), one slot per MFMA: [fragment reads of the next k-step]
    // MFMA, this slot's share of the previous tile's softmax tail (TAIL), scheduling fence.
    // Which K piece of the DMA goes into slot S of phase A (-1: none).  The slots whose share of the softmax tail is empty or a single
    // conversion: with DK = 192 (36 slots, 34 steps: slot S carries step S - 1 up to slot 17, slot 18 none, then step S - 2) the pack
    // steps 4j + 3 sit in slots 4, 8, 12, 16 and slots 0 and 18 are empty.  (Every third slot from slot 1, the first placement, put six
    // of the twelve pieces beside a pair of exponentials: a piece holds the wave's issue port longer than an MFMA runs.)
    auto k_dma_slot = [](int S) constexpr -> int {
        if constexpr (DK == 192) return S == 0 ? 0 : S == 18 ? 5 : (S % 4 == 0 && S <= 16) ? S / 4 : -1;
        else return (S % 3 == 0 && S / 3 < PW) ? S / 3 : -1;
    };
    unsigned koff[PW];               // K source offsets of the tile to fetch (kgo, or clamped rows for a partial tile)
    gchar_t* vptr[PW];               // V^T source pointers of the tile to fetch (zero vector for 8-key groups beyond klen)
    // ---- phase A: Q.K^T of the tile in ring slot BUF -> am, ac (3 NKS MFMAs), one slot per MFMA: [fragment reads two k-steps ahead]
    // MFMA, in the slots k_dma_slot names one LDS-DMA piece of the K tile to fetch, this slot's share of the previous tile's softmax
    // tail (TAIL), scheduling fence.
    auto phase_a = [&](auto buf_tag, auto tail_tag, gchar_t* kbase_t, unsigned kdst) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool TAIL = decltype(tail_tag)::value != 0, DMA = TAIL;      // (the loop's phase A carries the tail and the DMA; the prologue's neither)
        constexpr int NSLOT = 3 * NKS;
        // (fragments requested kFA k-steps ahead of their MFMAs: a wave alone on its SIMD has nothing else to cover the LDS round trip)
        constexpr int kFA = 2;
        bf16x8_t kh[kFA + 1], kl[kFA + 1];
        auto fetch = [&](auto c_tag) __attribute__((always_inline)) {
            constexpr int C = decltype(c_tag)::value, BB = C % (kFA + 1), SH = 2 * C, SL = KSL + 2 * C;
            kh[BB] = *reinterpret_cast<const bf16x8_t*>(kp[(SH & 15) >> 1] + BUF * KB + (SH >> 4) * 256);
            kl[BB] = *reinterpret_cast<const bf16x8_t*>(kp[(SL & 15) >> 1] + BUF * KB + (SL >> 4) * 256);
        };
        for_seq([&](auto c_tag) __attribute__((always_inline)) { fetch(c_tag); }, std::make_integer_sequence<int, kFA>{});
        for_seq([&](auto s_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(s_tag)::value, C = S / 3, U = S % 3, BB = C % (kFA + 1);
            if constexpr (U == 0 && C + kFA < NKS && !(FS2_W32_ABL & 8)) fetch(std::integral_constant<int, C + kFA>{});
            // (k-step 0 starts the chains: ac from 0; am, in the loop, from -m, so that the scores arrive relative to the row's reference maximum)
            if constexpr (U == 0) mfma_s<(C == 0 ? 2 : 0)>(ac, kl[BB], qh[C], negm);
            else if constexpr (U == 1) mfma_s<(C == 0 ? (TAIL ? 1 : 2) : 0)>(am, kh[BB], qh[C], negm);
            else mfma_s<0>(ac, kh[BB], ql[C], negm);
            // DMA pieces of the tiles to fetch, one every third slot from the start of the phase (both ring slots they fill were released
            // by the barrier that opened this iteration)
            if constexpr (DMA && !(FS2_W32_ABL & (1 | 128))) {
                constexpr int Q = k_dma_slot(S);
                if constexpr (Q >= 0) dma16_so(kbase_t, koff[Q], kdst + Q * 4096);
            }
            if constexpr (TAIL && !(FS2_W32_ABL & 2))
                steps_of_slot(s_tag, std::integral_constant<int, NSLOT>{}, std::integral_constant<int, kTailSteps>{}, tail_step);
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, NSLOT>{});
    };
    // ---- phase B: O^T += V^T(tile in ring slot BUF) . P^T (6 NT MFMAs in groups of six: two n-tiles, i.e. two accumulators alternate
    // between dependent MFMAs), one slot per MFMA: [V^T fragment reads two groups ahead] MFMA, this slot's share of the next tile's
    // score sums (HEAD: from slot 6 on, i.e. six MFMAs behind the chains of phase A), scheduling fence.
    auto phase_b = [&](auto buf_tag, auto head_tag, auto dma_tag, unsigned vdst) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool HEAD = decltype(head_tag)::value != 0, DMA = decltype(dma_tag)::value != 0;
        constexpr int GPM = NT / 2;                // groups per 16-key half
        constexpr int NG = 2 * GPM, NSLOT = 6 * NG;
        constexpr int kFB = 2;                     // groups ahead of their MFMAs
        bf16x8_t vh[kFB + 1][2], vl[kFB + 1][2];
        auto fetch = [&](auto g_tag) __attribute__((always_inline)) {
            constexpr int G = decltype(g_tag)::value, M = G / GPM, N0 = (G % GPM) * 2, BB = G % (kFB + 1);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                vh[BB][u] = *reinterpret_cast<const bf16x8_t*>(vp[M] + BUF * VB + (N0 + u) * 4096);
                vl[BB][u] = *reinterpret_cast<const bf16x8_t*>(vp[2 + M] + BUF * VB + (N0 + u) * 4096);
            }
        };
        for_seq([&](auto g_tag) __attribute__((always_inline)) { fetch(g_tag); }, std::make_integer_sequence<int, (kFB < NG ? kFB : NG)>{});
        for_seq([&](auto s_tag) __attribute__((always_inline)) {
            constexpr int S = decltype(s_tag)::value, G = S / 6, U = S % 6, M = G / GPM, N0 = (G % GPM) * 2, BB = G % (kFB + 1);
            if constexpr (U == 0 && G + kFB < NG && !(FS2_W32_ABL & 8)) fetch(std::integral_constant<int, G + kFB>{});
            const bf16x8_t pf_h = __builtin_bit_cast(bf16x8_t, u32x4{phw[4 * M], phw[4 * M + 1], phw[4 * M + 2], phw[4 * M + 3]});
            const bf16x8_t pf_l = __builtin_bit_cast(bf16x8_t, u32x4{plw[4 * M], plw[4 * M + 1], plw[4 * M + 2], plw[4 * M + 3]});
            if constexpr (U == 0) { if constexpr (!(FS2_W32_ABL & 256)) mfma_o<N0, true>(vl[BB][0], pf_h); }          // (pads the VALU-written P fragment)
            else if constexpr (U == 1) { if constexpr (!(FS2_W32_ABL & 256)) mfma_o<N0 + 1, false>(vl[BB][1], pf_h); }
            else if constexpr (U == 2) mfma_o<N0, (FS2_W32_ABL & 256) != 0>(vh[BB][0], pf_l);
            else if constexpr (U == 3) mfma_o<N0 + 1, false>(vh[BB][1], pf_l);
            else if constexpr (U == 4) mfma_o<N0, false>(vh[BB][0], pf_h);
            else mfma_o<N0 + 1, false>(vh[BB][1], pf_h);
            // the V^T pieces of the DMA: odd slots from the start of the phase (the head's steps start at slot 6 and are two adds each)
            if constexpr (DMA && (S & 1) && S / 2 < PW && !(FS2_W32_ABL & (1 | 64))) dma16((const void*)vptr[S / 2], vdst + (S / 2) * 4096);
            if constexpr (HEAD && S >= 6 && !(FS2_W32_ABL & 2)) {       // (the adds of head_step are volatile asm: they stay behind the six MFMAs of the slots before)
                steps_of_slot(std::integral_constant<int, S - 6>{}, std::integral_constant<int, NSLOT - 6>{}, std::integral_constant<int, kHeadSteps>{},
                              [&](auto t_tag) __attribute__((always_inline)) { head_step(t_tag, I0{}, 0); });
            }
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, NSLOT>{});
    };
    // the source addresses of K tile kk / V^T tile kv for the DMA pieces of phase A
    auto prep_dma = [&](int kk, int kv) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PW; ++u) koff[u] = kgo[u];
        if (kk >= 0 && kk * 32 + 32 > klen) {          // partial K tile: rows clamped to the last key
            int ln = lane;
            asm volatile("" : "+v"(ln));
#pragma unroll
            for (int u = 0; u < PW; ++u) {
                int key, bir;
                k_piece(u, ln, key, bir);
                koff[u] = (unsigned)((min(kk * 32 + key, klen - 1) - kk * 32) * a.ldqk * 2 + bir);
            }
        }
        if (kv >= 0) {
            gchar_t* base = vbase + (size_t)kv * 64;
#pragma unroll
            for (int u = 0; u < PW; ++u) vptr[u] = base + vgo[u];
            if (kv * 32 + 32 > klen) {                 // partial V^T tile: 8-key groups beyond klen come from the zero vector
                int ln = lane;
                asm volatile("" : "+v"(ln));
#pragma unroll
                for (int u = 0; u < PW; ++u) {
                    int n, j, plane;
                    v_piece(u, ln, n, j, plane);
                    if (klen - (kv * 32 + 8 * j) <= 0) vptr[u] = (gchar_t*)reinterpret_cast<const char*>(g_zero16);
                }
            }
        }
    };

    // One tile: the exponentials of tile kt between the MFMAs of Q.K^T of tile kt + 1, then P.V of tile kt with the score sum / row
    // maximum of tile kt + 1 inside.  Ring of THREE slots per operand, tile t in slot t % 3 (R = kt % 3 at compile time): phase A reads
    // K slot R + 1, phase B reads V^T slot R; the DMA issued in phase A fills K slot R with tile kt + 3 (free since Q.K^T(kt) ran in the
    // previous iteration) and V^T slot R + 2 with tile kt + 2 (free since P.V(kt - 1)) and has until the end of the NEXT iteration to
    // land: the wait that closes an iteration is vmcnt(2 PW) -- everything but this iteration's own pieces.  (With a ring of two and
    // vmcnt(0) the closing wait cost 500 cycles of a 4,900-cycle tile: an LDS-DMA round trip under load is ~2,500 cycles.)
    auto tile = [&](auto r_tag, int kt) __attribute__((always_inline)) {
        constexpr int R = decltype(r_tag)::value;
        using IV = std::integral_constant<int, R>;                // V^T(kt)
        using IK = std::integral_constant<int, (R + 1) % 3>;      // K(kt + 1)
        constexpr int KD = R, VD = (R + 2) % 3;                   // DMA targets
        const bool has_next = kt + 1 < ntiles;
        // (beyond the last tile the last one is fetched again, into the free slot, so that the DMA pieces of phase A need no branches
        //  and every iteration issues the same number of them)
        const int dk = min(kt + 3, ntiles - 1), dv = min(kt + 2, ntiles - 1);
#ifdef FS2_W32_TIMING
        long long tprev = __builtin_readcyclecounter();
        wstamp[4] += 1;
#endif
        if (wave_live && !bail) {
            if (has_next) {
                prep_dma(dk, dv);
                gchar_t* kb = uniform_ptr((const void*)(kbase + (size_t)dk * 32 * a.ldqk * 2));
                asm volatile("s_nop 4" ::: "memory");      // v_readfirstlane -> SGPR base of a VMEM instruction: 5 states
                const unsigned kdst = lds0 + KD * KB + wave * 1024, vdst = lds0 + 3 * KB + VD * VB + wave * 1024;
                // Two complete copies of the tile body, chosen before phase A, so that in the common one phase A and phase B form ONE basic
                // block: a branch between them lets hipcc sink the bf16 conversions of the softmax tail out of their slots into phase B's
                // head (40 instructions nothing covers).
                if ((kt + 1) * 32 + 32 <= klen) {
                    phase_a(IK{}, I1{}, kb, kdst);
                    FS2_WT(0)
                    phase_b(IV{}, I1{}, I1{}, vdst);
                } else {            // the next tile is the utterance's last and partial: its head needs the key mask
                    phase_a(IK{}, I1{}, kb, kdst);
                    FS2_WT(0)
                    phase_b(IV{}, I0{}, I1{}, vdst);
                    for_seq([&](auto t_tag) __attribute__((always_inline)) { head_step(t_tag, I1{}, (kt + 1) * 32); }, std::make_integer_sequence<int, kHeadSteps>{});
                }
            } else {
                for_seq(tail_step, std::make_integer_sequence<int, kTailSteps>{});
                phase_b(IV{}, I0{}, I0{}, 0u);
            }
            // (lazily, after the tile went into O: a wave that leaves recomputes its rows from scratch, so what it accumulated does not matter)
            bail = __builtin_amdgcn_readfirstlane(__any(!(psum_row < kW32SumLimit)));
            if (FS2_W32_ABL & 250) bail = 0;        // (ablations that leave the sums undefined must stay on the fast path to time it)
        } else if (has_next) {
            issue_K(dk, KD);
            issue_V(dv, VD);
        }
        FS2_WT(1)
        if (has_next && !(FS2_W32_ABL & 4)) {
            // this wave's pieces of K(kt + 2) and V^T(kt + 1) -- issued one iteration ago -- have landed: only this iteration's 2 PW are in flight
            if constexpr (FS2_W32_ABL & 16) {}
            else if constexpr (2 * PW == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if constexpr (2 * PW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (2 * PW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FS2_WT(2)
            if (kt + 2 == ntiles) fix_V(kt + 1, (R + 1) % 3);      // the next tile is the last: mask its straddling key vectors (own pieces)
            if constexpr (!(FS2_W32_ABL & 32)) __syncthreads();
            FS2_WT(3)
        }
    };

    if (ntiles > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();              // K(0) landed
        FS2_WP(5)
        issue_K(min(1, ntiles - 1), 1);
        issue_K(min(2, ntiles - 1), 2);
        issue_V(0, 0);
        issue_V(min(1, ntiles - 1), 1);
        if (wave_live) {
            phase_a(I0{}, I0{}, kbase, 0u);
            mfma_drain();
            if (32 <= klen) for_seq([&](auto t_tag) __attribute__((always_inline)) { head_step(t_tag, I0{}, 0); }, std::make_integer_sequence<int, kHeadSteps>{});
            else for_seq([&](auto t_tag) __attribute__((always_inline)) { head_step(t_tag, I1{}, 0); }, std::make_integer_sequence<int, kHeadSteps>{});
            const float m0 = row_max();         // the row's reference maximum: its first tile's (every tile holds at least one key)
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] -= m0; negm[r] = -m0; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (the Q fragments are "used" here on every path: otherwise hipcc, seeing a path into the loop on which their loads might still be in
        //  flight, waits for vmcnt(0) inside the loop at their first use)
#pragma unroll
        for (int c = 0; c < NKS; ++c) asm volatile("" : : "a"(qh[c]), "a"(ql[c]));
        if (ntiles == 1) fix_V(0, 0);
        __syncthreads();              // K(1), K(2), V^T(0), V^T(1) landed; every wave is done with K(0)
        FS2_WP(6)
        for (int kt = 0; kt < ntiles; kt += 3) {
            tile(I0{}, kt);
            if (kt + 1 < ntiles) tile(I1{}, kt + 1);
            if (kt + 2 < ntiles) tile(std::integral_constant<int, 2>{}, kt + 2);
        }
    }
#ifdef FS2_W32_TIMING
    tpro = __builtin_readcyclecounter();
#endif
    const bool staged = a.ctxp != nullptr && a.ctx == nullptr;      // the model's form: the context leaves as planes only
    if (staged) {
        // every wave is done with the rings, whose memory stages the output tiles -- AND every DMA piece has landed: the pieces issued in
        // the last iteration but one (the last tile fetched once more, so that every iteration issues the same number) are waited for
        // by nobody, and one that lands after the rows below were staged overwrites them (seen as ~17 corrupted frames in one
        // utterance of c4, a different one from run to run).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (!wave_live) return;
    if (bail) {
        attn_w32_rows_slow<DK>(a, s0, len, klen, q0, h, lane);
        return;
    }

    // O^T: register r of n-tile nt = head channel 32 nt + 8 (r >> 2) + 4 hi + (r & 3) of query q0 + l31: four consecutive
    // channels per register quad.
    mfma_drain();
    const int qrow = q0 + l31;
    const float linv = (l_run > 0.f) ? 1.f / l_run : 0.f;
    const bool dead = a.mask_q && qrow >= klen;
    if (staged) {
        // The wave's 32 context rows as planes: per row the head's NT chunks of 128 B [hi 32 | lo 32] are contiguous in global memory
        // (common.h: plane_byte), so the tile is staged in LDS in exactly that image (rows 16 bytes apart from a multiple of 256: the
        // 8-byte writes of a 16-lane group land on 16 banks) and leaves as whole rows, 1 KB per store instruction.  (Stored from the
        // registers -- 8 + 8 bytes per lane and four channels, 32 rows apart per instruction -- the epilogue took 6,300 cycles.)
        constexpr int RS = DK * 4 + 16;
        char* stage = smem_w + wave * (32 * RS);
        for_seq([&](auto n_tag) __attribute__((always_inline)) {
            constexpr int nt = decltype(n_tag)::value;
            float e[16];
            read_o<nt>(e);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = f32x4{e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]} * linv;
                if (dead) v = f32x4{0.f, 0.f, 0.f, 0.f};
                uint2 vh, vl;
                split4(v, vh, vl);
                char* d = stage + l31 * RS + nt * 128 + 16 * (2 * (g & 1) + hi) + 8 * (g >> 1);      // plane_byte of channel 32 nt + 8 g + 4 hi inside its chunk
                *reinterpret_cast<uint2*>(d) = vh;
                *reinterpret_cast<uint2*>(d + 64) = vl;
            }
        }, std::make_integer_sequence<int, NT>{});
        constexpr int UPR = DK * 4 / 16;            // 16-byte units per row
        char* out = reinterpret_cast<char*>(a.ctxp) + ((size_t)(s0 + q0) * a.ctxp_chunks + (size_t)h * NT) * 128;
        const size_t row_bytes = (size_t)a.ctxp_chunks * 128;
#pragma unroll
        for (int i = 0; i < 32 * UPR / 64; ++i) {
            const int U = i * 64 + lane, r = U / UPR, ps = U - r * UPR;
            const u32x4 w = *reinterpret_cast<const u32x4*>(stage + r * RS + ps * 16);
            if (q0 + r < len) *reinterpret_cast<u32x4*>(out + (size_t)r * row_bytes + ps * 16) = w;
        }
    } else {
        if (qrow >= len) return;
        const size_t row = (size_t)(s0 + qrow);
        for_seq([&](auto n_tag) __attribute__((always_inline)) {
            constexpr int nt = decltype(n_tag)::value;
            float e[16];
            read_o<nt>(e);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = h * DK + 32 * nt + 8 * g + 4 * hi;
                f32x4 v = f32x4{e[4 * g], e[4 * g + 1], e[4 * g + 2], e[4 * g + 3]} * linv;
                if (dead) v = f32x4{0.f, 0.f, 0.f, 0.f};
                if (a.ctx) *reinterpret_cast<f32x4*>(a.ctx + row * a.ldc + col) = v;
                if (a.ctxp) store_planes4(a.ctxp, row, a.ctxp_chunks, col, v);
            }
        }, std::make_integer_sequence<int, NT>{});
    }
#ifdef FS2_W32_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const long long t_ = __builtin_readcyclecounter();
        wstamp[7] = t_ - tpro;
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
            for (int i = 0; i < 12; ++i) g_w32_phase[i] += wstamp[i];
    }
#endif
}

}  // namespace fs2
