// The "mx" arithmetic of the FFN convolution (precision mode mix_mx): TWO MFMA-equivalents per product instead of the
// three of split-bf16, at the same (parity-grade) accuracy.  Every operand is x = xh + rx with xh = fp16(x); the product is
//     a.w  =  ah.wh                      v_mfma_f32_16x16x32_f16, 16 cycles per 32 k             (exact to 2^-22 in the operands)
//          +  ra.wh + ah.rw              v_mfma_scale_f32_16x16x128_f8f6f4 on e4m3 operands, 32 cycles per 128 k each = 8 per 32 k:
//                                        the cross terms are 2^-11 of the product, e4m3 keeps 2^-4 of them -> 2^-15
//          (+ ra.rw ~ 2^-22, dropped)
// i.e. 16 + 8 + 8 = 32 MFMA cycles per 16 x 16 x 32 block of products against 48 for lo.hi + hi.lo + hi.hi on bf16.
//
// Round 2's first version of this mode took the scaled MFMA's K = 128 as 4 taps x 32 channels: its weight operand was four taps wide,
// did not fit the LDS stage and went global -> VGPR per wave, and 9 taps padded to 12 -- slower than split-bf16.  Here K = 128 is ONE tap
// x 128 channels and the whole thing is expressed in the data layout, so that the kernel IS the split-bf16 conv loop (gemm_planes.h,
// gemm_pl_bf16<.., ARITH = 2>):
//  * activation ("mx planes", common.h: store_planes4_mx): a row of C channels = 4C bytes = C/32 units of 128 B
//        [ fp16(a): C/64 units | ra8 = e4m3((a - ah) 2^(ka+11)): C/128 units | ah8 = e4m3(ah 2^ka): C/128 units ]
//    written once by the producing LayerNorm epilogue; ka is ONE exponent per tensor (e4m3 is a floating-point format: a static scale from
//    the a-priori bound |LN(x)_c| <= sqrt(D) |gamma_c| + |beta_c| loses nothing), so no scale arrays exist and no reduction runs at
//    inference time; the 2^11 on the residuals lets both cross terms share one pair of E8M0 scale bytes.
//  * weights (repack_weight_mx below): per output channel n the same unit sequence, one 128-byte row per (unit, tap):
//        [ fp16(w) | wh8 = e4m3(wh 2^kw) (meets ra8) | rw8 = e4m3((w - wh) 2^(kw+11)) (meets ah8) ],   kw from max |w|.
//  * the loop walks units x taps exactly as it walks 32-channel chunks x taps in the split-bf16 case: one A tile (BM + halo rows x 128 B)
//    per unit, one B stage (128 x 128 B) per step, both by LDS-DMA; a fragment is the two 16-byte pieces (slot lg, slot 4 + lg) of a
//    row in both cases.  Units of the first half issue 2 fp16 MFMAs per fragment pair, units of the second half 1 scaled MFMA.
//  * v_mfma_scale_f32_16x16x128_f8f6f4 operand layout (measured, tools/probes/mx_probe.hip): lane (i = l & 15, g = l >> 4) supplies row /
//    column i; registers 0-3 = k 16 g .. 16 g + 15, registers 4-7 = k 64 + 16 g ..; the scale of 32-block b comes from lane i + 16 b
//    (here one constant in every lane and byte); C/D as every 16x16 MFMA.
// Any odd kernel size up to 17 taps, C % 128 == 0, N % 128 == 0.
#pragma once
#include "gemm_planes.h"

namespace fs2 {

// bytes of the mx weight image: Npad rows x (C / 32 units) x ktaps x 128 B  (= the split-bf16 image's size)
__host__ __device__ inline size_t mx_image_bytes(int Npad, int C, int ktaps) { return (size_t)Npad * (C / 32) * ktaps * 128; }

// weights [N][ldw or C][k] fp32 -> mx image [Npad][unit][tap][128 B]; kw: exponent of the static weight scale (|w| 2^kw <= 448).
// repacked_ld > 0: the source is the library's own fp32 image [Npad][k][repacked_ld] (repack_weight: BatchNorm already folded in -- the Postnet's convolutions).
__global__ void repack_weight_mx(const float* w, int N, int C, int k, int Npad, int kw, unsigned short* out, int repacked_ld = 0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 2-byte word per thread
    const int units = C / 32;
    const int64_t total = (int64_t)Npad * units * k * 64;
    if (i >= total) return;
    const int word = (int)(i & 63);
    int64_t rest = i >> 6;
    const int tap = (int)(rest % k); rest /= k;
    const int unit = (int)(rest % units);
    const int n = (int)(rest / units);
    const int nmain = C / 64, ncorr = C / 128;
    auto wv = [&](int c) { return (n < N && c < C) ? (repacked_ld ? w[((size_t)n * k + tap) * repacked_ld + c] : w[((size_t)n * C + c) * k + tap]) : 0.f; };
    unsigned short o;
    if (unit < nmain) {
        const _Float16 h = (_Float16)wv(unit * 64 + word);
        o = __builtin_bit_cast(unsigned short, h);
    } else {
        const bool resid = unit >= nmain + ncorr;
        const int c0 = (unit - nmain - (resid ? ncorr : 0)) * 128 + 2 * word;
        float x[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float v = wv(c0 + j);
            const float wh = (float)(_Float16)v;
            x[j] = resid ? (v - wh) * exp2f((float)(kw + 11)) : wh * exp2f((float)kw);
            x[j] = fminf(fmaxf(x[j], -448.f), 448.f);
        }
        o = (unsigned short)(__builtin_amdgcn_cvt_pk_fp8_f32(x[0], x[1], 0, false) & 0xffff);
    }
    out[i] = o;
}

// ---- mx4 (gemm_planes.h: ARITH = 3; precision mode mix_mx4): the weight image with both cross terms in e2m1 and ONE scale per output channel.
// units per (n, tap): C/64 of fp16 channels, then C/128 cross units (layout: gemm_planes.h)
__host__ __device__ inline size_t mx4_image_bytes(int Npad, int C, int ktaps) { return (size_t)Npad * (C / 64 + C / 128) * ktaps * 128; }

// bytes of the weight scale image: per (N tile of 128 output channels, cross unit, tap) 1 KB = [g = 0 .. 3][n = 0 .. 127][2]: byte 0 = the E8M0 scale of the
// 16-byte slot g of that weight row (16 channels, both terms: the block lane (n, g) hands the first scaled MFMA of a cross-unit step), byte 1 = of slot 4 + g
// (the second MFMA's).  One LDS-DMA piece per step; a lane's four output channels are eight contiguous bytes.
__host__ __device__ inline size_t mx4_scale_image_bytes(int Npad, int C, int ktaps) { return (size_t)(Npad / 128) * (C / 128) * ktaps * 1024; }

// weights [N][C][k] fp32 -> mx4 image [Npad][unit][tap][128 B] + its scale image; one 4-byte word (four channels of one part) per thread.
// The scale of a 16-channel block (one 16-byte slot: four consecutive threads) follows the OCP rule from the block's own largest |fp16(w)|: a weight
// outlier costs the resolution of its 15 neighbours in one tap, not of its output channel's 3,455 other weights (round 6, simulated: tools/arith_sim_ffn_pareto.py,
// "native block": 0.1 % of the weights x 30: 4.4e-4 -> 1.3e-4 on the mel; Student-t weights 4.7e-4 -> 1.9e-4).
__global__ void repack_weight_mx4(const float* w, int N, int C, int k, int Npad, unsigned char* wsb, unsigned* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nmain = C / 64, ncross = C / 128, units = nmain + ncross;
    const int64_t total = (int64_t)Npad * units * k * 32;
    const bool live = i < total;
    const int64_t ii = live ? i : total - 1;          // (every lane takes part in the shuffles below)
    const int word = (int)(ii & 31);
    int64_t rest = ii >> 5;
    const int tap = (int)(rest % k); rest /= k;
    const int unit = (int)(rest % units);
    const int n = (int)(rest / units);
    auto wv = [&](int c) { return (n < N && c < C) ? w[((size_t)n * C + c) * k + tap] : 0.f; };
    unsigned o;
    float h[4] = {0.f, 0.f, 0.f, 0.f}, r[4] = {0.f, 0.f, 0.f, 0.f}, m = 0.f;
    if (unit >= nmain) {
        const int c = (unit - nmain) * 128 + 4 * word;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = wv(c + j);
            h[j] = (float)(_Float16)v; r[j] = v - h[j];
            m = fmaxf(m, fabsf(h[j]));
        }
    }
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));                  // the four words of a 16-byte slot: threads 4 s .. 4 s + 3 (rows are 32 words: slots never straddle a wave)
    if (unit < nmain) {          // fp16 channels 64 unit + 2 word, + 1
        const _Float16 h0 = (_Float16)wv(unit * 64 + 2 * word), h1 = (_Float16)wv(unit * 64 + 2 * word + 1);
        o = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    } else {                     // cross unit: channels c .. c + 3: bytes [wh4 wh4 | wh4 wh4 | rw4 rw4 | rw4 rw4] (common.h: store_planes4_mx4 writes the activations the same way)
        const int eb = mx4_scale_byte(m);
        const float inv = mx4_inv_scale(eb);
        o = 0;
        o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, h[0] * inv, h[1] * inv, 1.f, 0);
        o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, h[2] * inv, h[3] * inv, 1.f, 1);
        o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, r[0] * 2048.f * inv, r[1] * 2048.f * inv, 1.f, 2);
        o = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(o, r[2] * 2048.f * inv, r[3] * 2048.f * inv, 1.f, 3);
        if (live && (word & 3) == 0) {
            const int slot = word >> 2;
            wsb[((size_t)((n >> 7) * ncross + (unit - nmain)) * k + tap) * 1024 + (size_t)((slot & 3) * 128 + (n & 127)) * 2 + (slot >> 2)] = (unsigned char)eb;
        }
    }
    if (live) out[i] = o;
}

}  // namespace fs2
