// gemm_mx_conv9: the 9-tap FFN convolution with TWO MFMA-equivalents per product at parity-grade accuracy (arithmetic mode "mx",
// DESIGN.md section 3).  Every operand is x = xh + rx with xh = fp16(x); the product is
//     a.w  =  ah.wh                      one v_mfma_f32_16x16x32_f16 per 32 k               (exact to 2^-22 in the operands)
//          +  ra.wh + ah.rw              two v_mfma_scale_f32_16x16x128_f8f6f4 per 128 k    (fp8 e4m3 operands: 2^-11 x 2^-4 = 2^-15)
//          (+ ra.rw ~ 2^-22, dropped)
// The block-scaled fp8 MFMA runs at 2.25x the 16-bit rate and takes K = 128 per instruction, so the correction terms cost 0.89 of a
// 16-bit MFMA per 32 k in the ideal case and 1.19 here (9 taps = 4 + 4 + 1: the third instruction of a chunk is a quarter full):
// 2.2 MFMA-equivalents per product instead of 3, rms error 5e-6 per GEMM (split-bf16: 2e-6; two-term fp16: 1e-4).
//
// Operands:
//  * A planes in the "f16mx" format (common.h: store_planes4_mx), 128 B per (row, 32-channel chunk), the LDS row image:
//      [ ah: 32 fp16 in kperm order | ah8: 32 e4m3 = fp8(a 2^ka) in channel order | ra8: 32 e4m3 = fp8((a - ah) 2^(ka+11)) ]
//    ka is ONE exponent per tensor (fp8 is floating point: a static scale from an a-priori bound of the LayerNorm output loses no
//    precision), so no scale arrays exist: the MFMA's block scales are two constants.
//  * main-term weights: the fp16 image of gemm_bf16.h (hi half only is read), LDS-DMA'd per (chunk, tap) as in gemm_pl_bf16.
//  * correction weights W8: fp8 images wh8 = fp8(wh 2^kw), rw8 = fp8((w - wh) 2^(kw+11)) laid out so that ONE coalesced 1-KB load gives a
//    wave its B operand half: [N/128][chunk][tap group 3][wn 2][nt 4][term 2][half 2][lane 64][16 B]; they go global -> VGPR (a K = 128
//    step needs four taps of B at once: 64 KB per step through LDS would not fit two workgroups per CU).
//  * v_mfma_scale_f32_16x16x128_f8f6f4 layout (measured, tools/probes/mx_probe.hip): lane (i = l & 15, g = l >> 4) supplies row i;
//    registers 0-3 = the 16 k-values [16 (g & 1), +16) of 32-block (g >> 1), registers 4-7 = the same of block 2 + (g >> 1); the scale
//    of block b comes from lane i + 16 b (here: the same constant everywhere, replicated over the four bytes); C/D as every 16x16 MFMA.
//    Block b of a correction instruction = tap 4 tg + b of the current chunk (A rows shifted by the tap, as in the main loop).
#pragma once
#include "gemm_planes.h"

namespace fs2 {

typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));

constexpr int kMxTaps = 9, kMxGroups = 3;      // taps 0-3, 4-7, 8 (+ three empty blocks)

// bytes of the W8 image for N outputs and nchunks chunks
__host__ __device__ inline size_t mx_w8_bytes(int Npad, int nchunks) { return (size_t)(Npad / 128) * nchunks * kMxGroups * 2 * 4 * 2 * 2 * 1024; }

// weights [N][C][9] fp32 -> W8 image; kw: exponent of the static weight scale (|w| 2^kw <= 448)
__global__ void repack_weight_mx8(const float* w, int N, int C, int Npad, int nchunks, int kw, unsigned char* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // one byte per thread
    const int64_t total = (int64_t)mx_w8_bytes(Npad, nchunks);
    if (i >= total) return;
    const int byte = (int)(i & 15), lane = (int)((i >> 4) & 63), half = (int)((i >> 10) & 1), term = (int)((i >> 11) & 1);
    const int nt = (int)((i >> 12) & 3), wn = (int)((i >> 14) & 1);
    int64_t rest = i >> 15;
    const int tg = (int)(rest % kMxGroups); rest /= kMxGroups;
    const int chunk = (int)(rest % nchunks);
    const int ntile = (int)(rest / nchunks);
    const int lr = lane & 15, g = lane >> 4;
    const int n = ntile * 128 + wn * 64 + 4 * lr + nt;                       // the output channel acc[.][nt] of lane lr belongs to
    const int tap = 4 * tg + 2 * half + (g >> 1);
    const int c = chunk * 32 + 16 * (g & 1) + byte;
    float v = 0.f;
    if (n < N && c < C && tap < kMxTaps) v = w[((size_t)n * C + c) * kMxTaps + tap];
    const float wh = (float)(_Float16)v;
    const float x = term == 0 ? wh * exp2f((float)kw) : (v - wh) * exp2f((float)(kw + 11));
    const int packed = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(x, -448.f), 448.f), 0.f, 0, false);
    out[i] = (unsigned char)(packed & 0xff);
}

template <int BM>
__global__ __launch_bounds__(256, 2) void gemm_mx_conv9(GemmArgs a) {
    constexpr int MT = BM / 32;
    constexpr int AROWS = BM + kMaxHalo;
    constexpr int ktaps = kMxTaps, P = 4;
    extern __shared__ __attribute__((aligned(16))) char smem_m[];
    char* As0 = smem_m;
    char* Bs0 = smem_m + AROWS * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * kB16BN, m0 = blockIdx.y * BM;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;
    const int lr = lane & 15, lg = lane >> 4;
    const int lp = rperm(lr);
    const __bf16* Wb = reinterpret_cast<const __bf16*>(a.W);
    const __bf16* Xp = reinterpret_cast<const __bf16*>(a.Xp);
    const int nchunks = a.Cpad / 32;
    const int niter = nchunks * ktaps;
    const int jrow = lane >> 3, jslot = lane & 7;
    const int a_instr = (BM + 2 * P + 7) >> 3;
    const int sA = jslot ^ (jrow >> 1) ^ ((wave & 1) << 2);
    const int arow0 = m0 - P + wave * 8 + jrow;
    const __bf16* a_src0 = Xp + (ptrdiff_t)arow0 * nchunks * 64 + sA * 8;
    const size_t a_qstride = (size_t)32 * nchunks * 64;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_void_t*)smem_m);
    const unsigned ldsA = lds0 + wave * 1024, ldsB = lds0 + AROWS * 128 + wave * 1024;
    auto dma_A = [&](int ch) {
        unsigned dst = ldsA;
        const __bf16* src = a_src0 + (size_t)ch * 64;
        int row = arow0;
        for (int q = wave; q < a_instr; q += 4) {
            const bool ok = row >= 0 && row < a.R;
            const void* sp = ok ? static_cast<const void*>(src) : static_cast<const void*>(g_zero16);
            dma16(sp, dst);
            dst += 4096; src += a_qstride; row += 32;
        }
    };
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
    // correction weights: this wave's slice of the W8 image; unit (tg, nt) = 4 KB = [term 2][half 2][lane 64][16 B]
    const char* w8 = reinterpret_cast<const char*>(a.W8) + ((size_t)blockIdx.x * nchunks * kMxGroups * 2 + wn) * (size_t)(4 * 4096) + lane * 16;
    const size_t w8_tg = (size_t)2 * 4 * 4096, w8_chunk = (size_t)kMxGroups * w8_tg;      // strides: tap group (both wn), chunk
    const int scale = a.mx_scale;                       // the E8M0 byte of the A side replicated x4; the B side sits in mx_scale_b
    const int scale_b = a.mx_scale_b;

    dma_A(0);
    dma_B(0, 0);
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
                v += load4_or_zero(a.resid + (size_t)row * a.ldr + col, a.resid != nullptr && row < a.R && col < a.N);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt][r] = v[nt];
            }
    }
    // one correction unit = tap group tg x n-tile nt: 4 coalesced 16-byte loads per lane (term x register half), then 2 MT scaled MFMAs.
    // (Named registers, no arrays of vectors / references: hipcc keeps those in scratch memory.)
#define FS2_MX_LOAD(P_, X_) \
    const v4i_t X_##00 = *reinterpret_cast<const v4i_t*>((P_)); const v4i_t X_##01 = *reinterpret_cast<const v4i_t*>((P_) + 1024); \
    const v4i_t X_##10 = *reinterpret_cast<const v4i_t*>((P_) + 2048); const v4i_t X_##11 = *reinterpret_cast<const v4i_t*>((P_) + 3072);
#define FS2_MX_TERM(NT_, SLOT_, H0_, H1_) { \
        const v8i_t bv = {H0_[0], H0_[1], H0_[2], H0_[3], H1_[0], H1_[1], H1_[2], H1_[3]}; \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) { \
            const int r = wm * (BM / 2) + mt * 16 + lp; \
            const v4i_t x0 = *reinterpret_cast<const v4i_t*>(As0 + swz(r + t0, (SLOT_) + (lg & 1))); \
            const v4i_t x1 = *reinterpret_cast<const v4i_t*>(As0 + swz(r + t1, (SLOT_) + (lg & 1))); \
            const v8i_t av = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]}; \
            acc[mt][NT_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc[mt][NT_], 0, 0, 0, scale, 0, scale_b); \
            if (MT > 4 || (mt & 1) == 1) asm volatile("" ::: "memory");      /* bound the A fragments in flight (registers) */ \
        } }
#define FS2_MX_UNIT(NT_, X_) FS2_MX_TERM(NT_, 6, X_##00, X_##01) FS2_MX_TERM(NT_, 4, X_##10, X_##11)      /* term 0: ra8 . wh8, term 1: ah8 . rw8 */
    // The twelve correction units of a chunk (3 tap groups x 4 n-tiles) are spread over its nine k-steps: step `tap` handles unit `tap`, steps
    // 0-2 also units 9-11.  A unit's weights are requested right after the step's LDS-DMA issue and consumed after the step's main MFMAs
    // (an L2 round trip, ~1 us, against a ~2 us step), its A operands come from the resident tile.  The tap loop is unrolled so that the n-tile
    // of a unit is a compile-time accumulator index; the compiler barriers keep hipcc from hoisting the loads of later steps (spills).
    // (tg passes through an opaque register: otherwise hipcc hoists the ~80 distinct LDS addresses of a chunk out of the loop and spills)
#define FS2_MX_T(U_) int tgv = (U_) >> 2; asm volatile("" : "+v"(tgv)); \
    const int t0 = min(4 * tgv + (lg >> 1), ktaps - 1), t1 = min(4 * tgv + 2 + (lg >> 1), ktaps - 1);
    int it = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const char* wc = w8 + (size_t)chunk * w8_chunk;
#pragma unroll
        for (int tap = 0; tap < ktaps; ++tap, ++it) {
            dma_barrier();
            if (it + 1 < niter) dma_B(it + 1, (it + 1) & 1);
            FS2_MX_LOAD(wc + (size_t)(tap >> 2) * w8_tg + (size_t)(tap & 3) * 4096, ua)
            FS2_MX_LOAD(wc + (size_t)2 * w8_tg + (size_t)((tap < 3 ? tap + 1 : 0)) * 4096, ub)      // (used by steps 0-2 only; dead code otherwise)
            const char* Bs = Bs0 + (it & 1) * (kB16BN * 128);
            bf16x8_t bh[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) bh[nt] = *reinterpret_cast<const bf16x8_t*>(Bs + swz(wn * 64 + nt * 16 + lp, lg));
            if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(1);
            int tapv = tap;
            asm volatile("" : "+v"(tapv));
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int r = wm * (BM / 2) + mt * 16 + lp + tapv;
                const bf16x8_t ah = *reinterpret_cast<const bf16x8_t*>(As0 + swz(r, lg));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = mfma16<true>(ah, bh[nt], acc[mt][nt]);
            }
            asm volatile("" ::: "memory");
            { FS2_MX_T(tap) FS2_MX_UNIT(tap & 3, ua) }
            if (tap < 3) { FS2_MX_T(9 + tap) FS2_MX_UNIT((9 + tap) & 3, ub) }
            if (FS2_SETPRIO) __builtin_amdgcn_s_setprio(0);
            asm volatile("" ::: "memory");
        }
        if (chunk + 1 < nchunks) {
            __syncthreads();              // every wave has read its last fragments of this chunk's A tile
            dma_A(chunk + 1);
        }
    }
#undef FS2_MX_T
#undef FS2_MX_LOAD
#undef FS2_MX_TERM
#undef FS2_MX_UNIT
    pl_epilogue<MT>(a, acc, m0 + wm * (BM / 2), col, lg);
}

}  // namespace fs2
