// fp32 conv-as-GEMM kernels on the f32-input matrix core (v_mfma_f32_16x16x4_f32, exact fp32:
// bit-for-bit a k-ordered fmaf chain, 157 TFLOP/s peak on MI355X).
//
// Both kernels compute  Y[m, n] = epi( sum_{tap, c} X[m + tap - P, c] * W[n, tap, c] )  over the gapped
// packed row layout (common.h), i.e. Linear (ktaps = 1) and "same" Conv1d along time (ktaps = 3/5/9) with
// the zero padding supplied by the gap rows.  Replaces torch.nn.Linear / Conv1d call sites of the
// reference: core/attention.py:48-50,71-74, core/modules.py:247-248, core/encoder.py:118-125,
// core/variance_predictor.py:46-58, core/modules.py:350-359, fastspeech.py:228-230.
//
// LDS staging: an A tile of (BM + ktaps - 1) rows x 32 channels is loaded once per channel chunk and
// re-used by all taps (each tap reads it shifted by one row); the B tile [BN][32] of the current
// (tap, chunk) is register-prefetched one iteration ahead so its L2/HBM latency hides under the MFMAs.
// MFMA operand mapping (16x16x4): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; one
// ds_read_b128 at k-offset 4*(l>>4) feeds four consecutive MFMAs (k index = 4*(l>>4)+s for step s, the
// same permutation on A and B, which leaves the sum unchanged).  C/D: col = l&15, row = 4*(l>>4)+reg.
#pragma once
#include "common.h"

namespace fs2 {

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Stage rows [row0, row0+nrows) x channels [c0, c0+32) of X into As[nrows][kLd]; out-of-range -> 0.
__device__ __forceinline__ void stage_A(float* As, const GemmArgs& a, int row0, int nrows, int c0, int tid,
                                        int nthreads) {
    for (int idx = tid; idx < nrows * (kBK / 4); idx += nthreads) {
        const int r = idx >> 3, kq = idx & 7;
        const int g = row0 + r, c = c0 + kq * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g >= 0 && g < a.R && c < a.C) v = *reinterpret_cast<const float4*>(a.X + (size_t)g * a.ldx + c);
        *reinterpret_cast<float4*>(As + r * kLd + kq * 4) = v;
    }
}

// ---------------------------------------------------------------------------------------------------
// Kernel "tile": 128 x 128 output tile, 4 waves as 2 x 2, each wave 64 x 64 (4 x 4 MFMA tiles).
// Elementwise epilogue only (bias, residual, activation).  Used for QKV, the k=9 FFN conv (+ReLU) and
// the Postnet 256->256 convs (+tanh).
// ---------------------------------------------------------------------------------------------------
constexpr int kTileBM = 128, kTileBN = 128;
constexpr size_t kTileLds = (size_t)((kTileBM + kMaxHalo) + kTileBN) * kLd * sizeof(float);

__global__ __launch_bounds__(256) void gemm_tile_f32(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Bs = smem + (kTileBM + kMaxHalo) * kLd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * kTileBM, n0 = blockIdx.y * kTileBN;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int P = (a.ktaps - 1) >> 1;
    const int lr = lane & 15, lg = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.Cpad / kBK;
    const int niter = nchunks * a.ktaps;
    constexpr int BREG = kTileBN * (kBK / 4) / 256;   // 4 float4 per thread
    f32x4 breg[BREG];    // native vector type: stays in registers (common.h note)
#define FS2_GLOAD_B(c_, t_)                                                                                  \
    {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < BREG; ++i) {                                                   \
            const int idx = tid + i * 256;                                                                   \
            breg[i] = *reinterpret_cast<const f32x4*>(a.W + ((size_t)(n0 + (idx >> 3)) * a.ktaps + (t_)) * a.Cpad + \
                                                       (c_) * kBK + (idx & 7) * 4);                          \
        }                                                                                                    \
    }
    FS2_GLOAD_B(0, 0)
    int it = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk)
    for (int tap = 0; tap < a.ktaps; ++tap, ++it) {       // nested loops: no integer division per k-step
        __syncthreads();   // everyone finished reading As/Bs of the previous iteration
        if (tap == 0) stage_A(As, a, m0 - P, kTileBM + 2 * P, chunk * kBK, tid, 256);
#pragma unroll
        for (int i = 0; i < BREG; ++i) {
            const int idx = tid + i * 256;
            *reinterpret_cast<f32x4*>(Bs + (idx >> 3) * kLd + (idx & 7) * 4) = breg[i];
        }
        __syncthreads();
        if (it + 1 < niter) {
            if (tap + 1 < a.ktaps) FS2_GLOAD_B(chunk, tap + 1) else FS2_GLOAD_B(chunk + 1, 0)
        }
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            f32x4 af[4], bf[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                af[mt] = *reinterpret_cast<const f32x4*>(As + (wm * 64 + mt * 16 + lr + tap) * kLd + kk * 16 + lg * 4);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                bf[nt] = *reinterpret_cast<const f32x4*>(Bs + (wn * 64 + nt * 16 + lr) * kLd + kk * 16 + lg * 4);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[mt][nt] = mfma16(af[mt].x, bf[nt].x, acc[mt][nt]);
                    acc[mt][nt] = mfma16(af[mt].y, bf[nt].y, acc[mt][nt]);
                    acc[mt][nt] = mfma16(af[mt].z, bf[nt].z, acc[mt][nt]);
                    acc[mt][nt] = mfma16(af[mt].w, bf[nt].w, acc[mt][nt]);
                }
        }
    }
#undef FS2_GLOAD_B
    // epilogue: row = 4*lg + reg, col = lr inside each 16x16 tile
    tile_epilogue_64x64<4>(a, acc, m0 + wm * 64, n0 + wn * 64, lr, lg, a.relu_pre != 0);
}

// ---------------------------------------------------------------------------------------------------
// Kernel "rows": every wave owns 16 complete output rows (all N = 16*NT columns), so LayerNorm and the
// scalar-head dot product are wave-local: in-lane sums over the NT tiles + a 16-lane xor-shuffle tree.
// Block = 4 waves = 64 rows.  Used for out-proj+residual+LN, FFN w_2+residual+LN, the predictor conv
// stacks (+ReLU+LN(+Linear(.,1))), the decoder input layer (Linear+LN+ReLU+PE), feat_out and the last
// Postnet conv (+residual).
// ---------------------------------------------------------------------------------------------------
constexpr int kRowsBM = 64;
template <int NT>
constexpr size_t rows_lds_bytes() {
    return (size_t)((kRowsBM + kMaxHalo) + NT * 16) * kLd * sizeof(float);
}

template <int NT>
__global__ __launch_bounds__(256) void gemm_rows_f32(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = NT * 16;
    float* As = smem;
    float* Bs = smem + (kRowsBM + kMaxHalo) * kLd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * kRowsBM;
    if (a.Rp != nullptr && m0 >= ((*a.Rp + 127) & ~127)) return;      // device-driven layout: tile beyond the rows in use
    const int P = (a.ktaps - 1) >> 1;
    const int lr = lane & 15, lg = lane >> 4;

    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.Cpad / kBK;
    const int niter = nchunks * a.ktaps;
    constexpr int NB4 = BN * (kBK / 4);                 // float4 per B tile
    constexpr int BREG = (NB4 + 255) / 256;
    f32x4 breg[BREG];
#define FS2_GLOAD_B(c_, t_)                                                                                  \
    {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < BREG; ++i) {                                                   \
            const int idx = tid + i * 256;                                                                   \
            if (idx < NB4)                                                                                   \
                breg[i] = *reinterpret_cast<const f32x4*>(a.W + ((size_t)(idx >> 3) * a.ktaps + (t_)) * a.Cpad + (c_) * kBK + (idx & 7) * 4); \
        }                                                                                                    \
    }
    FS2_GLOAD_B(0, 0)
    int it = 0;
    for (int chunk = 0; chunk < nchunks; ++chunk)
    for (int tap = 0; tap < a.ktaps; ++tap, ++it) {
        __syncthreads();
        if (tap == 0) stage_A(As, a, m0 - P, kRowsBM + 2 * P, chunk * kBK, tid, 256);
#pragma unroll
        for (int i = 0; i < BREG; ++i) {
            const int idx = tid + i * 256;
            if (idx < NB4) *reinterpret_cast<f32x4*>(Bs + (idx >> 3) * kLd + (idx & 7) * 4) = breg[i];
        }
        __syncthreads();
        if (it + 1 < niter) {
            if (tap + 1 < a.ktaps) FS2_GLOAD_B(chunk, tap + 1) else FS2_GLOAD_B(chunk + 1, 0)
        }
#pragma unroll
        for (int kk = 0; kk < kBK / 16; ++kk) {
            const float4 af = *reinterpret_cast<const float4*>(As + (wave * 16 + lr + tap) * kLd + kk * 16 + lg * 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float4 bf = *reinterpret_cast<const float4*>(Bs + (nt * 16 + lr) * kLd + kk * 16 + lg * 4);
                acc[nt] = mfma16(af.x, bf.x, acc[nt]);
                acc[nt] = mfma16(af.y, bf.y, acc[nt]);
                acc[nt] = mfma16(af.z, bf.z, acc[nt]);
                acc[nt] = mfma16(af.w, bf.w, acc[nt]);
            }
        }
    }

#undef FS2_GLOAD_B
    // ---- epilogue: each (lg, r) pair is one output row spread over 16 lanes x NT tiles ----
    const float inv_n = 1.f / (float)a.N;
    const float alpha = (a.pe && a.pe_alpha) ? a.pe_alpha[0] : 1.f;
    const float dotb = (a.dot_w && a.dot_b) ? a.dot_b[0] : 0.f;
    const float* __restrict__ biasp = a.bias;
    const float* __restrict__ residp = a.resid;
    const int* __restrict__ rposp = a.row_pos;
    int posr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {       // row flags of all four rows up front (one latency, not four)
        const int row = m0 + wave * 16 + lg * 4 + r;
        posr[r] = (row < a.R) ? (rposp ? rposp[row] : 0) : -1;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + wave * 16 + lg * 4 + r;
        const bool inb = row < a.R;
        const int pos = posr[r];
        const bool valid = pos >= 0;
        float v[NT];
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 16 + lr;
            float t = acc[nt][r];
            if (biasp) t += biasp[col];
            if (residp && inb) t += residp[(size_t)row * a.ldr + col];
            if (a.relu_pre) t = fmaxf(t, 0.f);
            v[nt] = t;
            s += t;
        }
        if (a.ln_g) {
            const float mean = wave16_sum(s) * inv_n;
            float q = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float d = v[nt] - mean;
                q += d * d;
            }
            const float rstd = 1.f / sqrtf(wave16_sum(q) * inv_n + a.ln_eps);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 16 + lr;
                v[nt] = (v[nt] - mean) * rstd * a.ln_g[col] + a.ln_b[col];
            }
        }
        float dsum = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = nt * 16 + lr;
            float t = apply_act(v[nt], a.act_post);
            if (a.pe && valid) t = t * a.x_scale + alpha * a.pe[(size_t)pos * a.pe_ld + col];
            if (a.dot_w) dsum += t * a.dot_w[col];
            if (a.Y && inb) a.Y[(size_t)row * a.ldy + col] = valid ? t : 0.f;
        }
        if (a.dot_w) {
            dsum = wave16_sum(dsum);
            if (lr == 0 && inb) a.dot_out[row] = valid ? dsum + dotb : 0.f;
        }
    }
}

}  // namespace fs2
