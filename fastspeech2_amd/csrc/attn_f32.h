// fp32 flash-style self-attention over packed variable-length sequences, on the f32-input matrix core.
// Replaces reference core/attention.py:55-70 (scores [B,H,T,T] materialised in HBM, masked_fill, softmax,
// matmul) with one kernel that never writes scores: per (utterance, head, 64-query tile) a workgroup
// streams 32-key K/V tiles through LDS and keeps the online-softmax state in registers.
//
// Wave w of the workgroup owns 16 query rows.  It computes the TRANSPOSED score tile S^T = K . Q^T
// (MFMA A operand = K rows from LDS, B operand = Q^T from registers), so that lane l ends up holding, for
// query q = l&15, the scores of keys 4*(l>>4)+reg of each 16-key sub-tile.  That is exactly the A-operand
// layout P[q = l&15][k = l>>4] of the following P.V MFMAs, so P never leaves registers and needs no
// transpose; row max / row sum are an in-lane reduce plus two xor-shuffles (lanes l, l^16, l^32, l^48).
//   QK^T :  st[t] += mfma(K[key=16t+(l&15)][k], Q[q=l&15][k])            k = 16c + 4*(l>>4) + s
//   P.V  :  O[nt] += mfma(p[t][s], V[key=16t+4*(l>>4)+s][n=16nt+(l&15)])
// O tiles are C-layout (col n = l&15, row q = 4*(l>>4)+reg); the per-query rescale factors are fetched
// from the lane that owns that query's statistics with a ds_bpermute (__shfl).
#pragma once
#include "common.h"
#include "gemm_f32.h"

namespace fs2 {

constexpr int kAttKT = 32;   // keys per LDS tile

struct AttnArgs {
    const float* qkv; int ld;        // [R, 3D]: q at col h*dk, k at D + h*dk, v at 2D + h*dk
    float* ctx; int ldc;             // [R, D]
    const int* start; const int* len; const int* klen;   // per utterance
    const int2* work;                // (utterance, query tile)
    const int* nwork;                // device-driven layout: number of valid work items (the grid is a capacity), or nullptr
    int nitems;                      // host-driven layout: number of work items
    int D; int mask_q;               // mask_q: query rows >= klen yield zeros (reference masked_fill(0))
    float scale;                     // 1/sqrt(dk)
};

template <int DK>
constexpr size_t attn_lds_bytes() { return (size_t)2 * kAttKT * (DK + 4) * sizeof(float); }

template <int DK>
__global__ __launch_bounds__(256) void attn_f32(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDK = DK + 4;          // == 4 (mod 8): conflict-free ds_read_b32 of V, 16-B aligned rows
    constexpr int NC = DK / 16;          // 16-wide k chunks of the head dim == 16-wide n tiles of O
    float* Ks = smem;
    float* Vs = smem + kAttKT * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    int b, qb;
    if (!att_item64(a.work, a.nwork, a.nitems, b, qb)) return;
    const int h = blockIdx.y;
    const int s0 = __builtin_amdgcn_readfirstlane(a.start[b]), len = __builtin_amdgcn_readfirstlane(a.len[b]), klen = __builtin_amdgcn_readfirstlane(a.klen[b]);
    if (qb >= len) return;                     // second half of an utterance's last 128-query item
    const int q0 = qb + wave * 16;
    const float* qbase = a.qkv + (size_t)h * DK;
    const float* kbase = a.qkv + a.D + (size_t)h * DK;
    const float* vbase = a.qkv + 2 * a.D + (size_t)h * DK;

    // Q fragments for query row q0 + lr: NC float4 at k = 16c + 4*lg
    f32x4 qf[NC];
    {
        const int qrow = q0 + lr;
        const bool ok = qrow < len;
        const float* qp = qbase + (size_t)(s0 + (ok ? qrow : 0)) * a.ld + lg * 4;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            qf[c] = ok ? *reinterpret_cast<const f32x4*>(qp + c * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 o[NC];
#pragma unroll
    for (int n = 0; n < NC; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;   // statistics of query q0 + lr (replicated over lg)

    const int ntiles = (klen + kAttKT - 1) / kAttKT;
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * kAttKT;
        __syncthreads();
        {   // all loads of the tile are issued before the first LDS write (no branch between them)
            constexpr int NLD = kAttKT * (DK / 4) / 256;
            f32x4 kreg[NLD], vreg[NLD];
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int idx = tid + u * 256;
                const int r = idx / (DK / 4), c4 = idx - r * (DK / 4);
                const int key = min(key0 + r, klen - 1);           // clamp: rows >= klen are masked below
                const size_t off = (size_t)(s0 + key) * a.ld + c4 * 4;
                kreg[u] = *reinterpret_cast<const f32x4*>(kbase + off);
                vreg[u] = *reinterpret_cast<const f32x4*>(vbase + off);
            }
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int idx = tid + u * 256;
                const int r = idx / (DK / 4), c4 = idx - r * (DK / 4);
                const bool ok = key0 + r < klen;
                const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(Ks + r * LDK + c4 * 4) = ok ? kreg[u] : z;
                *reinterpret_cast<f32x4*>(Vs + r * LDK + c4 * 4) = ok ? vreg[u] : z;
            }
        }
        __syncthreads();

        // ---- S^T = K . Q^T for two 16-key sub-tiles ----
        f32x4 st[2];
        st[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        st[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const f32x4 k0 = *reinterpret_cast<const f32x4*>(Ks + lr * LDK + c * 16 + lg * 4);
            const f32x4 k1 = *reinterpret_cast<const f32x4*>(Ks + (16 + lr) * LDK + c * 16 + lg * 4);
            st[0] = mfma16(k0.x, qf[c].x, st[0]);
            st[1] = mfma16(k1.x, qf[c].x, st[1]);
            st[0] = mfma16(k0.y, qf[c].y, st[0]);
            st[1] = mfma16(k1.y, qf[c].y, st[1]);
            st[0] = mfma16(k0.z, qf[c].z, st[0]);
            st[1] = mfma16(k1.z, qf[c].z, st[1]);
            st[0] = mfma16(k0.w, qf[c].w, st[0]);
            st[1] = mfma16(k1.w, qf[c].w, st[1]);
        }
        // st[t][r] = q(l&15) . key(key0 + 16t + 4lg + r)
        float p[2][4];
        float tmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = key0 + t * 16 + lg * 4 + r;
                const float s = (key < klen) ? st[t][r] * a.scale : -INFINITY;
                p[t][r] = s;
                tmax = fmaxf(tmax, s);
            }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);          // finite: every tile holds >= 1 valid key
        const float alpha = expf(m_run - m_new);         // exp(-inf) = 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(p[t][r] - m_new);
                p[t][r] = e;
                psum += e;
            }
        psum += __shfl_xor(psum, 16);
        psum += __shfl_xor(psum, 32);
        l_run = l_run * alpha + psum;
        m_run = m_new;
        // rescale O: row q = 4*lg + r of the C layout <- alpha held by lane (4*lg + r)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ar = __shfl(alpha, lg * 4 + r);
#pragma unroll
            for (int n = 0; n < NC; ++n) o[n][r] *= ar;
        }
        // ---- O += P . V ----
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* vrow = Vs + (t * 16 + lg * 4 + s) * LDK + lr;
#pragma unroll
                for (int n = 0; n < NC; ++n) o[n] = mfma16(p[t][s], vrow[n * 16], o[n]);
            }
    }
    // ---- finalize: O / l, store ctx rows q0 + 4*lg + r ----
    const float linv = (l_run > 0.f) ? 1.f / l_run : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float lr_inv = __shfl(linv, lg * 4 + r);
        const int qrow = q0 + lg * 4 + r;
        if (qrow >= len) continue;
        const bool dead = a.mask_q && qrow >= klen;
        float* dst = a.ctx + (size_t)(s0 + qrow) * a.ldc + (size_t)h * DK + lr;
#pragma unroll
        for (int n = 0; n < NC; ++n) dst[n * 16] = dead ? 0.f : o[n][r] * lr_inv;
    }
}

}  // namespace fs2
