// Split-bf16 ("bf16x3") / plain-bf16 flash-style self-attention on v_mfma_f32_16x16x32_bf16.
// Same algorithm and register choreography as attn_f32.h (transposed scores S^T = K.Q^T so that P is born in
// the A-operand layout of P.V, online softmax in registers), with every operand split x = hi + lo (bf16) and
// three MFMAs per fragment pair (lo*hi + hi*lo + hi*hi, fp32 accumulate).  Replaces reference
// core/attention.py:55-70.
//
// Two kernels:
//  * qkv_split   : one pass over the fp32 QKV projection [R, 3D] -> bf16 hi/lo planes  QK [R, 2D]  (Q pre-scaled
//                  by 1/sqrt(d_k), row-major) and  V^T [D, Rvt]  (key index contiguous), because the MFMA wants
//                  the contraction index contiguous per lane: d for Q.K^T, the key for P.V.
//  * attn_bf16   : per (utterance, head, 64-query tile) streams 32-key tiles of K and V^T through LDS.
//
// MFMA 16x16x32 operands: lane l supplies A[i = l&15][k = 8g..8g+7] and B[k = 8g..8g+7][j = l&15], g = l>>4;
// C/D: col = l&15, row = 4g + reg.
//   S^T sub-tile t (16 keys x 16 queries): A = K rows, B = Q rows.  Row i of sub-tile t is loaded from key
//   8*(i>>2) + 4t + (i&3), so that after both sub-tiles lane (q, g) holds the scores of keys 8g..8g+7 of the
//   32-key tile in order -- exactly the A-operand k-slots of the P.V MFMA against V^T stored in natural key order.
// LDS: K tile rows = keys, [hi d_k | lo d_k] bf16 per row, 16-B slot index XOR-swizzled with the row's position
// i inside its sub-tile (rows are a multiple of 256 B apart, so unswizzled every lane of a ds_read_b128 group
// would hit the same bank slot); V^T tile rows = head-dim n, [hi 32 keys | lo 32 keys] = 128 B, swizzled like
// the GEMM tiles (gemm_bf16.h swz()).
#pragma once
#include "common.h"
#include "gemm_bf16.h"

namespace fs2 {

constexpr int kAttAlign = 8;    // utterance starts are multiples of this many rows: a 16-byte V^T load is 8 consecutive keys (rows).
                                // (Round 1 used 32: an average of 12 more dead rows per utterance in every row-proportional kernel, 2.6 % at c3.)

struct QkvSplitArgs {
    const float* qkv; int R; int Rvt; int D; int dk; float scale;
    __bf16 *qk_hi, *qk_lo;     // [R][2D]
    __bf16 *vt_hi, *vt_lo;     // [D][Rvt]
};

// grid.x = Rvt / 32 row tiles; 256 threads.
__global__ __launch_bounds__(256) void qkv_split(QkvSplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float vt_s[];   // [32][D + 1]
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * 32;
    const int D = a.D, ld = 3 * D;
    // Q | K : 8 consecutive columns per item, row-major copy with split
    const int groups = 2 * D / 8;
    for (int idx = tid; idx < 32 * groups; idx += 256) {
        const int r = idx / groups, c = (idx - r * groups) * 8;
        const int row = r0 + r;
        if (row >= a.R) continue;
        const float* src = a.qkv + (size_t)row * ld + c;
        float4 p = *reinterpret_cast<const float4*>(src), q = *reinterpret_cast<const float4*>(src + 4);
        if (c < D) {
            p.x *= a.scale; p.y *= a.scale; p.z *= a.scale; p.w *= a.scale;
            q.x *= a.scale; q.y *= a.scale; q.z *= a.scale; q.w *= a.scale;
        }
        const SplitPair s = split8(p, q);
        *reinterpret_cast<uint4*>(a.qk_hi + (size_t)row * 2 * D + c) = s.hi;
        *reinterpret_cast<uint4*>(a.qk_lo + (size_t)row * 2 * D + c) = s.lo;
    }
    // V : transpose a [32 rows][D] tile through LDS
    const int ldv = D + 1;
    for (int idx = tid; idx < 32 * (D / 4); idx += 256) {
        const int r = idx / (D / 4), c = (idx - r * (D / 4)) * 4;
        const int row = r0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < a.R) v = *reinterpret_cast<const float4*>(a.qkv + (size_t)row * ld + 2 * D + c);
        float* d = vt_s + r * ldv + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    for (int idx = tid; idx < D * 4; idx += 256) {
        const int n = idx >> 2, j = idx & 3;           // head-dim column n, keys 8j..8j+7 of the tile
        float4 p, q;
        const float* s = vt_s + (8 * j) * ldv + n;
        p.x = s[0]; p.y = s[ldv]; p.z = s[2 * ldv]; p.w = s[3 * ldv];
        q.x = s[4 * ldv]; q.y = s[5 * ldv]; q.z = s[6 * ldv]; q.w = s[7 * ldv];
        const SplitPair sp = split8(p, q);
        const size_t off = (size_t)n * a.Rvt + r0 + 8 * j;
        *reinterpret_cast<uint4*>(a.vt_hi + off) = sp.hi;
        *reinterpret_cast<uint4*>(a.vt_lo + off) = sp.lo;
    }
}

struct AttnB16Args {
    const __bf16 *qk_hi, *qk_lo; int ldqk;     // [R][2D]
    const __bf16 *vt_hi, *vt_lo; int Rvt;      // [D][Rvt]
    float* ctx; int ldc;                       // fp32 context [R][ldc] (or nullptr) ...
    void* ctxp; int ctxp_chunks;               // ... and / or split-bf16 planes, the A operand of the output projection (gemm_planes.h)
    const int* start; const int* len; const int* klen;
    const int2* work;                          // (utterance, 128-query block) items, attn_f32.h: kAttBlk
    const int* nwork;                          // device-driven layout: number of valid work items (the grid is a capacity), or nullptr
    int nitems;                                // host-driven layout: number of work items
    int D; int mask_q;
    unsigned qk_lo_bytes, vt_lo_bytes;         // attn_w32.h: byte distance of the lo planes behind the hi planes (both < 2^31)
    int *slow_count, *slow_count2;             // attn_w32.h: device counters of the waves that left the fast path (the handle's cumulative one, the call's status word), or nullptr
};

// max / sum over the four lanes {l, l^16, l^32, l^48} (the four 8-key groups of one query column) with the gfx950 row swaps:
// v_permlane16_swap / v_permlane32_swap with both operands = x leave (x[own row pair half], x[other half]) in the two results,
// so one VALU op finishes each butterfly step - no ds_bpermute round trip through the LDS crossbar on the softmax chain.
__device__ __forceinline__ float quad_rows_max(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float quad_rows_sum(float v) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

template <int DK>
constexpr size_t attn_b16_lds_bytes() { return (size_t)32 * DK * 4 + (size_t)DK * 128; }

// Phase timing for tools/probes/attn_probe.hip (compiled only with -DFS2_ATT_TIMING): cycles of wave 0 of workgroup (0, 0)
// accumulated per phase of the tile loop.
#ifdef FS2_ATT_TIMING
__device__ long long g_att_phase[8];
#define FS2_T(i) { const long long t_ = __builtin_readcyclecounter(); if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_att_phase[i] += t_ - tprev; tprev = t_; }
#else
#define FS2_T(i)
#endif

template <int DK, int NSPLIT>
__global__ __launch_bounds__(256, 2) void attn_bf16(AttnB16Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem_a[];
    constexpr int KROW = DK * 4;            // bytes per K row: [hi DK bf16 | lo DK bf16]
    constexpr int KSL = DK / 8;             // 16-B slots per plane per row (24 or 16: multiples of 8)
    constexpr int NC = DK / 32;             // k-steps of Q.K^T
    constexpr int NT = DK / 16;             // n-tiles of O
    char* Ks = smem_a;
    char* Vs = smem_a + 32 * KROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, lg = lane >> 4;
    int b, qb;
    if (!att_item64(a.work, a.nwork, a.nitems, b, qb)) return;
    const int h = blockIdx.y;
    const int s0 = __builtin_amdgcn_readfirstlane(a.start[b]), len = __builtin_amdgcn_readfirstlane(a.len[b]), klen = __builtin_amdgcn_readfirstlane(a.klen[b]);
    if (qb >= len) return;                     // second half of an utterance's last 128-query item
    const int q0 = qb + wave * 16;
    const bool wave_live = __builtin_amdgcn_readfirstlane(q0) < len;      // wave-uniform: any of this wave's 16 queries inside the utterance

    // Q fragments (B operand of S^T): row q0 + lr, d = 32c + 8g .. +7
    bf16x8_t qh[NC], ql[NC];
    {
        const int qrow = q0 + lr;
        const bool ok = qrow < len;
        const size_t off = (size_t)(s0 + (ok ? qrow : 0)) * a.ldqk + (size_t)h * DK + lg * 8;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            u32x4 vh = u32x4{0, 0, 0, 0}, vl = vh;
            if (ok) {
                vh = *reinterpret_cast<const u32x4*>(a.qk_hi + off + c * 32);
                vl = *reinterpret_cast<const u32x4*>(a.qk_lo + off + c * 32);
            }
            qh[c] = *reinterpret_cast<bf16x8_t*>(&vh);
            ql[c] = *reinterpret_cast<bf16x8_t*>(&vl);
        }
    }
    f32x4 o[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    const __bf16* kh_base = a.qk_hi + a.D + (size_t)h * DK;
    const __bf16* kl_base = a.qk_lo + a.D + (size_t)h * DK;
    const __bf16* vh_base = a.vt_hi + (size_t)h * DK * a.Rvt;
    const __bf16* vl_base = a.vt_lo + (size_t)h * DK * a.Rvt;

    // Keys beyond klen (last tile of an utterance): their scores are overwritten with -inf whatever K holds, but P = 0 times a
    // non-finite V would still poison P.V, and the rows behind the last utterance are not this call's data: K rows are clamped
    // to the last key; 8-key V^T vectors that start beyond klen are not read at all, and in vectors that straddle klen the keys
    // beyond it are zeroed in registers (they end inside the gap rows / the 8 rows every layout appends after its last utterance,
    // so the read itself stays inside the buffers).
    // Staging: NLD 16-byte loads per thread per operand tile, all issued back-to-back (no branches), held in the
    // same registers for K and V^T in turn.  K(t+1) is fetched while P.V(t) runs, V^T(t) while Q.K^T(t) + softmax run.
    constexpr int NLD = DK / 32;     // 32 * 2*KSL / 256 == DK * 8 / 256
    u32x4 stg[NLD];
    const int ntiles = (klen + 31) / 32;
#define FS2_LOAD_K(key0_)                                                                                   \
    _Pragma("unroll") for (int u = 0; u < NLD; ++u) {                                                       \
        const int idx = tid + u * 256;                                                                      \
        const int key = idx / (2 * KSL), s = idx - key * (2 * KSL);                                         \
        const __bf16* src = ((s >= KSL) ? kl_base - KSL * 8 : kh_base) + (size_t)(s0 + min((key0_) + key, klen - 1)) * a.ldqk + s * 8; \
        stg[u] = *reinterpret_cast<const u32x4*>(src);                                                      \
    }
#define FS2_STORE_K()                                                                                       \
    _Pragma("unroll") for (int u = 0; u < NLD; ++u) {                                                       \
        const int idx = tid + u * 256;                                                                      \
        const int key = idx / (2 * KSL), s = idx - key * (2 * KSL);                                         \
        const int i = ((key >> 1) & 12) | (key & 3);                                                        \
        *reinterpret_cast<u32x4*>(Ks + key * KROW + ((s ^ i) << 4)) = stg[u];                               \
    }
#define FS2_LOAD_V(key0_)                                                                                   \
    _Pragma("unroll") for (int u = 0; u < NLD; ++u) {                                                       \
        const int idx = tid + u * 256;                                                                      \
        const int n = idx >> 3, s = idx & 7;                                                                \
        const __bf16* src = ((s & 4) ? vl_base : vh_base) + (size_t)n * a.Rvt + s0 + (key0_) + (s & 3) * 8; \
        const int nv = klen - ((key0_) + (s & 3) * 8);      /* keys of this 8-key vector that exist */          \
        const void* sp = (nv > 0) ? static_cast<const void*>(src) : static_cast<const void*>(g_zero16);     \
        stg[u] = *reinterpret_cast<const u32x4*>(sp);                                                       \
        if (nv < 8) {               /* last tile only: zero the keys beyond klen whatever the memory holds */ \
            _Pragma("unroll") for (int w = 0; w < 4; ++w)                                                   \
                stg[u][w] &= (nv > 2 * w + 1) ? 0xffffffffu : ((nv > 2 * w) ? 0x0000ffffu : 0u);            \
        }                                                                                                   \
    }
#define FS2_STORE_V()                                                                                       \
    _Pragma("unroll") for (int u = 0; u < NLD; ++u) {                                                       \
        const int idx = tid + u * 256;                                                                      \
        const int d = idx >> 3;     /* head channel d -> LDS row 16 (4 (d >> 6) + (d & 3)) + ((d & 63) >> 2) */ \
        *reinterpret_cast<u32x4*>(Vs + swz((((d >> 6) << 2) | (d & 3)) * 16 + ((d & 63) >> 2), idx & 7)) = stg[u]; \
    }
    // Full tiles (every tile but possibly the last of an utterance) take the FAST staging path: each 16-byte piece a lane moves is described by
    // loop-invariant 32-bit offsets against a wave-uniform base that advances by a scalar add per tile -- a hi-plane and a lo-plane
    // instruction share them -- so no per-tile address arithmetic runs on the VALU (PMC, round 2: 3.5 VALU instructions per MFMA, most of
    // them the 64-bit address chains, the clamps and the masks of the macros above, which now serve the partial tile only).
    constexpr int NH = NLD / 2;               // instructions per plane and operand tile
    unsigned kgo[NH], vgo[NH];                // global element offsets inside a K tile / a V^T tile
#pragma unroll
    for (int u = 0; u < NH; ++u) {
        const int idx = tid + u * 256;
        const int key = idx / KSL, sl = idx - key * KSL;
        kgo[u] = (unsigned)(key * a.ldqk + sl * 8);
        vgo[u] = (unsigned)((idx >> 2) * a.Rvt + (idx & 3) * 8);
    }
#define FS2_LOAD_K_FAST(key0_)                                                                              \
    {                                                                                                       \
        const __bf16* bh_ = kh_base + (size_t)(s0 + (key0_)) * a.ldqk;                                      \
        const __bf16* bl_ = kl_base + (size_t)(s0 + (key0_)) * a.ldqk;                                      \
        _Pragma("unroll") for (int u = 0; u < NH; ++u) {                                                    \
            stg[u] = *reinterpret_cast<const u32x4*>(bh_ + kgo[u]);                                         \
            stg[NH + u] = *reinterpret_cast<const u32x4*>(bl_ + kgo[u]);                                    \
        }                                                                                                   \
    }
#define FS2_STORE_K_FAST()      /* (the 32-bit LDS addresses are recomputed: keeping them would cost 9 more registers at d_k = 192) */ \
    _Pragma("unroll") for (int u = 0; u < NH; ++u) {                                                        \
        const int idx = tid + u * 256;                                                                      \
        const int key = idx / KSL, sl = idx - key * KSL;                                                    \
        const int i = ((key >> 1) & 12) | (key & 3);                                                        \
        *reinterpret_cast<u32x4*>(Ks + key * KROW + ((sl ^ i) << 4)) = stg[u];                              \
        *reinterpret_cast<u32x4*>(Ks + key * KROW + (((KSL + sl) ^ i) << 4)) = stg[NH + u];                 \
    }
#define FS2_LOAD_V_FAST(key0_)                                                                              \
    {                                                                                                       \
        const __bf16* bh_ = vh_base + s0 + (key0_);                                                         \
        const __bf16* bl_ = vl_base + s0 + (key0_);                                                         \
        _Pragma("unroll") for (int u = 0; u < NH; ++u) {                                                    \
            stg[u] = *reinterpret_cast<const u32x4*>(bh_ + vgo[u]);                                         \
            stg[NH + u] = *reinterpret_cast<const u32x4*>(bl_ + vgo[u]);                                    \
        }                                                                                                   \
    }
#define FS2_STORE_V_FAST()                                                                                  \
    _Pragma("unroll") for (int u = 0; u < NH; ++u) {                                                        \
        const int idx = tid + u * 256;                                                                      \
        const int n = idx >> 2;                                                                             \
        const int o_ = swz((((n >> 6) << 2) | (n & 3)) * 16 + ((n & 63) >> 2), idx & 3);                    \
        *reinterpret_cast<u32x4*>(Vs + o_) = stg[u];                                                        \
        *reinterpret_cast<u32x4*>(Vs + (o_ ^ 64)) = stg[NH + u];                                            \
    }
    // (a tile is "full" when all its 32 keys exist; the staging of tile t and of tile t + 1 may take different paths)
    bool k_fast = false, v_fast = false;
    if (ntiles > 0) {
        k_fast = 32 <= klen;
        if (k_fast) { FS2_LOAD_K_FAST(0) FS2_STORE_K_FAST() } else { FS2_LOAD_K(0) FS2_STORE_K() }
        v_fast = k_fast;
        if (v_fast) { FS2_LOAD_V_FAST(0) } else { FS2_LOAD_V(0) }
    }
#ifdef FS2_ATT_TIMING
    long long tprev = __builtin_readcyclecounter();
#endif
    for (int kt = 0; kt < ntiles; ++kt) {
        const int key0 = kt * 32;
        FS2_T(5)
        __syncthreads();          // (A) K(kt) visible; every wave is done with P.V(kt-1), so the V^T buffer is free
        FS2_T(0)
        // (V^T(kt) was requested before this barrier -- at the end of the previous tile, or ahead of the loop -- and stays in flight during
        //  Q.K^T and the softmax)
        bf16x8_t ph, pl;
        if (wave_live) {          // (a wave whose 16 queries all lie beyond the utterance only helps with the staging)
        f32x4 st[2];
        st[0] = f32x4{0.f, 0.f, 0.f, 0.f};
        st[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const char* krow0 = Ks + (8 * (lr >> 2) + (lr & 3)) * KROW;     // sub-tile 0 row; sub-tile 1 is 4 keys further
            const char* krow1 = krow0 + 4 * KROW;
            // One k-step (32 channels) = 4 ds_read_b128 (hi / lo piece of the two sub-tiles' rows) + 6 MFMAs.  The fragments of step c + 1 are
            // requested BEFORE the MFMAs of step c (register double buffer, order pinned with sched_group_barrier): the ISA of the previous form
            // (8 reads -> s_waitcnt -> 12 MFMAs, three times) paid the LDS round trip three times per tile with nothing to cover it.  The cross
            // terms (lo.hi + hi.lo) and the main term (hi.hi) of a sub-tile go to separate accumulators, so no MFMA depends on either of the
            // two issued just before it.
            f32x4 sc[2], sm[2];
            sc[0] = f32x4{0.f, 0.f, 0.f, 0.f}; sc[1] = sc[0]; sm[0] = sc[0]; sm[1] = sc[0];
            bf16x8_t kh0[2], kh1[2], kl0[2], kl1[2];
            auto kfetch = [&](int c, int b) {
                const int sh = ((c * 4 + lg) ^ lr) << 4, sl = ((KSL + c * 4 + lg) ^ lr) << 4;
                kh0[b] = *reinterpret_cast<const bf16x8_t*>(krow0 + sh);
                kh1[b] = *reinterpret_cast<const bf16x8_t*>(krow1 + sh);
                if (NSPLIT == 3) {
                    kl0[b] = *reinterpret_cast<const bf16x8_t*>(krow0 + sl);
                    kl1[b] = *reinterpret_cast<const bf16x8_t*>(krow1 + sl);
                }
            };
            constexpr int kKR = NSPLIT == 3 ? 4 : 2, kKM = NSPLIT == 3 ? 6 : 2;
            kfetch(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, kKR, 0);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int b = c & 1;
                if (c + 1 < NC) {
                    kfetch(c + 1, b ^ 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, kKR, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, kKM, 0);
                if (NSPLIT == 3) {
                    sc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl0[b], qh[c], sc[0], 0, 0, 0);
                    sc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl1[b], qh[c], sc[1], 0, 0, 0);
                }
                sm[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh0[b], qh[c], sm[0], 0, 0, 0);
                sm[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh1[b], qh[c], sm[1], 0, 0, 0);
                if (NSPLIT == 3) {
                    sc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh0[b], ql[c], sc[0], 0, 0, 0);
                    sc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh1[b], ql[c], sc[1], 0, 0, 0);
                }
            }
            st[0] = sm[0] + sc[0];
            st[1] = sm[1] + sc[1];
        }
        // st[t][r] = log2(e) * score of key key0 + 8g + 4t + r for query lr (Q was pre-scaled by log2(e)/sqrt(d_k)),
        // so the softmax runs on v_exp_f32 (2^x) directly.
        FS2_T(1)
        float p[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[t * 4 + r] = st[t][r];
        if (key0 + 32 > klen) {            // only the last tile of an utterance has masked keys (wave-uniform branch)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = key0 + 8 * lg + 4 * (j >> 2) + (j & 3);
                if (key >= klen) p[j] = -INFINITY;
            }
        }
        float tmax = fmaxf(fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3])), fmaxf(fmaxf(p[4], p[5]), fmaxf(p[6], p[7])));
        tmax = quad_rows_max(tmax);
        const float m_new = fmaxf(m_run, tmax);
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            p[j] = __builtin_amdgcn_exp2f(p[j] - m_new);
            psum += p[j];
        }
        psum = quad_rows_sum(psum);
        if (__any(m_new != m_run)) {       // some row's running max moved: rescale the accumulators (rare after the first tiles)
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);      // 2^(-inf) = 0 on the first tile
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ar = __shfl(alpha, lg * 4 + r);
#pragma unroll
                for (int n = 0; n < NT; ++n) o[n][r] *= ar;
            }
            m_run = m_new;
        }
        l_run += psum;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const __bf16 hb = (__bf16)p[j];
            ph[j] = hb;
            pl[j] = (__bf16)(p[j] - (float)hb);
        }
        }
        FS2_T(2)
        if (v_fast) { FS2_STORE_V_FAST() } else { FS2_STORE_V() }
        // K(kt + 1) is requested as soon as the staging registers are free, i.e. BEFORE the barrier (the loads touch no LDS): it has the barrier
        // wait and P.V(kt) to arrive
        k_fast = key0 + 64 <= klen;
        if (kt + 1 < ntiles) { if (k_fast) { FS2_LOAD_K_FAST(key0 + 32) } else { FS2_LOAD_K(key0 + 32) } }
        __syncthreads();          // (B) V^T(kt) visible; every wave is done with Q.K^T(kt), so the K buffer is free
        FS2_T(3)
        constexpr int PG = (DK > 128) ? 2 : 4;     // n-tiles per group: independent accumulators between dependent MFMAs
        if (wave_live) {
            // V^T fragments of group i + 1 are requested before the MFMAs of group i (register double buffer; the previous form -- 2 PG reads ->
            // s_waitcnt -> 3 PG MFMAs, NT / PG times -- exposed the LDS round trip six times per tile at d_k = 192)
            bf16x8_t vh[2][PG], vl[2][PG];
            auto vfetch = [&](int n4, int b) {
#pragma unroll
                for (int u = 0; u < PG; ++u) {
                    const int row = (n4 + u) * 16 + lr;
                    vh[b][u] = *reinterpret_cast<const bf16x8_t*>(Vs + swz(row, lg));
                    if (NSPLIT == 3) vl[b][u] = *reinterpret_cast<const bf16x8_t*>(Vs + swz(row, 4 + lg));
                }
            };
            constexpr int kVR = (NSPLIT == 3 ? 2 : 1) * PG, kVM = (NSPLIT == 3 ? 3 : 1) * PG;
            vfetch(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, kVR, 0);
#pragma unroll
            for (int n4 = 0; n4 < NT; n4 += PG) {
                const int b = (n4 / PG) & 1;
                if (n4 + PG < NT) {
                    vfetch(n4 + PG, b ^ 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, kVR, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, kVM, 0);
                if (NSPLIT == 3) {
#pragma unroll
                    for (int u = 0; u < PG; ++u) o[n4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vh[b][u], o[n4 + u], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < PG; ++u) o[n4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vl[b][u], o[n4 + u], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < PG; ++u) o[n4 + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vh[b][u], o[n4 + u], 0, 0, 0);
            }
        }
        FS2_T(4)
        if (kt + 1 < ntiles) {
            if (k_fast) { FS2_STORE_K_FAST() } else { FS2_STORE_K() }
            v_fast = key0 + 64 <= klen;          // V^T(kt + 1): likewise ahead of barrier (A)
            if (v_fast) { FS2_LOAD_V_FAST(key0 + 32) } else { FS2_LOAD_V(key0 + 32) }
        }
    }
#undef FS2_LOAD_K_FAST
#undef FS2_STORE_K_FAST
#undef FS2_LOAD_V_FAST
#undef FS2_STORE_V_FAST
#undef FS2_LOAD_K
#undef FS2_STORE_K
#undef FS2_LOAD_V
#undef FS2_STORE_V
    // V^T rows sit in LDS so that n-tile n of lane lr is head channel 64 (n >> 2) + 4 lr + (n & 3): a lane's accumulators
    // o[4g .. 4g+3][r] are four consecutive channels of query row 4 lg + r -> 16-byte stores (or 8 + 8 bytes of planes).
    const float linv = (l_run > 0.f) ? 1.f / l_run : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float li = __shfl(linv, lg * 4 + r);
        const int qrow = q0 + lg * 4 + r;
        if (qrow >= len) continue;
        const bool dead = a.mask_q && qrow >= klen;
        const size_t row = (size_t)(s0 + qrow);
#pragma unroll
        for (int g = 0; g < NT / 4; ++g) {
            const int col = h * DK + 64 * g + 4 * lr;
            f32x4 v = f32x4{o[4 * g][r], o[4 * g + 1][r], o[4 * g + 2][r], o[4 * g + 3][r]} * li;
            if (dead) v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.ctx) *reinterpret_cast<f32x4*>(a.ctx + row * a.ldc + col) = v;
            if (a.ctxp) store_planes4(a.ctxp, row, a.ctxp_chunks, col, v);
        }
    }
}

}  // namespace fs2
