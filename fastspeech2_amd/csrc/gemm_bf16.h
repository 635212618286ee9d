// Shared pieces of the split-bf16 ("bf16x3") GEMM kernels (gemm_planes.h) on v_mfma_f32_16x16x32_bf16 (gfx950, 2.5 PFLOP/s
// dense peak): the operand split, the LDS image, the weight repack, the V^T part of the fused QKV epilogue and the row kernel.
//
// Plain bf16 misses the 1e-3 mel tolerance by 10x, so the parity mode splits every operand
// x = hi + lo (hi = bf16(x), lo = bf16(x - hi), ~16 mantissa bits) and issues three MFMAs per fragment pair,
// lo*hi + hi*lo + hi*hi, accumulating in fp32 (NSPLIT = 3).  NSPLIT = 1 is plain bf16.
//   * weights are split once at load time into the exact LDS image: [Npad][chunk][tap][hi 32 | lo 32] (k-step order, 128 B
//     per (n, tap, chunk)); activations travel as "planes" in the same image (common.h), written by their producer.
//   * LDS image (both operands): row r = 128 B = 8 slots of 16 B (slots 0-3: hi k 0-7 .. 24-31, slots 4-7: lo), physical
//     slot = slot ^ ((r >> 1) & 7); with the lane -> row permutation rperm (common.h) every ds_read_b128 lane group lands on
//     16 distinct 16-B bank slots for every row offset (conv tap).
//   * MFMA operands: lane l supplies A[i = l&15][k = 8*(l>>4) .. +7] and B[k = 8*(l>>4) .. +7][j = l&15];
//     C/D: col = l&15, row = 4*(l>>4) + reg.
#pragma once
#include "common.h"

namespace fs2 {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

// one 16x16x32 MFMA on fragments held as raw 16-byte vectors: bf16 or (F16) fp16 operands, fp32 accumulate, same rate
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const bf16x8_t a, const bf16x8_t b, const f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

// Barrier that also covers this wave's outstanding LDS-DMA (global_load_lds counts on vmcnt).  hipcc normally emits the
// vmcnt(0) itself in front of __syncthreads(), but in one kernel of this project (an LDS-DMA variant of the attention
// loop) it hoisted that wait out of the loop and the loop-top barrier raced with the DMA issued in the previous
// iteration; the explicit wait makes the ordering independent of that heuristic (a duplicate s_waitcnt is free).
__device__ __forceinline__ void dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// Counted form for ring-buffered loops (LDS-DMA completes in issue order): this wave's LDS-DMA except its newest N instructions has landed (and its LDS
// reads returned), then the workgroup meets.
template <int N> __device__ __forceinline__ void dma_wait_barrier() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One LDS-DMA instruction (64 lanes x 16 B -> 1 KB of LDS at byte offset lds_off, which must be wave-uniform).  Written as
// inline asm so that M0 comes from an SGPR operand: with __builtin_amdgcn_global_load_lds hipcc keeps the LDS pointer in a
// VGPR and emits v_readfirstlane + s_mov m0 per instruction, and the phase probe (tools/probes/gemm_probe.hip) showed the 4-6
// DMA instructions of a k-step costing 13-30 % of the step.  The compiler does not see this VMEM operation: completion is
// always awaited explicitly (dma_barrier), and its own vmcnt bookkeeping for other loads only becomes more conservative.
__device__ __forceinline__ void dma16(const void* src, unsigned lds_off) {
    // M0 is not on the clobber list on purpose: for the AMDGPU backend M0 is a RESERVED register -- never allocated to a value; every compiler use
    // (LDS-DMA builtins, movrel, sendmsg, GWS) writes it immediately in front of the instruction that reads it -- and hipcc says so itself when
    // "m0" is listed: "inline asm clobber list contains reserved registers: m0 ... clobbering them may lead to undefined behaviour" (-Winline-asm).
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_off) : "memory");
}

// Global-memory bytes (attn_w32.h, gemm_row4.h): the tile sources keep their address space through the pointer arithmetic below (rebuilt from integers as
// generic pointers they turn the register-staged loads into FLAT loads, which count on lgkmcnt as well and take a 64-bit address each).
typedef const __attribute__((address_space(1))) char gchar_t;
// a wave-uniform pointer as the compiler can see it (an "s" asm operand must be provably uniform)
__device__ __forceinline__ gchar_t* uniform_ptr(const void* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<gchar_t*>(((unsigned long long)hi << 32) | lo);
}

// One LDS-DMA instruction: wave-uniform 64-bit base (SGPR pair) + per-lane unsigned 32-bit byte offset -> 1 KB of LDS at lds_off.
__device__ __forceinline__ void dma16_so(gchar_t* base, unsigned off, unsigned lds_off) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(off), "s"(base), "s"(lds_off) : "memory");
}

#ifndef FS2_SETPRIO
#define FS2_SETPRIO 1
#endif
constexpr int kB16BN = 128;     // output columns per workgroup tile

struct SplitPair { uint4 hi, lo; };

// 8 consecutive fp32 -> (hi bf16 x8, lo bf16 x8)
__device__ __forceinline__ SplitPair split8(const float4& p, const float4& q) {
    const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
    bf16x8_t h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 hb = (__bf16)v[i];
        h[i] = hb;
        l[i] = (__bf16)(v[i] - (float)hb);
    }
    SplitPair s;
    s.hi = *reinterpret_cast<uint4*>(&h);
    s.lo = *reinterpret_cast<uint4*>(&l);
    return s;
}

__device__ __forceinline__ int swz(int row, int slot) { return (row << 7) + ((slot ^ ((row >> 1) & 7)) << 4); }

// the operand of one block-scaled 16x16x128 MFMA on e4m3 data: the 16-byte slots g and 4 + g of LDS row `row` as registers 0-3 | 4-7
__device__ __forceinline__ v8i_t lds_frag8(const char* base, int row, int g) {
    const v4i_t x0 = *reinterpret_cast<const v4i_t*>(base + swz(row, g));
    const v4i_t x1 = *reinterpret_cast<const v4i_t*>(base + swz(row, 4 + g));
    return __builtin_shufflevector(x0, x1, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Channel order inside a 32-channel chunk of the bf16 images: the 16-byte slot g (k-group of lane-group g) holds channels
// {4g..4g+3} and {16+4g..16+4g+3}.  Any permutation is legal as long as A and B agree (the MFMA sums over k); with this one a
// producer that holds 4 consecutive channels writes 8 contiguous bytes of the hi half and 8 of the lo half (store_planes4).
__device__ __host__ __forceinline__ int kperm(int p) { const int slot = p >> 3, j = p & 7; return (j < 4) ? 4 * slot + j : 16 + 4 * slot + (j - 4); }

// V part of the fused QKV epilogue: the BM x 128 fp32 tile (+bias, staged in LDS by the GEMM) leaves as V^T hi/lo planes
// [D][Rvt], 8 consecutive keys of one channel per 16-byte store (key index contiguous: what the P.V MFMA wants as B operand).
// Rows that are gaps or beyond R are written as zeros (P = 0 times a non-finite V would poison the P.V sum).
constexpr int kQkvLd = kB16BN + 4;      // fp32 tile row stride in LDS (floats)
// Column swizzle of the fp32 tile the V^T planes are transposed through: the transposing read has the four lanes of a column
// 8 rows apart, and 8 rows of 132 floats are a multiple of 32 banks -- without the term those four lanes meet in one bank (15 %
// of gemm_qkv8_bf16's LDS cycles were conflicts).  XOR-ing the column with 8 ((row >> 3) & 3) sends the four 8-row groups to
// four different 8-column bank groups; a multiple of 8, so the 16-byte writes stay aligned and whole.
__device__ __forceinline__ int qkv_tile_swz(int row) { return ((row >> 3) & 3) << 3; }

template <int BM>
__device__ __forceinline__ void vt_tile_store(const GemmArgs& a, const float* tile, int m0, int n0, int tid) {
    const int* __restrict__ rpos = a.row_pos;
    const int D = a.att_D;
    __bf16* vth = reinterpret_cast<__bf16*>(a.vt_hi);
    __bf16* vtl = reinterpret_cast<__bf16*>(a.vt_lo);
#pragma unroll
    for (int u = 0; u < BM / 16; ++u) {
        const int idx = tid + u * 256;
        // column c, rows 8j .. 8j+7 of the tile; 4 consecutive lanes share a column (64-byte V^T segments),
        // consecutive lane quads take consecutive columns (different LDS banks)
        const int c = (idx >> 2) & 127, j = ((idx >> 9) << 2) | (idx & 3);
        const int row = m0 + 8 * j;
        if (row >= a.Rvt) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int rr = row + e;
            const bool ok = rr < a.R && (rpos == nullptr || rpos[rr] >= 0);
            v[e] = ok ? tile[(8 * j + e) * kQkvLd + (c ^ qkv_tile_swz(8 * j))] : 0.f;
        }
        const SplitPair sp = split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
        const size_t off = (size_t)(n0 - 2 * D + c) * a.Rvt + row;
        *reinterpret_cast<uint4*>(vth + off) = sp.hi;
        *reinterpret_cast<uint4*>(vtl + off) = sp.lo;
    }
}

__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};   // LDS-DMA source for out-of-range pieces (a masked lane would leave stale LDS bytes)

// Per-lane optional loads without a branch: a lane that must not load reads the zero vector instead (address select).  hipcc
// compiles `if (ok) v += *p` inside an unrolled tile prologue into a branch with its own s_waitcnt vmcnt(0) per load: the 24-32
// residual loads of a tile became as many serialized memory round trips (~11 us per LayerNorm-fused GEMM launch at c3).
__device__ __forceinline__ f32x4 load4_or_zero(const float* p, bool ok) {
    const float* q = ok ? p : g_zero16;
    return *reinterpret_cast<const f32x4*>(q);
}
__device__ __forceinline__ uint2 load8_or_zero(const char* p, bool ok) {
    const char* q = ok ? p : reinterpret_cast<const char*>(g_zero16);
    return *reinterpret_cast<const uint2*>(q);
}
__device__ __forceinline__ int loadi_or_zero(const int* p, bool ok) {
    const int* q = ok ? p : reinterpret_cast<const int*>(g_zero16);
    return *q;
}

// Row epilogue as its own HBM-bound kernel (used after gemm_pl_bf16 when the op ends in a LayerNorm, a
// positional-encoding add or the scalar head): in place on Y [R, N], one wavefront per row, N <= 1024.
//   v = LN(y) (if ln_g) -> act_post -> v*x_scale + alpha*pe[pos] -> store; dot_out[row] = v . dot_w + dot_b
// With Yp the result is also written as split-bf16 planes, the A operand of the next GEMM (gemm_planes.h).
__global__ __launch_bounds__(256) void ln_rows(GemmArgs a) {
    // ln_groups G > 1: every row holds G independent column groups (stacked layers over one input): one wavefront per (row, group)
    const int G = a.ln_groups > 1 ? a.ln_groups : 1;
    const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int row = G > 1 ? unit / G : unit, grp = G > 1 ? unit - row * G : 0;
    const int lane = threadIdx.x & 63;
    if (row >= a.R) return;
    const int N = a.N / G, co = grp * N;             // this unit's columns [co, co + N)
    const int pos = a.row_pos ? a.row_pos[row] : 0;
    float* y = a.Y + (size_t)row * a.ldy + co;
    const float* ysrc = a.Ysrc ? a.Ysrc + (size_t)row * a.ldsrc + co : y;      // out-of-place form (pre-LN blocks keep the un-normalised stream)
    const int pc = a.Yp ? (G > 1 ? N : a.yp_chunks * 32) : 0;      // channels of the output planes (>= N, zero padded)
    const float* ln_g = a.ln_g ? a.ln_g + co : nullptr;
    const float* ln_b = a.ln_b ? a.ln_b + co : nullptr;
    const float* dot_w = a.dot_w ? a.dot_w + co : nullptr;
    float* dot_out = a.dot_out ? a.dot_out + (size_t)grp * a.dot_gstride : nullptr;
    if (pos < 0) {
        if (a.Y)
            for (int c = lane * 4; c < N; c += 256) *reinterpret_cast<float4*>(y + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = lane * 4; c < pc; c += 256) store_planes4m(a.Yp, row, a.yp_chunks, co + c, f32x4{0.f, 0.f, 0.f, 0.f}, a.yp_f16, a.yp_scale);
        if (dot_w && lane == 0) dot_out[row] = 0.f;
        return;
    }
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + j * 256;
        v[j] = (c < N) ? *reinterpret_cast<const float4*>(ysrc + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < N)
            for (int z = 1; z < a.ksplit; ++z) {      // split-K partials of the GEMM, added in a fixed order
                const float4 q = *reinterpret_cast<const float4*>(a.kpart + (size_t)(z - 1) * a.kpart_stride + (size_t)row * a.ldy + co + c);
                v[j].x += q.x; v[j].y += q.y; v[j].z += q.z; v[j].w += q.w;
            }
        if (a.relu_pre) {      // ReLU in front of the LayerNorm (predictors); idempotent when the GEMM epilogue already applied it
            v[j].x = fmaxf(v[j].x, 0.f); v[j].y = fmaxf(v[j].y, 0.f); v[j].z = fmaxf(v[j].z, 0.f); v[j].w = fmaxf(v[j].w, 0.f);
        }
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    if (ln_g) {
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)N;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane * 4 + j * 256;
            if (c < N) {
                const float dx = v[j].x - mean, dy = v[j].y - mean, dz = v[j].z - mean, dw = v[j].w - mean;
                q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.f / sqrtf(q / (float)N + a.ln_eps);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = lane * 4 + j * 256;
            if (c < N) {
                const float4 g = *reinterpret_cast<const float4*>(ln_g + c);
                const float4 b = *reinterpret_cast<const float4*>(ln_b + c);
                v[j].x = (v[j].x - mean) * rstd * g.x + b.x;
                v[j].y = (v[j].y - mean) * rstd * g.y + b.y;
                v[j].z = (v[j].z - mean) * rstd * g.z + b.z;
                v[j].w = (v[j].w - mean) * rstd * g.w + b.w;
            }
        }
    }
    const float alpha = (a.pe && a.pe_alpha) ? a.pe_alpha[0] : 1.f;
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = lane * 4 + j * 256;
        if (c < N) {
            float4 t = v[j];
            t.x = apply_act(t.x, a.act_post); t.y = apply_act(t.y, a.act_post);
            t.z = apply_act(t.z, a.act_post); t.w = apply_act(t.w, a.act_post);
            if (a.pe) {
                const float4 p = *reinterpret_cast<const float4*>(a.pe + (size_t)pos * a.pe_ld + co + c);
                t.x = t.x * a.x_scale + alpha * p.x; t.y = t.y * a.x_scale + alpha * p.y;
                t.z = t.z * a.x_scale + alpha * p.z; t.w = t.w * a.x_scale + alpha * p.w;
            }
            if (dot_w) {
                const float4 w = *reinterpret_cast<const float4*>(dot_w + c);
                d += (t.x * w.x + t.y * w.y) + (t.z * w.z + t.w * w.w);
            }
            if (a.Y) *reinterpret_cast<float4*>(y + c) = t;
            if (a.Yp) store_planes4m(a.Yp, row, a.yp_chunks, co + c, f32x4{t.x, t.y, t.z, t.w}, a.yp_f16, a.yp_scale);
        } else if (c < pc) {
            store_planes4m(a.Yp, row, a.yp_chunks, co + c, f32x4{0.f, 0.f, 0.f, 0.f}, a.yp_f16, a.yp_scale);
        }
    }
    if (dot_w) {
        for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
        if (lane == 0) dot_out[row] = d + (a.dot_b ? a.dot_b[grp] : 0.f);
    }
}

// weights [N][C][k] fp32 -> split bf16 LDS image [Npad][nchunks][k][hi 32 | lo 32]; optional BatchNorm fold.
// f16 != 0: the same image in _Float16 (fp16 hi + fp16 lo), the operand of the two- / one-term FFN arithmetic.
__global__ void repack_weight_bf16(const float* w, int N, int C, int k, int Npad, int nchunks, const float* bn_g,
                                   const float* bn_v, float bn_eps, __bf16* out, int f16 = 0, int ldw = 0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)Npad * k * nchunks * 32;
    if (i >= total) return;
    const int kk = (int)(i & 31);
    const int chunk = (int)((i >> 5) % nchunks);
    const int tap = (int)((i / ((int64_t)32 * nchunks)) % k);
    const int n = (int)(i / ((int64_t)32 * nchunks * k));
    const int c = chunk * 32 + kperm(kk);
    float v = 0.f;
    if (n < N && c < C) {
        v = w[((size_t)n * (ldw ? ldw : C) + c) * k + tap];
        if (bn_g) v *= bn_g[n] / sqrtf(bn_v[n] + bn_eps);
    }
    const size_t base = (((size_t)n * nchunks + chunk) * k + tap) * 64;     // k-step order: it = chunk * k + tap
    if (f16) {
        _Float16* o16 = reinterpret_cast<_Float16*>(out);
        const _Float16 hi = (_Float16)v;
        o16[base + kk] = hi;
        o16[base + 32 + kk] = (_Float16)(v - (float)hi);
        return;
    }
    const __bf16 hi = (__bf16)v;
    const __bf16 lo = (__bf16)(v - (float)hi);
    out[base + kk] = hi;
    out[base + 32 + kk] = lo;
}

}  // namespace fs2
