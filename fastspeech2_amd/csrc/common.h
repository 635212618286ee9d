// Shared definitions for the gfx950 kernels of libfs2_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fs2 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// NOTE: register-resident staging arrays must use these native vector types.  Arrays of HIP's float4 / uint4
// (struct wrappers) larger than 64 bytes are not scalarised by hipcc and end up in scratch memory.
typedef short bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kGap = 8;        // upper bound of the zero rows between packed sequences (capacity formulas); a model uses
constexpr int kTailRows = 8;   // zero rows every layout appends behind its last utterance
constexpr int kMinGap = 4;     // max(kMinGap, its largest conv halo (k - 1) / 2): 4 for the default 9-tap FFN (fs2_handle::gap)
constexpr int kMaxHalo = 16;   // LDS rows reserved for conv halos (ktaps <= 17)
constexpr int kBK = 32;        // K-chunk (channels per LDS stage) of the fp32 GEMMs
constexpr int kLd = kBK + 4;   // LDS row stride in floats (16-B aligned, breaks the 128-B bank period)

constexpr int kAttBQ = 64;   // queries per workgroup (4 waves x 16) of attn_f32 / attn_bf16
constexpr int kAttBlk = 128; // queries per work-list item: one workgroup of attn_w32 (4 waves x 32), two of the 64-query kernels

// The 64-query kernels (attn_f32, attn_bf16) over the 128-query work list: grid.x = 2 x (items rounded up to a multiple of 8);
// workgroup x serves half (x >> 3) & 1 of item 8 (x >> 4) + (x & 7), so x % 8 -- the XCD the workgroup runs on -- is the item's
// position in the eight interleaved queues (fs2_runtime.hip: build_work_list) and the halves of an item are dispatched 8 apart.
// Returns false when there is nothing to do; q0 = first query of the workgroup.
inline unsigned att_grid64(int nitems) { return 2u * (unsigned)((nitems + 7) & ~7); }
__device__ __forceinline__ bool att_item64(const int2* work, const int* nwork, int nitems, int& b, int& q0) {
    const int x = (int)blockIdx.x;
    const int item = ((x >> 4) << 3) | (x & 7);
    if (item >= (nwork != nullptr ? *nwork : nitems)) return false;
    const int2 wk = work[item];
    // (explicit scalars: the values are wave-uniform, but loaded through the vector cache they would occupy vector registers -- and
    //  attn_bf16<192,3> has none to spare)
    b = __builtin_amdgcn_readfirstlane(wk.x);
    q0 = __builtin_amdgcn_readfirstlane(wk.y) * kAttBlk + ((x >> 3) & 1) * kAttBQ;
    return b >= 0;                             // (-1: padding entry of the XCD-interleaved work list)
}

// Activation layout ("gapped packed rows"): utterance b owns rows [start[b], start[b]+len[b]) of every
// [R, width] activation buffer; at least kGap zero rows separate utterances and precede the first one,
// so a k-tap convolution along the row axis needs no boundary logic: it simply reads the zero rows.
// Every kernel that produces an activation buffer writes zeros into the gap rows (row_pos < 0).
struct SeqMeta {
    const int* start;     // [B] first row
    const int* len;       // [B] rows stored / seen by convolutions
    const int* klen;      // [B] attention keys (<= len)
    const int* vlen;      // [B] true (unpadded) length
    const int* row_pos;   // [Rpad] position inside the utterance, -1 for gap rows
    const int* row_seq;   // [Rpad] utterance index, -1 for gap rows
    int B, R;
};

// Arguments of the conv-as-GEMM kernels (gemm_f32.h).  Y = epilogue(sum_taps X[row+tap-P] . W[tap]).
struct GemmArgs {
    const float* X; int ldx; int C;        // input [R, ldx], C channels contracted per tap (C % 4 == 0)
    const float* W; int Cpad; int ktaps;   // repacked weights [Npad][ktaps][Cpad], zero padded
    int N; int R;
    const int* row_pos;                    // [>= R] or nullptr (all rows valid)
    const float* bias;                     // [N] or nullptr
    const float* resid; int ldr;           // [R, ldr] or nullptr
    int relu_pre;                          // ReLU before the LayerNorm
    const float* ln_g; const float* ln_b; float ln_eps;   // LayerNorm over the N outputs if ln_g
    int act_post;                          // 0 none, 1 relu, 2 tanh
    const float* pe; int pe_ld; const float* pe_alpha; float x_scale;  // v = v*x_scale + alpha*pe[pos] if pe
    const float* dot_w; const float* dot_b; float* dot_out;            // dot_out[row] = v . dot_w + dot_b
    float* Y; int ldy;                     // output [R, ldy] or nullptr
    const float* Ysrc; int ldsrc;          // ln_rows only: read the rows from here instead of Y (out-of-place LayerNorm)
    const void* Wb;                        // split-bf16 weight image (gemm_bf16.h) or nullptr
    // fused QKV epilogue (bf16 attention operands, attn_bf16.h): when qk_hi != nullptr the tile is not written to Y
    // but split into bf16 hi/lo planes: columns [0,2D) -> qk_hi/lo [Rvt][2D] (Q scaled by q_scale), [2D,3D) -> V^T [D][Rvt]
    void *qk_hi, *qk_lo, *vt_hi, *vt_lo; int att_D; int Rvt; float q_scale;
    float* scratch;                        // [R, N] scratch for two-pass epilogues when Y == nullptr
    // split-bf16 activation planes (gemm_planes.h): [rows][Cpad/32][hi 32 | lo 32] bf16, 128 B per (row, chunk), the
    // exact LDS row image of the MFMA kernels.  Xp: input planes (same rows as X); xp_scratch: where launch_gemm may
    // build them from X when the producer did not; Yp: output planes (yp_chunks 32-channel chunks per row), written
    // by the epilogue next to / instead of Y.
    const void* Xp; void* xp_scratch; void* Yp; int yp_chunks;
    // fp16 variant of the planes / weight image (same layout, [hi 32 | lo 32] _Float16): the two- and one-term arithmetic of the FFN
    // convolution.  yp_f16: write Yp as fp16 planes; f16_terms: 0 = bf16 arithmetic, else Xp / W are fp16 images and
    // the kernel issues f16_terms MFMAs per fragment pair (3: lo*hi + hi*lo + hi*hi, 2: lo*hi + hi*hi = weights rounded once, 1: hi*hi)
    int yp_f16; int f16_terms;
    // "mx" arithmetic of the FFN convolution (gemm_mx.h): yp_f16 == 2 writes Yp as mx planes with the static scale yp_scale = 2^ka;
    // mx != 0: Xp are mx planes, W the mx weight image (same unit order), mx_scale / mx_scale_b the E8M0 bytes (x 0x01010101) of the
    // A / B side of the scaled MFMA (127 - ka - 11 and 127 - kw)
    float yp_scale; int mx; int mx_scale, mx_scale_b;
    // deterministic split-K (small grids with a long K: the loop is a serial chain of k-steps): workgroup z of grid.z accumulates the
    // 32-channel chunks [z, z+1) * Cpad/32/ksplit (all taps of them); split 0 (which also adds bias + residual) writes Y, split z > 0
    // writes kpart + (z-1) * kpart_stride; the row kernel that follows (ln_rows) adds the partials in a fixed order and applies the
    // whole epilogue (ReLU, LayerNorm, activation, planes)
    int ksplit; float* kpart; size_t kpart_stride;
    size_t kpart_cap;                      // floats available at kpart (launch_gemm picks a split that fits)
    int regime_rows;                       // row count the kernel-variant choice is based on (0: R).  Frame-level launches pass an estimate derived from
                                           // the PHONEME count, which the host knows in both layout modes, so that the host- and the device-driven
                                           // layout of one batch always pick the same variants (-> bit-identical results); see fs2_decode
    // grouped operands (the pitch and the energy predictor as ONE launch per layer, fs2_runtime.hip: run_predictors_fused):
    //   xp_row_chunks: 32-channel chunks per row of the A planes when the GEMM contracts only a slice of them (0: Cpad / 32);
    //   k_groups G > 1: the N outputs form G groups, group g contracts the chunks [g Cpad/32, (g+1) Cpad/32) of the plane row (a grouped conv);
    //   ln_groups G > 1: ReLU / LayerNorm / scalar head apply to each of the G column groups of a row separately (their parameters are
    //   stacked along N); the scalar head of group g goes to dot_out + g dot_gstride and uses dot_b[g]
    int xp_row_chunks, k_groups, ln_groups, dot_gstride;
    int yp_col_off;                        // column offset of this launch's outputs inside the rows of Yp (a layer that fills one group of a stacked plane buffer)
    const int* Rp;                         // device-driven layout: rows actually used (tiles at or beyond round_up(*Rp, 128) exit at once); nullptr: R
    // planes-only residual stream (gemm_row4.h: RES): the residual as the PLANES the producing launch wrote (residp_chunks 32-channel chunks per row;
    // residp_mx != 0: mx planes, whose e4m3 residual words carry the scale 1 / residp_scale = 2^(ka+11)), instead of fp32 rows (resid must then be
    // nullptr); a launch with Y == nullptr and Yp != nullptr writes planes only.  Only gemm_row4_bf16 implements both: launch_gemm refuses otherwise.
    const void* residp; int residp_chunks; int residp_mx; float residp_scale;      // residp_mx: 1 = mx planes (e4m3 residual at byte 2 C + c of the row), 2 = mx4 planes (at 3 C + c)
    // mx4 (gemm_planes.h: ARITH = 3; yp_f16 == 3 / mx == 2): one E8M0 scale byte per ROW of the activation planes (2^-11 folded in) -- written by the
    // producing LayerNorm epilogue (yp_rowscale), read by the conv (x_rowscale) -- and one per output channel of the weight image (w_rowscale)
    unsigned char* yp_rowscale; const unsigned char* x_rowscale; const unsigned char* w_rowscale;
};

__device__ __forceinline__ float wave16_sum(float v) {
    // sum over the 16 lanes that share lane>>4 (one MFMA row group)
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return tanhf(v);
    return v;
}

// MFMA lane -> tile-row permutation of the bf16 kernels.  ds_read_b128 serves a wave in four 16-lane groups
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...), each mixing two k-slot columns; with tile row = lane&15 the XOR-swizzled
// 128-byte-row layout has 2-way bank conflicts in most groups (brute-forced over all tap offsets: 2/3 of the reads).  Giving
// lanes 4-11 the even rows and lanes 0-3,12-15 the odd rows makes every group hit 16 distinct 16-byte bank slots for EVERY
// row offset (tap), because the two slot columns of a group then differ by the XOR of row bit 1 within closed row pairs.
// The same permutation applies to the B operand (output column) and therefore to the C/D element -> (row, col) map.
__device__ __forceinline__ int rperm(int i) { return i < 4 ? 2 * i + 1 : (i < 12 ? 2 * (i - 4) : 2 * (i - 12) + 9); }
__device__ __forceinline__ int rperm_inv(int x) { return (x & 1) ? (x < 8 ? (x - 1) >> 1 : ((x - 9) >> 1) + 12) : (x >> 1) + 4; }

// Split-bf16 activation planes: element (row, c) lives in chunk c>>5 at k-position p with kperm(p) == c&31 (gemm_bf16.h):
// 4 consecutive channels c..c+3 (c % 4 == 0) are 8 contiguous bytes of the hi half and 8 of the lo half (+64 B).
__device__ __host__ __forceinline__ size_t plane_byte(size_t row, int nchunks, int c) {
    return (row * nchunks + (c >> 5)) * 128 + (((c & 15) >> 2) << 4) + (((c >> 4) & 1) << 3);
}
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4(const f32x4 v, uint2& hi, uint2& lo) {
    bf16x4_t h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 hb = (__bf16)v[j];
        h[j] = hb;
        l[j] = (__bf16)(v[j] - (float)hb);
    }
    hi = *reinterpret_cast<uint2*>(&h);
    lo = *reinterpret_cast<uint2*>(&l);
}
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split4_f16(const f32x4 v, uint2& hi, uint2& lo) {
    f16x4_t h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = fminf(fmaxf(v[j], -65504.f), 65504.f);      // fp16 range (activations of this network are O(1); no inf / NaN lo parts)
        const _Float16 hb = (_Float16)x;
        h[j] = hb;
        l[j] = (_Float16)(x - (float)hb);
    }
    hi = *reinterpret_cast<uint2*>(&h);
    lo = *reinterpret_cast<uint2*>(&l);
}
// "mx" planes (gemm_mx.h): a row of C channels is 4C bytes = C/32 units of 128 B, the LDS row images of the conv loop:
//   [ fp16(a): C/64 units of 64 channels | ra8: C/128 units = e4m3((a - fp16(a)) 2^(ka+11)) | ah8: C/128 units = e4m3(fp16(a) 2^ka) ],
// channels in their natural order inside every unit (requires C % 128 == 0; nchunks = C/32 as for the split planes).
__device__ __forceinline__ unsigned pack_fp8x4(const f32x4 v) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[0], -448.f), 448.f), fminf(fmaxf(v[1], -448.f), 448.f), 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[2], -448.f), 448.f), fminf(fmaxf(v[3], -448.f), 448.f), w, true);
    return (unsigned)w;
}
__device__ __forceinline__ void store_planes4_mx(void* planes, size_t row, int nchunks, int c, const f32x4 v, float sa) {
    f16x4_t h;
    f32x4 hs, rs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = fminf(fmaxf(v[j], -65504.f), 65504.f);
        const _Float16 hb = (_Float16)x;
        h[j] = hb;
        hs[j] = (float)hb * sa;
        rs[j] = (x - (float)hb) * (sa * 2048.f);
    }
    const size_t C = (size_t)nchunks * 32;
    char* p = reinterpret_cast<char*>(planes) + row * (4 * C);
    *reinterpret_cast<uint2*>(p + 2 * c) = *reinterpret_cast<uint2*>(&h);
    *reinterpret_cast<unsigned*>(p + 2 * C + c) = pack_fp8x4(rs);
    *reinterpret_cast<unsigned*>(p + 3 * C + c) = pack_fp8x4(hs);
}
// E8M0 byte of the scale 2^e of a slice whose largest |fp16 value| is m: the OCP MX rule, e = floor(log2 m) - 2 (2 = e2m1's largest exponent), so that
// m / 2^e lies in [4, 8): the values of the slice's top quarter-binade (6, 8) saturate at 6, in exchange for one more binade at the small end than a
// scale that fits the maximum would leave (simulated both ways: tools/arith_sim_ffn_pareto.py, "variant" rows -- 9.7e-5 against 1.3e-4 on the mel).
// Clamped to [40, 200]: a slice of zeros gets a harmless scale instead of 2^-127 (whose reciprocal would turn 0 into NaN).
__host__ __device__ inline int mx4_scale_byte(float m) {
    const unsigned u = __builtin_bit_cast(unsigned, m);
    const int eb = (int)((u >> 23) & 0xff) - 2;
    return eb < 40 ? 40 : (eb > 200 ? 200 : eb);
}
__host__ __device__ inline float mx4_inv_scale(int eb) { return __builtin_bit_cast(float, (unsigned)(254 - eb) << 23); }      // 2^(127 - eb) = 1 / 2^(eb - 127)

// "mx4" planes (gemm_planes.h: ARITH = 3): [ fp16(a): C/64 units | cross units: per 4 channels c .. c + 3 the bytes (ra4 ra4 | ra4 ra4 | ah4 ah4 | ah4 ah4), e2m1 of
// (a - fp16(a)) 2^11 / s_row and of fp16(a) / s_row, at byte 2 C + c of the row | ra8 = e4m3((a - fp16(a)) 2^(ka+11)) at byte 3 C + c: the residual reader's ].
// inv_s = 1 / s_row (the row's own scale: the caller reduced the row maximum); sa = 2^ka (the static scale of the e4m3 residual, as in the mx planes).
__device__ __forceinline__ void store_planes4_mx4(void* planes, size_t row, int nchunks, int c, const f32x4 v, float sa, float inv_s) {
    f16x4_t h;
    f32x4 rs;
    float h4[4], r4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = fminf(fmaxf(v[j], -65504.f), 65504.f);
        const _Float16 hb = (_Float16)x;
        h[j] = hb;
        const float r = x - (float)hb;
        rs[j] = r * (sa * 2048.f);
        h4[j] = (float)hb * inv_s;
        r4[j] = r * 2048.f * inv_s;
    }
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, r4[0], r4[1], 1.f, 0);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, r4[2], r4[3], 1.f, 1);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, h4[0], h4[1], 1.f, 2);
    w = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(w, h4[2], h4[3], 1.f, 3);
    const size_t C = (size_t)nchunks * 32;
    char* p = reinterpret_cast<char*>(planes) + row * (4 * C);
    *reinterpret_cast<uint2*>(p + 2 * c) = *reinterpret_cast<uint2*>(&h);
    *reinterpret_cast<unsigned*>(p + 2 * C + c) = w;
    *reinterpret_cast<unsigned*>(p + 3 * C + c) = pack_fp8x4(rs);
}
// planes of 4 consecutive channels in the format `mode`: 0 split-bf16, 1 split-fp16, 2 mx (scale = 2^ka)
__device__ __forceinline__ void store_planes4m(void* planes, size_t row, int nchunks, int c, const f32x4 v, int mode, float scale);

__device__ __forceinline__ void store_planes4(void* planes, size_t row, int nchunks, int c, const f32x4 v, bool f16 = false) {
    uint2 hi, lo;
    if (f16) split4_f16(v, hi, lo); else split4(v, hi, lo);
    char* p = reinterpret_cast<char*>(planes) + plane_byte(row, nchunks, c);
    *reinterpret_cast<uint2*>(p) = hi;
    *reinterpret_cast<uint2*>(p + 64) = lo;
}

__device__ __forceinline__ void store_planes4m(void* planes, size_t row, int nchunks, int c, const f32x4 v, int mode, float scale) {
    if (mode == 2) store_planes4_mx(planes, row, nchunks, c, v, scale);
    else store_planes4(planes, row, nchunks, c, v, mode == 1);
}

// Elementwise epilogue of a (16 MT) x 64 wave tile held as acc[MT][4] (16 x 16 MFMA tiles, C layout col = l&15,
// row = 4*(l>>4) + reg) - the fp32 kernels.  All loads of a 16-row slab (row flags, residual) are issued before its stores
// and the pointers are __restrict__, so the compiler does not serialise a memory round trip per element behind
// possibly-aliasing stores (that cost ~25 us per workgroup before).
template <int MT>
__device__ __forceinline__ void tile_epilogue_64x64(const GemmArgs& a, f32x4 (&acc)[MT][4], int row_base, int col_base, int lr, int lg,
                                                    bool relu_first) {
    const float* __restrict__ biasp = a.bias;
    const float* __restrict__ residp = a.resid;
    const int* __restrict__ rpos = a.row_pos;
    float* __restrict__ Y = a.Y;
    float bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int col = col_base + nt * 16 + lr;
        bv[nt] = (biasp && col < a.N) ? biasp[col] : 0.f;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        bool inb[4], valid[4];
        float rv[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row_base + mt * 16 + lg * 4 + r;
            inb[r] = row < a.R;
            valid[r] = inb[r] && (rpos == nullptr || rpos[row] >= 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row_base + mt * 16 + lg * 4 + r;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = col_base + nt * 16 + lr;
                rv[r][nt] = (residp && inb[r] && col < a.N) ? residp[(size_t)row * a.ldr + col] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row_base + mt * 16 + lg * 4 + r;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int col = col_base + nt * 16 + lr;
                float v = acc[mt][nt][r] + bv[nt] + rv[r][nt];
                if (relu_first) v = fmaxf(v, 0.f);
                v = apply_act(v, a.act_post);
                if (inb[r] && col < a.N) Y[(size_t)row * a.ldy + col] = valid[r] ? v : 0.f;
            }
        }
    }
}

}  // namespace fs2
