// HBM-bound kernels of the path: row metadata, embedding + positional encoding, duration post-op,
// length regulator (prefix sum + expand), bucketize + embedding add, packed -> padded output copies.
// All of them move whole rows with 16-byte accesses, one wavefront per row.
#pragma once
#include "common.h"

namespace fs2 {

// row_pos / row_seq for a gapped packed layout (start[] ascending).
__global__ void build_row_meta(const int* start, const int* len, int B, int rpad, int* row_pos, int* row_seq) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rpad) return;
    int lo = 0, hi = B;   // first b with start[b] > row
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= row) lo = mid + 1; else hi = mid;
    }
    const int b = lo - 1;
    int pos = -1, seq = -1;
    if (b >= 0 && row - start[b] < len[b]) { pos = row - start[b]; seq = b; }
    row_pos[row] = pos;
    row_seq[row] = seq;
}

// Frame-side layout built on the device from the frame counts the duration scan left there (device-driven mode of
// fs2_decode: no host read-back).  Mirrors build_layout() / build_work_list() in fs2_runtime.hip exactly: 32-row aligned starts,
// kGap zero rows between utterances, attention work list = (utterance, 64-query block) items in eight interleaved per-XCD queues,
// utterances dealt longest-key-range first to the shortest queue, padding entries (-1, 0).
// One workgroup.  dims = {rows used, work list length, overflow flags, longest utterance, valid frames}; pcum[b] = valid frames of the
// utterances before b (row offset of utterance b in the packed output); 8 = kGap = kAttAlign, 128 = kAttBlk.
__global__ __launch_bounds__(1024) void frame_layout_dev(const int* olens, int B, int compat, int masked, int row_cap, int work_cap,
                                                         int lmax_cap, int pe_rows, int* start, int* len, int* klen, int* vlen,
                                                         int* rank_tmp, int* woff_tmp, int* pcum, int2* work, int* dims, int* status = nullptr, int gap = 8) {
    // The serial parts (row prefix with alignment, longest-first dealing to eight queues) run on one thread: their operands are staged
    // in LDS first -- from global memory every iteration was a dependent round trip.  Round 2: 53 -> 28 us per call at B = 64 (one LDS
    // atomic per wave instead of 1024 on one address; the dealing loop walks its operands front to back); what is left is seven
    // barrier phases that each drain thread 0's global stores, and the launch.
    constexpr int kStage = 4096;                      // utterances staged in LDS (beyond: the same code on the global arrays)
    __shared__ int s_len[kStage], s_vlen[kStage], s_order[kStage];
    __shared__ int s_max, s_min;
    const int tid = threadIdx.x;
    const bool staged = B <= kStage;
    if (tid == 0) { s_max = 0; s_min = 0x7fffffff; }
    __syncthreads();
    int mx = 0, mn = 0x7fffffff;
    for (int b = tid; b < B; b += 1024) { mx = max(mx, olens[b]); mn = min(mn, olens[b]); }
    for (int o = 32; o > 0; o >>= 1) { mx = max(mx, __shfl_xor(mx, o)); mn = min(mn, __shfl_xor(mn, o)); }
    if ((tid & 63) == 0) { atomicMax(&s_max, mx); atomicMin(&s_min, mn); }      // one LDS atomic per wave, not 1024 on one address
    __syncthreads();
    mx = s_max;
    for (int b = tid; b < B; b += 1024) {
        const int v = max(olens[b], 0);
        const int l = compat ? mx : v;
        vlen[b] = v;
        len[b] = l;
        klen[b] = compat ? (masked ? v : mx) : v;
        if (staged) { s_vlen[b] = v; s_len[b] = l; }
    }
    for (int i = tid; i < work_cap; i += 1024) work[i] = make_int2(-1, 0);
    __syncthreads();
    // position of utterance b in the dealing order: longer key ranges first, ties in batch order
    for (int b = tid; b < B; b += 1024) {
        const int k = compat ? (masked ? max(olens[b], 0) : mx) : max(olens[b], 0);
        int r = 0;
        for (int j = 0; j < B; ++j) {
            const int vj = staged ? s_vlen[j] : max(olens[j], 0);
            const int kj = compat ? (masked ? vj : mx) : vj;
            r += (kj > k) || (kj == k && j < b);
        }
        if (staged) s_order[r] = b; else rank_tmp[r] = b;
    }
    __syncthreads();
    __shared__ int s_row, s_frames;
    if (tid == 0) {
        int row = gap;
        int frames = 0;
        for (int b = 0; b < B; ++b) {
            row = (row + 7) & ~7;      // kAttAlign
            start[b] = row;
            row += (staged ? s_len[b] : len[b]) + gap;
            pcum[b] = frames;
            frames += staged ? s_vlen[b] : vlen[b];
        }
        s_row = row + 8; s_frames = frames;      // + kTailRows, as build_layout on the host
    }
    __syncthreads();
    // blocks of 128 queries (kAttBlk) per utterance IN DEALING ORDER, so that the serial loop below walks two arrays front to back instead of
    // chasing s_order[r] -> s_len[b] through LDS (two dependent round trips per utterance on one thread)
    if (staged)
        for (int r = tid; r < B; r += 1024) s_vlen[r] = (s_len[s_order[r]] + 127) >> 7;
    __syncthreads();
    if (tid == 0) {
        const int row = s_row, frames = s_frames;
        // (qlen is only ever indexed by unrolled constants: registers)
        int qlen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int depth = 0;
        for (int r = 0; r < B; ++r) {
            const int b = staged ? s_order[r] : rank_tmp[r];
            int j = 0, best = qlen[0];
#pragma unroll
            for (int t = 1; t < 8; ++t)
                if (qlen[t] < best) { best = qlen[t]; j = t; }
            woff_tmp[b] = best * 8 + j;           // list position of the utterance's first block; the next ones follow 8 apart
            const int add = staged ? s_vlen[r] : ((len[b] + 127) >> 7);
#pragma unroll
            for (int t = 0; t < 8; ++t) qlen[t] += (t == j) ? add : 0;
            depth = max(depth, best + add);
        }
        int ovf = 0;
        if (row > row_cap || depth * 8 > work_cap) ovf |= 1;
        if (mx > lmax_cap) ovf |= 2;
        if (mx > pe_rows) ovf |= 4;
        if (s_min < 0) ovf |= 16;           // FS2_OVF_BAD_ID: dur_scan marked an utterance whose phoneme ids leave [0, idim)
        else if (s_min <= 0) ovf |= 8;
        dims[0] = row; dims[1] = ovf ? 0 : depth * 8; dims[2] = ovf; dims[3] = mx; dims[4] = frames; dims[5] = 0; dims[6] = 0; dims[7] = 0;
        if (status) {     // the caller's copy of the same eight words
            status[0] = row; status[1] = ovf ? 0 : depth * 8; status[2] = ovf; status[3] = mx; status[4] = frames; status[5] = 0; status[6] = 0; status[7] = 0;
        }
    }
    __syncthreads();
    if (dims[2] != 0) {           // a capacity is too small: leave an empty layout (all rows are gap rows, no work) so that
        for (int b = tid; b < B; b += 1024) { start[b] = 8; len[b] = 0; klen[b] = 0; vlen[b] = 0; }    // nothing indexes out of range
        return;
    }
    for (int b = tid; b < B; b += 1024) {
        const int nq = (len[b] + 127) >> 7, o = woff_tmp[b];
        for (int q = 0; q < nq; ++q) work[o + 8 * q] = make_int2(b, q);
    }
}

// reduction_factor r > 1: the Postnet's row layout = the decoder's with every row expanded to r rows
__global__ void scale_layout(const int* start, const int* len, const int* vlen, const int* dims, int B, int r, int* start2, int* len2,
                             int* vlen2, int* dims2) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { start2[b] = start[b] * r; len2[b] = len[b] * r; vlen2[b] = vlen[b] * r; }
    if (b == 0 && dims != nullptr) {
        dims2[0] = dims[0] * r;      // rows in use
#pragma unroll
        for (int i = 1; i < 8; ++i) dims2[i] = dims[i];
    }
}

// h[row] = E[xs[b, t]] * xscale + alpha * pe[t]      (reference encoder.py:196, embedding.py:77-80,105-120)
__global__ __launch_bounds__(256) void embed_pe(const int64_t* xs, int Tmax, const float* E, int idim, int D,
                                                const float* pe, const float* alpha_p, float xscale,
                                                const int* row_pos, const int* row_seq, int R, float* h, void* planes = nullptr) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const int t = row_pos[row];
    float* dst = h + (size_t)row * D;
    if (t < 0) {
        for (int c = lane * 4; c < D; c += 256) {
            *reinterpret_cast<float4*>(dst + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (planes) store_planes4(planes, row, D / 32, c, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        return;
    }
    const int b = row_seq[row];
    int64_t id = (t < Tmax) ? xs[(size_t)b * Tmax + t] : 0;
    if (id < 0 || id >= idim) id = 0;
    const float alpha = alpha_p ? alpha_p[0] : 1.f;
    const float* e = E + (size_t)id * D;
    const float* p = pe + (size_t)t * D;
    for (int c = lane * 4; c < D; c += 256) {
        const float4 ev = *reinterpret_cast<const float4*>(e + c);
        const float4 pv = *reinterpret_cast<const float4*>(p + c);
        float4 o;
        o.x = ev.x * xscale + alpha * pv.x;
        o.y = ev.y * xscale + alpha * pv.y;
        o.z = ev.z * xscale + alpha * pv.z;
        o.w = ev.w * xscale + alpha * pv.w;
        *reinterpret_cast<float4*>(dst + c) = o;
        if (planes) store_planes4(planes, row, D / 32, c, f32x4{o.x, o.y, o.z, o.w});
    }
}

// d = clamp(round_half_even(exp(y) - 1), 0) as int64 (reference duration_predictor.py:77-81: torch.clamp(torch.round(xs.exp() -
// offset), min=0).long()).  rintf = round half to even (torch.round); +inf and values beyond the int64 range saturate.
__device__ __forceinline__ int64_t duration_from_log(float y) {
    const float f = fmaxf(rintf(expf(y) - 1.0f), 0.f);
    return (f >= 9.0e18f) ? INT64_MAX : (int64_t)f;      // (NaN: fmaxf returns 0)
}

__global__ void duration_kernel(const float* d_log, int64_t n, int64_t* d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = duration_from_log(d_log[i]);
}

// Duration post-op (reference duration_predictor.py:77-84): packed per-row log-durations ->
// padded [B,Tmax] outputs; d = clamp(round_half_even(exp(y) - 1), 0), pads -> 0.
__global__ void dur_finalize(const float* dlog_rows, const int* start, const int* vlen, int B, int Tmax,
                             float* d_log, int64_t* d_int, int64_t* d_int2 = nullptr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * Tmax) return;
    const int b = i / Tmax, t = i - b * Tmax;
    float y = 0.f;
    int64_t d = 0;
    if (t < vlen[b]) {
        y = dlog_rows[start[b] + t];
        d = duration_from_log(y);
    }
    if (d_log) d_log[i] = y;
    if (d_int) d_int[i] = d;
    if (d_int2) d_int2[i] = d;      // the caller's copy (saves a device-to-device memcpy launch)
}

// Per-utterance inclusive prefix sum of the durations actually used (reference length_regulator.py:60,
// 85-88: slice to ilen, an all-zero row becomes all ones).  One workgroup per utterance.
// alpha: the length regulator's speed control (length_regulator.py:57-59): d <- round_half_even(float(d) * alpha) when alpha != 1.
__device__ __forceinline__ int scaled_duration(int64_t d, float alpha) {
    if (d <= 0) return 0;
    if (alpha == 1.0f) return (int)d;
    const float f = rintf((float)d * alpha);
    return f > 0.f ? (int)f : 0;
}

// ids != nullptr (fs2_encode): an utterance whose row of xs (all Tmax positions, pads included) holds a phoneme id outside [0, idim) gets the frame count -1 -- the reference's
// torch.nn.Embedding raises for it (fastspeech.py:65-67, core/encoder.py:196); embed_pe reads row 0 instead of indexing out of range, and
// the marker reaches the caller with the frame counts it reads anyway (host-driven layout) or as FS2_OVF_BAD_ID in the status word of
// fs2_decode's device-driven layout (frame_layout_dev), whose outputs are then NaN-filled.
__global__ __launch_bounds__(256) void dur_scan(const int64_t* ds, int Tmax, const int* ilen, int* cum,
                                                int64_t* olens, int* olens32, float alpha = 1.0f, const int64_t* ids = nullptr, int idim = 0) {
    __shared__ int wsum[4];
    __shared__ int carry_s;
    __shared__ int bad_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = ilen[b];
    const int64_t* d = ds + (size_t)b * Tmax;
    if (tid == 0) bad_s = 0;
    __syncthreads();
    // pass 1: total
    int tot = 0, bad = 0;
    for (int t = tid; t < T; t += 256) tot += scaled_duration(d[t], alpha);
    // (ALL Tmax positions of the row, pads included: the reference's nn.Embedding indexes the whole padded xs, so an id outside [0, idim) in the pad
    //  region -- a -1 padding convention, say -- raises there too; round-5 advisor finding)
    if (ids != nullptr)
        for (int t = tid; t < Tmax; t += 256) { const int64_t id = ids[(size_t)b * Tmax + t]; bad |= (id < 0 || id >= idim) ? 1 : 0; }
    if (bad) bad_s = 1;
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    if (lane == 0) wsum[wave] = tot;
    __syncthreads();
    const int total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    const bool ones = (total == 0);
    __syncthreads();
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < T; base += 256) {
        const int t = base + tid;
        int v = (t < T) ? (ones ? 1 : scaled_duration(d[t], alpha)) : 0;
        int x = v;   // inclusive scan inside the wave
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        int off = carry_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (t < T) cum[(size_t)b * Tmax + t] = off + x;
        __syncthreads();
        if (tid == 255) carry_s = off + x;
        __syncthreads();
    }
    if (tid == 0) {
        const int tt = bad_s ? -1 : (ones ? T : total);
        if (olens) olens[b] = tt;
        if (olens32) olens32[b] = tt;
    }
}

// Length-regulator expand (reference length_regulator.py:90-95 + utils/util.py:91-104):
// out[row] = hs[tok_start[b] + idx], idx = #{i : cum[b,i] <= j}; rows beyond the utterance -> 0.
// One wavefront per output row.  If row_seq == nullptr the output is a dense [B, uniform_len] layout.
__global__ __launch_bounds__(256) void lr_expand(const float* hs, int D, const int* tok_start, const int* ilen,
                                                 const int* cum, int Tmax, const int* row_pos, const int* row_seq,
                                                 int uniform_len, const int* vlen, int R, float* out,
                                                 int* index_rows, void* planes = nullptr) {     // planes: also as split-bf16 planes (D / 32 chunks)
    // (the row as a scalar: everything up to the copy -- utterance, position, the binary search over the duration sums -- is then scalar
    //  loads and SALU, a chain of K$ round trips instead of vector-memory ones)
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    int b, j;
    if (row_seq) { b = row_seq[row]; j = row_pos[row]; } else { b = row / uniform_len; j = row - b * uniform_len; }
    float* dst = out + (size_t)row * D;
    int idx = -1;
    if (b >= 0 && j < vlen[b]) {
        const int* c = cum + (size_t)b * Tmax;
        int lo = 0, hi = ilen[b];   // first i with c[i] > j
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (c[mid] <= j) lo = mid + 1; else hi = mid;
        }
        idx = lo;
    }
    if (index_rows && lane == 0) index_rows[row] = idx;
    if (idx < 0) {
        for (int c4 = lane * 4; c4 < D; c4 += 256) {
            *reinterpret_cast<float4*>(dst + c4) = make_float4(0.f, 0.f, 0.f, 0.f);
            if (planes) store_planes4(planes, row, D / 32, c4, f32x4{0.f, 0.f, 0.f, 0.f});
        }
        return;
    }
    const float* src = hs + (size_t)(tok_start[b] + idx) * D;
    for (int c4 = lane * 4; c4 < D; c4 += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + c4);
        *reinterpret_cast<float4*>(dst + c4) = v;
        if (planes) store_planes4(planes, row, D / 32, c4, f32x4{v.x, v.y, v.z, v.w});
    }
}

// torch.bucketize(x, bins, right=False): first i with x <= bins[i]; NaN -> nb (the `!(b >= x)` form keeps
// torch's NaN behaviour).  reference variance_predictor.py:158,231
__device__ __forceinline__ int bucket_index(float x, const float* bins, int nb) {
    int lo = 0, hi = nb;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (!(bins[mid] >= x)) lo = mid + 1; else hi = mid;
    }
    return lo;
}

__global__ void bucketize_kernel(const float* x, int64_t n, const float* bins, int nb, int* idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = bucket_index(x[i], bins, nb);
}

// Variance adaptor tail (reference fastspeech.py:218-219 with the one-hot GEMMs collapsed to gathers):
// h[row] = (h[row] + Tp[qp]) + Te[qe], Tx[q][c] = fl(W[c][q] + b[c]) precomputed at weight load, which is
// bit-identical to one_hot(q) @ W^T + b.  e/p come from the caller ([B, stride], teacher forcing) or from
// the predictors (per packed row).  One wavefront per row.
__global__ __launch_bounds__(256) void bucket_embed(float* h, int D, const int* row_pos, const int* row_seq, int R,
                                                    const float* es, int es_stride, const float* ps, int ps_stride,
                                                    const float* e_rows, const float* p_rows,
                                                    const float* ebins, const float* pbins, int nb,
                                                    const float* Te, const float* Tp, int* qe_rows, int* qp_rows, void* planes = nullptr) {
    const int row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));      // (scalar: the two bin searches become scalar-load chains)
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const int j = row_pos[row];
    if (j < 0) {
        if (lane == 0) { if (qe_rows) qe_rows[row] = -1; if (qp_rows) qp_rows[row] = -1; }
        if (planes)
            for (int c = lane * 4; c < D; c += 256) store_planes4(planes, row, D / 32, c, f32x4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const int b = row_seq[row];
    const float e = es ? (j < es_stride ? es[(size_t)b * es_stride + j] : 0.f) : e_rows[row];
    const float p = ps ? (j < ps_stride ? ps[(size_t)b * ps_stride + j] : 0.f) : p_rows[row];
    const int qe = bucket_index(e, ebins, nb);
    const int qp = bucket_index(p, pbins, nb);
    if (lane == 0) { if (qe_rows) qe_rows[row] = qe; if (qp_rows) qp_rows[row] = qp; }
    float* dst = h + (size_t)row * D;
    const float* te = Te + (size_t)qe * D;
    const float* tp = Tp + (size_t)qp * D;
    for (int c = lane * 4; c < D; c += 256) {
        float4 v = *reinterpret_cast<const float4*>(dst + c);
        const float4 a = *reinterpret_cast<const float4*>(tp + c);
        const float4 g = *reinterpret_cast<const float4*>(te + c);
        v.x = (v.x + a.x) + g.x;
        v.y = (v.y + a.y) + g.y;
        v.z = (v.z + a.z) + g.z;
        v.w = (v.w + a.w) + g.w;
        *reinterpret_cast<float4*>(dst + c) = v;
        if (planes) store_planes4(planes, row, D / 32, c, f32x4{v.x, v.y, v.z, v.w});
    }
}

// packed rows [R, W] -> padded [B, Lout, W]; positions >= limit[b] are filled with `fill`.
template <typename T>
__global__ void unpack_rows(const T* src, int W, const int* start, const int* limit, int B, int Lout, T* dst, T fill) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * Lout * W;
    if (i >= total) return;
    const int w = (int)(i % W);
    const int64_t bj = i / W;
    const int j = (int)(bj % Lout), b = (int)(bj / Lout);
    dst[i] = (j < limit[b]) ? src[(size_t)(start[b] + j) * W + w] : fill;
}

// the same for float rows with W % 4 == 0 (the mel outputs: W = 80): 16-byte accesses, one (row, 4 channels) piece per thread
// ovf (device-driven layout): the overflow flags of the call; when set the outputs of the call do not exist -> NaN instead of zeros
__global__ void unpack_rows4(const float* src, int W4, const int* start, const int* limit, int B, int Lout, float* dst, const int* ovf = nullptr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * Lout * W4) return;
    const int w = (int)(i % W4);
    const int64_t bj = i / W4;
    const int j = (int)(bj % Lout), b = (int)(bj / Lout);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ovf && *ovf != 0) { const float q = __builtin_nanf(""); v = make_float4(q, q, q, q); }
    else if (j < limit[b]) v = reinterpret_cast<const float4*>(src)[(size_t)(start[b] + j) * W4 + w];
    reinterpret_cast<float4*>(dst)[i] = v;
}

// gapped packed rows -> dense packed rows (valid frames only, utterances back to back): dst row cum_scale cum[b] + j
// (ovf as in unpack_rows4; dst then holds R rows: the row capacity).  cum_scale: the reduction factor when the rows are mel frames
// (r per decoder frame) and cum[] counts decoder frames.
__global__ void pack_rows(const float* src, int W, const int* row_pos, const int* row_seq, const int* vlen, const int* cum, int R,
                          float* dst, const int* ovf = nullptr, int cum_scale = 1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int w4 = W / 4;
    if (i >= (int64_t)R * w4) return;
    if (ovf && *ovf != 0) { const float q = __builtin_nanf(""); reinterpret_cast<float4*>(dst)[i] = make_float4(q, q, q, q); return; }
    const int row = (int)(i / w4), c = (int)(i - (int64_t)row * w4) * 4;
    const int b = row_seq[row];
    if (b < 0) return;
    const int j = row_pos[row];
    if (j >= vlen[b]) return;
    *reinterpret_cast<float4*>(dst + ((size_t)cum[b] * cum_scale + j) * W + c) = *reinterpret_cast<const float4*>(src + (size_t)row * W + c);
}

// split-bf16 planes [R][nchunks x 128 B] -> fp32 rows [R][ld] (hi + lo, exact): what fs2_op_attention hands back when asked to run
// the attention kernels in the model's output form (FS2_OP_ATT_PLANES: the context leaves the kernel as planes only)
__global__ void planes_to_rows(const void* planes, int nchunks, int R, int D, float* dst, int ld) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int d4 = D / 4;
    if (i >= (int64_t)R * d4) return;
    const int row = (int)(i / d4), c = (int)(i - (int64_t)row * d4) * 4;
    const char* p = reinterpret_cast<const char*>(planes) + plane_byte((size_t)row, nchunks, c);
    const bf16x4_t h = *reinterpret_cast<const bf16x4_t*>(p), l = *reinterpret_cast<const bf16x4_t*>(p + 64);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)h[j] + (float)l[j];
    *reinterpret_cast<f32x4*>(dst + (size_t)row * ld + c) = v;
}

// device-driven layout: when the overflow flags (dims[2]) are set the outputs of the call are invalid -> fill them with NaN
__global__ void poison_on_overflow(const int* dims, float* a, int64_t na, float* b, int64_t nb, float* c, int64_t nc) {
    if (dims[2] == 0) return;
    const float q = __builtin_nanf("");
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = i0; i < na; i += stride) a[i] = q;
    for (int64_t i = i0; i < nb; i += stride) b[i] = q;
    for (int64_t i = i0; i < nc; i += stride) c[i] = q;
}

// [N, W] -> [W, N] through a 32 x 33 LDS tile (coalesced on both sides)
__global__ __launch_bounds__(256) void transpose_rows(const float* src, int64_t N, int W, float* dst) {
    __shared__ float t[32][33];
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    const int w0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int64_t n = n0 + r; const int w = w0 + tx;
        t[r][tx] = (n < N && w < W) ? src[n * W + w] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int w = w0 + r; const int64_t n = n0 + tx;
        if (w < W && n < N) dst[(int64_t)w * N + n] = t[tx][r];
    }
}

// ---- weight repacking (run once per load_state_dict) ----
// conv / linear weight [N][C][k] -> [Npad][k][Cpad], zero padded, optionally scaled per output channel by
// gamma / sqrt(var + eps) (eval-mode BatchNorm folded into the Postnet convs, reference modules.py:285-348).
// ldw: elements between consecutive output rows of the source (0: C; larger for a column slice of a wider Linear weight, k = 1).
__global__ void repack_weight(const float* w, int N, int C, int k, int Npad, int Cpad, const float* bn_g,
                              const float* bn_v, float bn_eps, float* out, int ldw = 0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)Npad * k * Cpad;
    if (i >= total) return;
    const int c = (int)(i % Cpad);
    const int tap = (int)((i / Cpad) % k);
    const int n = (int)(i / ((int64_t)Cpad * k));
    float v = 0.f;
    if (n < N && c < C) {
        v = w[((size_t)n * (ldw ? ldw : C) + c) * k + tap];
        if (bn_g) v *= bn_g[n] / sqrtf(bn_v[n] + bn_eps);
    }
    out[i] = v;
}

// folded BatchNorm bias: beta - mean * gamma / sqrt(var + eps)
__global__ void bn_fold_bias(const float* g, const float* b, const float* m, const float* v, float eps, int N, float* out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) out[n] = b[n] - m[n] * (g[n] / sqrtf(v[n] + eps));
}

// embedding table of a Linear applied to one-hot rows: T[q][c] = W[c][q] + b[c]
__global__ void onehot_table(const float* W, const float* bias, int D, int nbins, float* T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * nbins) return;
    const int c = i % D, q = i / D;
    T[i] = W[(size_t)c * nbins + q] + bias[c];
}

}  // namespace fs2
