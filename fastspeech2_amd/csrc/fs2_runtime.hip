// libfs2_hip.so -- runtime + C ABI (include/fs2.h) of the MI355X FastSpeech2 mel-generation path.
// Orchestrates the kernels over the gapped packed row layout: gemm_planes.h / attn_bf16.h (split-bf16 and bf16 modes, activations as
// planes), gemm_f32.h / attn_f32.h (exact fp32 mode), elementwise.h (HBM-bound steps, device-side layout).  Weight repacking at load
// time, workspace carving (no allocation inside a forward), host- or device-driven frame layout, per-launch hipEvent profiling.
// Replaces the tensor work of reference fastspeech.py:169-243 (`FeedForwardTransformer._forward`).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/fs2.h"
#include "attn_bf16.h"
#include "attn_w32.h"
#include "attn_f32.h"
#include "common.h"
#include "elementwise.h"
#include "gemm_bf16.h"
#include "gemm_planes.h"
#include "gemm_row4.h"
#include "gemm_mx.h"
#include "gemm_f32.h"

using namespace fs2;

namespace {

std::string g_create_error;

struct Gemm {          // one repacked Linear / Conv1d
    float* w = nullptr;      // [Npad][ktaps][Cpad]
    void* wb = nullptr;      // split-bf16 image [Npad][ktaps][Cpad/32][hi 32 | lo 32]
    void* wf = nullptr;      // the same image in fp16 (FFN w_1 only): operand of the two- / one-term arithmetic modes
    void* wm = nullptr;      // weight image of the "mx" mode (gemm_mx.h; FFN w_1 convolutions with C % 128 == 0 and N % 128 == 0) ...
    int kw = 0;              // ... and the exponent of its static scale: |w| 2^kw <= 448
    void* wm4 = nullptr;     // weight image of the "mx4" mode (gemm_planes.h ARITH = 3: both cross terms in e2m1) ...
    unsigned char* ws4 = nullptr;      // ... and its scale image: one E8M0 byte per 16-channel block of every weight row (gemm_mx.h: mx4_scale_image_bytes)
    float* bias = nullptr;   // [N] or null
    int N = 0, C = 0, Cpad = 0, ktaps = 1;
};
struct Layer {
    Gemm qkv, out, w1, w2;
    Gemm cat_x, cat_a;       // concat_after: concat_linear [D, 2D] split into its x half (carries the bias) and its attention half
    float *ln1g = nullptr, *ln1b = nullptr, *ln2g = nullptr, *ln2b = nullptr;
    int ka = 0;              // "mx" mode: exponent of the static scale of the FFN input (the LN1 output): |x| 2^ka <= 448 from the bound
                             // |LN(x)_c| <= sqrt(D) |gamma_c| + |beta_c|
    int kh = 0;              // ... and of the FFN hidden layer (w_2's input in the mx arithmetic, gemm_row4.h ARITH = 2): |h| 2^kh <= 448 from the bound
                             // |h_n| <= sum_{c,tap} |w_1[n,c,tap]| xmax + |b_1[n]|, xmax = the LayerNorm bound above (w2.wm != nullptr: the mode exists)
};
struct Predictor {
    std::vector<Gemm> conv;
    std::vector<float*> lng, lnb;
    float *lin_w = nullptr, *lin_b = nullptr;
};
// The energy and the pitch predictor read the same input (reference fastspeech.py:194-196,214-217): in the bf16 modes they run as ONE launch per
// layer.  Layer 0: the two convolutions stacked along N (2 x chans outputs over one A tile).  Layer 1: a grouped convolution (group g
// contracts the channels of predictor g's hidden layer) with both scalar heads.  Parameters stacked along N in the order energy | pitch.
struct FusedPredictors {
    bool ok = false;
    Gemm c0, c1;
    float *ln0g = nullptr, *ln0b = nullptr, *ln1g = nullptr, *ln1b = nullptr, *lin_w = nullptr, *lin_b = nullptr;
};
struct Stack {
    std::vector<Layer> layers;
    float* pe = nullptr; int pe_rows = 0;
    float* alpha = nullptr;
    int dk = 0, Dp = 0;                             // true head dim; attention width heads x padded head dim (= D when dk is a kernel size)
    bool pre_ln = false, concat = false;            // encoder.py:53-71: normalize_before / concat_after
    float *after_g = nullptr, *after_b = nullptr;   // after_norm (applied only when pre_ln, encoder.py:201-202)
};

// Split-K serves small batches (regime_rows <= kSplitRegime: beyond, the grids fill the chip anyway).  Its scratch holds up to 4 slabs of
// kSplitRows x 1024 floats, enough for every launch with R <= kSplitRows, so that the decision is a function of regime_rows alone (the same
// in the host- and the device-driven layout) as long as the row capacity stays below twice the estimate.
constexpr int kSplitRegime = 8192, kSplitRows = 16384;
constexpr int kXcds = 8;      // MI355X: 8 accelerator dies, workgroups of a launch are dealt to them round-robin

// A/B switches of the kernel choice.  Read from the environment ONCE, when the library is first used, and
// changed afterwards only through fs2_set_option(): the launch path never touches the environment.  -1 = automatic choice.
struct Options {
    int bm = -1;         // FS2_BM       tile height of gemm_pl_bf16 (64 | 128 | 256)
    int bal = 0;         // FS2_BAL      tall conv tiles: 0 = always 256 rows (default: interleaved A/B, profiles/r03_ab_conv_tile_balance.txt), 1 = height balanced over
                         //              whole rounds of the rows in use, 2 = of the row capacity
    int row8 = -1;       // FS2_ROW8     force (1) / forbid (0) the row-complete LayerNorm-fused k = 1 GEMM
    int qkv8 = -1;       // FS2_QKV8     force / forbid the 8-wave fused QKV projection
    int nosplitk = 0;    // FS2_NOSPLITK no split-K of the token-level k = 1 GEMMs
    int f32_rows = 0;    // FS2_F32_ROWS row-complete fp32 GEMM for LayerNorm-terminated ops
    int fuse_var = 1;    // FS2_FUSE_VAR the pitch and the energy predictor as one launch per layer (0: separate launches)
    int mt8 = -1;        // FS2_MT8      m-tiles per wave of the 8-wave row-complete kernels (2 | 3: 128 / 192-row workgroups)
    int op_att_planes = 0;   // FS2_OP_ATT_PLANES  fs2_op_attention (split-bf16 modes) takes the context from the kernels as planes, the model's form, and converts (tests)
    int qkv_split = -1;  // FS2_QKV_SPLIT  the Q, K and V passes of gemm_qkv8_bf16 as three workgroups per row tile (-1: by the round count)
    int w32 = -1;        // FS2_ATTN_W32 split-bf16 attention with 32 queries per wave (attn_w32.h): 0 never, 1 whenever the head dim allows, -1 by regime
    int row4 = -1;       // FS2_ROW4     the one-wave-per-SIMD row-complete kernel (gemm_row4.h) wherever gemm_row8_bf16 would run and it has the epilogue: 0 never, else yes
    int mt4 = -1;        // FS2_MT4      its m-tiles per wave (4 | 5: 128 / 160-row workgroups; -1: by the round count)
    int qkv4 = -1;       // FS2_QKV4     the fused QKV projection's passes on gemm_row4_bf16 (EPI 3) wherever gemm_qkv8_bf16 would run at D = 384: 0 never, else yes
    int ffn2_mx = 1;     // FS2_FFN2_MX  mix_mx mode: the second FFN GEMM in the mx arithmetic too, wherever gemm_row4_bf16 runs it (0: split-bf16 as in round 4)
    int post_mx = 1;     // FS2_POST_MX  mixed modes: the Postnet's 512 -> 512 convolutions in the mx arithmetic (0: split-bf16 as until round 5)
};
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// ---- the ISA-audit gate (round-5 advisor finding: the gate used to live in the ctypes binding only).  attn_w32 and gemm_row4_bf16 keep their
// accumulators in literal registers the compiler does not know to be live; fastspeech2_amd/_lib.py::build() audits the device
// assembly of the binary it ships and records the outcome in libfs2_hip.audit.json next to it, tied to the file's SHA-256.  The library looks for
// that record of ITSELF when it is first used: without a clean one the three switches below start at 0 for every consumer (ctypes, the TorchScript
// op, a C program linked against the ABI) and only an explicit fs2_set_option turns them on (probes).
struct Sha256 {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    unsigned char buf[64]; size_t fill = 0; uint64_t total = 0;
    static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    void block(const unsigned char* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u,
            0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
            0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
            0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
            0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u,
            0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] << 24 | (uint32_t)p[4 * i + 1] << 16 | (uint32_t)p[4 * i + 2] << 8 | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            const uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; ++i) {
            const uint32_t t1 = hh + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    void update(const unsigned char* p, size_t n) {
        total += n;
        while (n) {
            const size_t k = std::min(n, 64 - fill);
            memcpy(buf + fill, p, k); fill += k; p += k; n -= k;
            if (fill == 64) { block(buf); fill = 0; }
        }
    }
    std::string hex16() {      // first 16 hex digits of the digest (what _lib.py records as so_sha16)
        const uint64_t bits = total * 8;
        unsigned char pad[72] = {0x80};
        const size_t padn = (fill < 56 ? 56 - fill : 120 - fill);
        unsigned char len[8];
        for (int i = 0; i < 8; ++i) len[i] = (unsigned char)(bits >> (56 - 8 * i));
        update(pad, padn); update(len, 8);
        char out[17];
        snprintf(out, sizeof out, "%08x%08x", h[0], h[1]);
        return out;
    }
};
bool audit_clean_of_this_binary() {
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void*>(&fs2_abi_version), &info) || !info.dli_fname) return false;
    const std::string so = info.dli_fname;
    const size_t slash = so.rfind('/');
    const std::string rec_path = (slash == std::string::npos ? std::string() : so.substr(0, slash + 1)) + "libfs2_hip.audit.json";
    std::ifstream rf(rec_path);
    if (!rf) return false;
    const std::string rec((std::istreambuf_iterator<char>(rf)), std::istreambuf_iterator<char>());
    const size_t k = rec.find("\"so_sha16\"");
    if (k == std::string::npos) return false;
    const size_t q0 = rec.find('"', rec.find(':', k) + 1), q1 = q0 == std::string::npos ? q0 : rec.find('"', q0 + 1);
    if (q1 == std::string::npos) return false;
    const std::string want = rec.substr(q0 + 1, q1 - q0 - 1);
    const size_t c = rec.find("\"clean\"");
    const size_t v = c == std::string::npos ? c : rec.find_first_not_of(" :", c + 7);
    if (v == std::string::npos || rec.compare(v, 4, "true") != 0) return false;
    std::ifstream sf(so, std::ios::binary);
    if (!sf) return false;
    Sha256 sh;
    std::vector<unsigned char> blk(1 << 20);
    while (sf) {
        sf.read(reinterpret_cast<char*>(blk.data()), (std::streamsize)blk.size());
        if (sf.gcount() > 0) sh.update(blk.data(), (size_t)sf.gcount());
    }
    return sh.hex16() == want;
}
bool audit_clean() {
    static const bool ok = audit_clean_of_this_binary();
    return ok;
}

Options& opts() {
    static Options o = [] {
        Options x;
        x.bm = env_int("FS2_BM", -1); x.row8 = env_int("FS2_ROW8", -1); x.qkv8 = env_int("FS2_QKV8", -1);
        x.nosplitk = env_int("FS2_NOSPLITK", 0) != 0; x.f32_rows = env_int("FS2_F32_ROWS", 0) != 0; x.mt8 = env_int("FS2_MT8", -1); x.qkv_split = env_int("FS2_QKV_SPLIT", -1); x.op_att_planes = env_int("FS2_OP_ATT_PLANES", 0); x.fuse_var = env_int("FS2_FUSE_VAR", 1); x.bal = env_int("FS2_BAL", 0); x.w32 = env_int("FS2_ATTN_W32", -1);
        x.row4 = env_int("FS2_ROW4", -1); x.mt4 = env_int("FS2_MT4", -1); x.ffn2_mx = env_int("FS2_FFN2_MX", 1); x.qkv4 = env_int("FS2_QKV4", -1); x.post_mx = env_int("FS2_POST_MX", 1);
        if (!audit_clean()) {
            x.w32 = 0; x.row4 = 0; x.qkv4 = 0;
            fprintf(stderr, "libfs2_hip: no clean ISA-audit record of this binary (libfs2_hip.audit.json next to it): attn_w32 and gemm_row4_bf16 are switched "
                            "off, attn_bf16 / gemm_row8_bf16 run instead; rebuild with `python -c 'import __graft_entry__ as g; g.build()'`\n");
        }
        return x;
    }();
    return o;
}

// The attention kernels exist for head dims 64 / 128 / 192 / 256.  Any other adim / aheads the reference accepts (core/attention.py:18-20)
// runs with the head dim zero-padded to the next of these: the Q / K / V projection weights get zero rows, the output projection zero
// columns, at load time, so scores and context are unchanged (the softmax scale keeps the true d_k); the attention buffers are
// heads x padded wide.  0: unsupported (d_k > 256).
inline int padded_head_dim(int dk) { return dk <= 64 ? 64 : (dk <= 128 ? 128 : (dk <= 192 ? 192 : (dk <= 256 ? 256 : 0))); }
inline int att_width(int D, int heads) { return heads * padded_head_dim(D / heads); }

// Mixed modes: everything as bf16x3 except the FFN convolution w_1, which runs on fp16 operands with 2 or 1 MFMA per fragment pair.
inline int base_precision(int p) { return (p == FS2_PREC_MIX_F16X2 || p == FS2_PREC_MIX_F16X1 || p == FS2_PREC_MIX_MX || p == FS2_PREC_MIX_MX4) ? FS2_PREC_BF16X3 : p; }
constexpr int kFfnMx = 9;      // value of "ffn_terms" that selects the fp16 + block-scaled-fp8 arithmetic (gemm_mx.h)
constexpr int kPostKa = 8;     // static scale exponent of the Postnet's mx operands: tanh outputs, |x| <= 1 -> 2^8 = 256 <= 448
constexpr int kFfnMx4 = 10;    // ... the fp16 + block-scaled-fp4 arithmetic where the planes-only regime offers it (gemm_planes.h ARITH = 3), kFfnMx's elsewhere
inline int ffn_f16_terms(int p) { return p == FS2_PREC_MIX_F16X2 ? 2 : (p == FS2_PREC_MIX_F16X1 ? 1 : (p == FS2_PREC_MIX_MX ? kFfnMx : (p == FS2_PREC_MIX_MX4 ? kFfnMx4 : 0))); }
inline int scale_byte4(int e) { const int b = std::min(std::max(e, 1), 254); return b * 0x01010101; }

// max over the rows n of a [N][K] weight of (sum_k |w[n][k]|) xmax + |b[n]|: the a-priori bound of relu(w x + b) for |x| <= xmax (weight-load time only)
__global__ void row_l1_bound_kernel(const float* w, const float* b, int N, int K, float xmax, unsigned* out) {
    const int n = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) s += fabsf(w[(size_t)n * K + k]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint((part[0] + part[1] + part[2] + part[3]) * xmax + (b ? fabsf(b[n]) : 0.f)));
}
float device_row_l1_bound(hipStream_t s, const float* w, const float* b, int N, int K, float xmax, unsigned* scratch) {
    hipMemsetAsync(scratch, 0, 4, s);
    hipLaunchKernelGGL(row_l1_bound_kernel, dim3(N), dim3(256), 0, s, w, b, N, K, xmax, scratch);
    unsigned u = 0;
    hipMemcpyAsync(&u, scratch, 4, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// max |x| of a device array (weight-load time only: one small launch + a blocking 4-byte read-back)
__global__ void absmax_kernel(const float* x, int64_t n, unsigned* out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(x[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));      // non-negative floats order like their bit patterns
}
float device_absmax(hipStream_t s, const float* x, int64_t n, unsigned* scratch) {
    hipMemsetAsync(scratch, 0, 4, s);
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 1024)), dim3(256), 0, s, x, n, scratch);
    unsigned bits = 0;
    hipMemcpyAsync(&bits, scratch, 4, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    float f; memcpy(&f, &bits, 4);
    return f;
}
inline int fp8_scale_exponent(float bound) { return bound > 0.f ? (int)std::floor(std::log2(448.0 / (double)bound)) : 0; }

// Every entry point that takes a handle runs on the handle's device and leaves the caller's current device as it found it.
struct DeviceGuard {
    int prev = -1; bool switched = false; hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) { err = hipSetDevice(dev); switched = (err == hipSuccess); }
    }
    ~DeviceGuard() { if (switched) hipSetDevice(prev); }
};

struct HostLayout {
    int B = 0, R = 0, Rpad = 0;
    std::vector<int> start, len, klen, vlen;
    std::vector<int2> work;
    int work_cap = -1;      // device-driven layout: R / Rpad / work_cap are capacities, the vectors stay empty
    int nwork() const { return work_cap >= 0 ? work_cap : (int)work.size(); }
};
struct DevLayout {
    int *start = nullptr, *len = nullptr, *klen = nullptr, *vlen = nullptr, *row_pos = nullptr, *row_seq = nullptr;
    int2* work = nullptr;
    int* dims = nullptr;    // device-driven layout: {rows used, work items, overflow flags, longest utterance, valid frames, 0, 0, 0}
    int* pcum = nullptr;    // device-driven layout: valid frames before each utterance (packed-output row offsets)
};

struct ProfRec { std::string name; hipEvent_t e0, e1; double flops, bytes; };

}  // namespace

struct fs2_handle {
    fs2_config cfg;
    std::string err;
    bool loaded = false;
    std::vector<void*> allocs;
    // weights
    float* enc_embed = nullptr;
    Stack enc, dec;
    Predictor dur, energy, pitch;
    FusedPredictors var2;
    float *ebins = nullptr, *pbins = nullptr, *Te = nullptr, *Tp = nullptr;
    Gemm dec_in; float *dec_in_lng = nullptr, *dec_in_lnb = nullptr;
    Gemm feat;
    std::vector<Gemm> post;
    // state carried from encode to decode
    bool encoded = false;
    HostLayout tok;
    DevLayout dtok;
    float* enc_final = nullptr;
    int* cum = nullptr;
    int* o32 = nullptr;        // device frame counts (int32) left by the duration scan
    int enc_B = 0, enc_Tmax = 0, enc_compat = 0;
    long enc_ntok = 0;         // phonemes of the encoded batch (basis of the frame-level kernel-variant choice)
    float* kp = nullptr; size_t kp_cap = 0;   // split-K scratch of the call in progress (carved from its workspace)
    int cur_regime = 0;        // regime_rows of the call in progress (0: the launch's own row count)
    int* counters = nullptr;   // device int[16], zeroed at creation: [0] = waves of attn_w32 that left the fast path (fs2_get_counter)
    int* cur_status = nullptr; // device-driven layout: the status words of the call in progress ([5] counts the same event for that call alone)
    int gap = kGap;            // zero rows between packed utterances: max(kMinGap, largest conv halo of this model)
    void* enc_ws = nullptr;
    std::vector<void*> graph_pinned;   // host staging owned by captured graphs (see upload_layout)
    // profiling
    bool prof = false;
    std::string prof_filter;      // when non-empty only launches with exactly this name are bracketed
    std::vector<ProfRec> recs;
};

namespace {

int fail(fs2_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(h, FS2_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int round_up(int x, int a) { return (x + a - 1) / a * a; }

struct Bump {   // carve a caller-provided workspace
    char* base; size_t off = 0, cap;
    Bump(void* p, size_t c) : base((char*)p), cap(c) {}
    template <typename T> T* take(size_t n) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base + off);
        off += n * sizeof(T);
        return p;
    }
    bool ok() const { return off <= cap; }
};

// ------------------------------------------------------------------ profiling wrappers
struct Scope {
    fs2_handle* h; hipStream_t s; size_t idx = (size_t)-1;
    Scope(fs2_handle* h_, hipStream_t s_, const char* name, double flops, double bytes) : h(h_), s(s_) {
        if (h && h->prof && (h->prof_filter.empty() || h->prof_filter == name)) {
            ProfRec r; r.name = name; r.flops = flops; r.bytes = bytes;
            hipEventCreate(&r.e0); hipEventCreate(&r.e1);
            hipEventRecord(r.e0, s);
            h->recs.push_back(r); idx = h->recs.size() - 1;
        }
    }
    ~Scope() { if (idx != (size_t)-1) hipEventRecord(h->recs[idx].e1, s); }
};

// Kernels that need more than 64 KB of dynamic LDS must raise their limit once per (kernel, device); the flags live at the
// call site (one array per kernel instantiation), indexed by the current device.
struct LdsAttr { bool done[32] = {}; };
inline void allow_lds(const void* kernel, size_t bytes, LdsAttr& st) {
    int dev = 0;
    hipGetDevice(&dev);
    bool& d = st.done[dev & 31];
    if (!d) { hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); d = true; }
}

// ------------------------------------------------------------------ kernel launchers
template <int NT>
hipError_t launch_rows(hipStream_t s, const GemmArgs& a) {
    static LdsAttr attr;
    allow_lds(reinterpret_cast<const void*>(&gemm_rows_f32<NT>), rows_lds_bytes<NT>(), attr);
    dim3 grid((a.R + kRowsBM - 1) / kRowsBM);
    hipLaunchKernelGGL(gemm_rows_f32<NT>, grid, dim3(256), rows_lds_bytes<NT>(), s, a);
    return hipGetLastError();
}

hipError_t launch_tile(hipStream_t s, const GemmArgs& a) {
    static LdsAttr attr;
    allow_lds(reinterpret_cast<const void*>(&gemm_tile_f32), kTileLds, attr);
    dim3 grid((a.R + kTileBM - 1) / kTileBM, (a.N + kTileBN - 1) / kTileBN);
    hipLaunchKernelGGL(gemm_tile_f32, grid, dim3(256), kTileLds, s, a);
    return hipGetLastError();
}

bool rows_supported(int N) { return N == 80 || N == 256 || N == 384; }

template <int NSPLIT, int BM, bool K1, int ARITH = 0>
hipError_t launch_pl_t(hipStream_t s, const GemmArgs& a) {
    static LdsAttr attr;
    constexpr size_t lds = pl_lds_bytes<BM, K1>();
    allow_lds(reinterpret_cast<const void*>(&gemm_pl_bf16<NSPLIT, BM, K1, ARITH>), lds, attr);
    dim3 grid((a.N + kB16BN - 1) / kB16BN, ((a.qk_hi ? a.Rvt : a.R) + BM - 1) / BM, a.ksplit > 1 ? a.ksplit : 1);
    hipLaunchKernelGGL((gemm_pl_bf16<NSPLIT, BM, K1, ARITH>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}
// Tile height of the 8-wave row-complete kernels (one workgroup per CU): 64 MT rows, MT = 2 or 3.  A launch of T tiles takes ceil(T / #CUs)
// rounds, and a nearly empty second round costs as much as a full one (c3 at 7.87 frames per phoneme: 286 tiles of 128 rows = 256 + 30).
// Same-box A/B: a 192-row tile costs 1.55-1.9 x a 128-row one (its epilogue spills), so it pays exactly when it turns two rounds into one
// (c3: dec.ffn2_ln 0.174 -> 0.134 ms, step 5.68 -> 5.34 ms; c4, 15 rounds against 10: 68.0 -> 70.8 ms, so not there).  Results do not depend on MT.
constexpr int kCus = 256;
// Rows a launch will really touch: in the device-driven layout a.R is a capacity (15-25 % above the rows in use, the surplus tiles exit at once);
// the regime estimate (8 frames per phoneme + alignment rows, the same number in both layout modes) is the better basis for balancing rounds.
inline long rows_in_use(const GemmArgs& a, long rows) { return a.regime_rows > 0 ? std::min<long>(rows, a.regime_rows) : rows; }
inline int row8_mt(long rows) {
    const long t128 = (rows + 127) / 128, t192 = (rows + 191) / 192;
    return (t128 > kCus && t192 <= kCus) ? 3 : 2;
}

template <int NSPLIT, int NB, int MT>
hipError_t launch_row8_t(hipStream_t s, const GemmArgs& a) {
    static LdsAttr attr;
    constexpr size_t lds = row8_lds_bytes<NB, MT>();
    allow_lds(reinterpret_cast<const void*>(&gemm_row8_bf16<NSPLIT, NB, MT>), lds, attr);
    hipLaunchKernelGGL((gemm_row8_bf16<NSPLIT, NB, MT>), dim3((a.R + 64 * MT - 1) / (64 * MT)), dim3(512), lds, s, a);
    return hipGetLastError();
}
template <int NSPLIT, int NB>
hipError_t launch_row8(hipStream_t s, const GemmArgs& a) {
    const int mt = opts().mt8 > 0 ? opts().mt8 : row8_mt(rows_in_use(a, a.R));
    if (mt >= 3) return launch_row8_t<NSPLIT, NB, 3>(s, a);
    return launch_row8_t<NSPLIT, NB, 2>(s, a);
}

// gemm_row4_bf16 (gemm_row4.h): one wave per SIMD, 128- or 160-row workgroups, one per CU.  The tile height that minimises rounds x height
// (ties: the taller tile -- fewer weight bytes per row): c3 (36.6 k rows) 160 rows = 229 workgroups in one round, the c5 shard (78 k rows) 160 rows =
// 2 rounds instead of 3, c4 either.  Results do not depend on the height (nor on the kernel: bit-identical to gemm_row8_bf16).
inline int row4_mt(long rows) {
    const long r4 = ((rows + 127) / 128 + kCus - 1) / kCus * 128, r5 = ((rows + 159) / 160 + kCus - 1) / kCus * 160;
    return r5 <= r4 ? 5 : 4;
}
// the epilogues it has (gemm_row4.h: EPI); -1: none, the launch stays on gemm_row8_bf16.  Since round 6 the kernel serves the PLANES-ONLY form of
// these launches only (gemm_row4.h: RES): no fp32 rows out (Y == nullptr, Yp given), the residual -- where there is one -- as the producing launch's
// planes (residp; resid == nullptr).  fs2_decode / run_stack build their arguments in that form exactly when planes_only_regime() says the
// decoder's LayerNorm-fused launches will run here; a launch in the other form stays on gemm_row8_bf16.
inline int row4_epi(const GemmArgs& a) {
    if (a.ktaps != 1 || a.N != 384 || !a.ln_g || a.Y || !a.Yp || a.resid || a.relu_pre || a.dot_w || a.k_groups > 1 || a.ln_groups > 1 || a.qk_hi || a.yp_col_off) return -1;
    if (a.Cpad % 64 != 0 || a.yp_chunks * 32 != a.N) return -1;      // an even number of k-steps; planes exactly N wide
    if (a.residp && a.residp_chunks * 32 != a.N) return -1;
    if (a.pe) return (a.act_post == 1 && a.yp_f16 == 0 && !a.residp) ? 2 : -1;
    if (a.act_post != 0 || !a.residp) return -1;
    if (a.yp_f16 == 2) return a.residp_mx ? -1 : 1;      // out-proj + LN1 of mix_mx: residual = split-bf16 planes of the block input, result = mx planes
    if (a.yp_f16 == 3) return (a.residp_mx || !a.yp_rowscale) ? -1 : 4;      // ... of mix_mx4: result = mx4 planes + one scale byte per row
    return a.yp_f16 == 0 ? 0 : -1;
}
template <int MT, int EPI, int ARITH, int RES>
hipError_t launch_row4_t(hipStream_t s, const GemmArgs& a) {
    static LdsAttr attr;
    constexpr size_t lds = row4_lds_bytes<3, MT>();
    allow_lds(reinterpret_cast<const void*>(&gemm_row4_bf16<3, 3, MT, EPI, 2, ARITH, RES>), lds, attr);
    hipLaunchKernelGGL((gemm_row4_bf16<3, 3, MT, EPI, 2, ARITH, RES>), dim3((a.R + 32 * MT - 1) / (32 * MT)), dim3(256), lds, s, a);
    return hipGetLastError();
}
template <int EPI, int ARITH, int RES>
hipError_t launch_row4_e(hipStream_t s, const GemmArgs& a) {
    const int mt = (opts().mt4 == 4 || opts().mt4 == 5) ? opts().mt4 : row4_mt(rows_in_use(a, a.R));
    return mt == 5 ? launch_row4_t<5, EPI, ARITH, RES>(s, a) : launch_row4_t<4, EPI, ARITH, RES>(s, a);
}
// (the DMA-spreading schedule is fixed at SCHED = 2: same-box A/B in the model, c3: 0 -> 7.21, 2 -> 7.20, 4 -> 7.17 M frames/s; stand-alone 2 and 4 are 5-13 % ahead of 0)
// The instantiations the library holds (fastspeech2_amd/_audit.py: EXPECTED_KERNELS counts them): EPI 0 x {split-bf16 arithmetic with the residual from
// split-bf16 or from mx planes, mx arithmetic with the residual from mx planes}, EPI 1 and EPI 4 (residual from split-bf16 planes), EPI 2 (no residual),
// each at two tile heights, + the QKV passes (launch_qkv4_t).
hipError_t launch_row4(hipStream_t s, const GemmArgs& a, int epi) {
    if (epi == 2) return launch_row4_e<2, 0, 3>(s, a);
    if (epi == 1) return launch_row4_e<1, 0, 1>(s, a);
    if (epi == 4) return launch_row4_e<4, 0, 1>(s, a);
    if (a.mx) return a.residp_mx ? launch_row4_e<0, 2, 2>(s, a) : hipErrorInvalidValue;      // (FFN2 in the mx arithmetic exists in mix_mx only, where LN1's output is mx planes)
    return a.residp_mx ? launch_row4_e<0, 0, 2>(s, a) : launch_row4_e<0, 0, 1>(s, a);
}
// Will the decoder's LayerNorm-fused k = 1 launches (out-proj + LN1, FFN2 + LN2, the input layer) run on gemm_row4_bf16 -- i.e. do its activations
// travel as planes ONLY?  Same predicates as use_row8 / launch_gemm, asked once per fs2_decode so that every site of the stack agrees.
inline bool planes_only_regime(const fs2_config& c, const Stack& st, int prec, long regime_rows) {
    if (prec != FS2_PREC_BF16X3 || opts().row4 == 0 || c.ddim != 384 || st.pre_ln || st.concat || c.dunits % 64 != 0 || c.adim % 64 != 0) return false;
    if (opts().row8 >= 0) return opts().row8 != 0;
    return (regime_rows + 127) / 128 >= 128;
}

template <int NSPLIT, int NB, int MT, int GROUPS = 1>
hipError_t launch_row8c_t(hipStream_t s, const GemmArgs& a) {
    static LdsAttr attr;
    constexpr size_t lds = row8c_lds_bytes<NB, MT>();
    allow_lds(reinterpret_cast<const void*>(&gemm_row8c_bf16<NSPLIT, NB, MT, GROUPS>), lds, attr);
    hipLaunchKernelGGL((gemm_row8c_bf16<NSPLIT, NB, MT, GROUPS>), dim3((a.R + 64 * MT - 1) / (64 * MT), a.k_groups > 1 ? a.k_groups : 1), dim3(512), lds, s, a);
    return hipGetLastError();
}
template <int NSPLIT, int NB>
hipError_t launch_row8c(hipStream_t s, const GemmArgs& a) {
    const int mt = opts().mt8 > 0 ? opts().mt8 : row8_mt(rows_in_use(a, a.R));
    if (mt >= 3) return launch_row8c_t<NSPLIT, NB, 3>(s, a);
    return launch_row8c_t<NSPLIT, NB, 2>(s, a);
}

template <int NSPLIT, int NB, int MT, bool APART>
hipError_t launch_qkv8_t(hipStream_t s, const GemmArgs& a) {
    static LdsAttr attr;
    constexpr size_t lds = qkv8_lds_bytes<NB, MT>();
    allow_lds(reinterpret_cast<const void*>(&gemm_qkv8_bf16<NSPLIT, NB, MT, APART>), lds, attr);
    hipLaunchKernelGGL((gemm_qkv8_bf16<NSPLIT, NB, MT, APART>), dim3((a.Rvt + 64 * MT - 1) / (64 * MT), APART ? 3 : 1), dim3(512), lds, s, a);
    return hipGetLastError();
}
// The Q, K and V passes of a row tile are independent (each re-streams the A tile): as three workgroups per tile (grid.y = 3) the
// unit of work is a third of a tile and the last, partly filled round of a launch costs a third.  Rounds in units of a 128-row
// tile's three passes, a 192-row tile at 1.7 (row8_mt): c3, 286 tiles of 128 rows: whole tiles 2.0 (128) / 1.7 (192, one round
// on 191 of 256 CUs: what ran until round 4), passes apart 4/3 (128) / 1.7 (192).  Never more rounds than whole tiles of the
// same height.  Results do not depend on either choice.
inline void qkv8_plan(long rows, int& mt, int& apart) {
    double best = 1e30;
    for (int m = 2; m <= 3; ++m)
        for (int ap = 0; ap <= 1; ++ap) {
            if (opts().mt8 > 0 && m != std::min(std::max(opts().mt8, 2), 3)) continue;
            if (opts().qkv_split >= 0 && ap != (opts().qkv_split != 0)) continue;
            const long tiles = (rows + 64 * m - 1) / (64 * m), units = ap ? 3 * tiles : tiles;
            const double cost = (double)((units + kCus - 1) / kCus) / (ap ? 3.0 : 1.0) * (m == 3 ? 1.7 : 1.0);
            if (cost < best - 1e-9) { best = cost; mt = m; apart = ap; }
        }
}
template <int NSPLIT, int NB>
hipError_t launch_qkv8(hipStream_t s, const GemmArgs& a) {
    int mt = 2, apart = 0;
    qkv8_plan(rows_in_use(a, a.Rvt), mt, apart);
    if (mt >= 3) return apart ? launch_qkv8_t<NSPLIT, NB, 3, true>(s, a) : launch_qkv8_t<NSPLIT, NB, 3, false>(s, a);
    return apart ? launch_qkv8_t<NSPLIT, NB, 2, true>(s, a) : launch_qkv8_t<NSPLIT, NB, 2, false>(s, a);
}

// The fused QKV projection's passes on the one-wave-per-SIMD structure (gemm_row4.h, EPI 3): always one workgroup per (row tile, pass); the tile height that
// minimises rounds x height of the 3 T pass-workgroups (c3: 160 rows = 687 units = 2.7 rounds against 858 = 3.4 rounds of 128 rows).  Bit-identical to gemm_qkv8_bf16.
inline int qkv4_mt(long rows) {
    const long r4 = (3 * ((rows + 127) / 128) + kCus - 1) / kCus * 128, r5 = (3 * ((rows + 159) / 160) + kCus - 1) / kCus * 160;
    return r5 <= r4 ? 5 : 4;
}
template <int MT>
hipError_t launch_qkv4_t(hipStream_t s, const GemmArgs& a) {
    static LdsAttr attr;
    constexpr size_t lds = row4_lds_bytes<3, MT>();
    static_assert((size_t)32 * MT * kQkvLd * 4 <= row4_lds_bytes<3, MT>(), "the V pass transposes its tile through the operand ring's memory");
    allow_lds(reinterpret_cast<const void*>(&gemm_row4_bf16<3, 3, MT, 3, 2, 0>), lds, attr);
    hipLaunchKernelGGL((gemm_row4_bf16<3, 3, MT, 3, 2, 0>), dim3((a.Rvt + 32 * MT - 1) / (32 * MT), 3), dim3(256), lds, s, a);
    return hipGetLastError();
}
hipError_t launch_qkv4(hipStream_t s, const GemmArgs& a) {
    const int mt = (opts().mt4 == 4 || opts().mt4 == 5) ? opts().mt4 : qkv4_mt(rows_in_use(a, a.Rvt));
    return mt == 5 ? launch_qkv4_t<5>(s, a) : launch_qkv4_t<4>(s, a);
}

// Fused QKV projection on the 8-wave structure when there is about a CU's worth of 128-row tiles (FS2_QKV8=0|1 forces the choice)
bool use_qkv8(const GemmArgs& a) {
    if (!a.qk_hi || a.ktaps != 1 || (a.att_D != 256 && a.att_D != 384) || a.N != 3 * a.att_D) return false;
    if (opts().qkv8 >= 0) return opts().qkv8 != 0;
    return ((a.regime_rows ? a.regime_rows : a.Rvt) + 127) / 128 >= 128;
}

// Row-complete LN-fused kernel (gemm_row8_bf16) for k = 1 GEMMs that end in a row epilogue: one workgroup per CU, so it
// needs about a CU's worth of 128-row tiles to pay (FS2_ROW8=0|1 forces the choice).
bool use_row8(const GemmArgs& a) {
    const int Ng = a.k_groups > 1 ? a.N / a.k_groups : a.N;      // (grouped conv: one workgroup row per group, gemm_row8c_bf16's grid.y)
    const bool two_ln_groups = a.ln_groups == 2 && a.k_groups <= 1 && a.N == 512 && a.ktaps > 1;      // two stacked 256-channel layers over one input
    if (a.qk_hi || (!two_ln_groups && Ng != 256 && Ng != 384)) return false;
    if (a.ln_groups > 1 && !two_ln_groups && a.ln_groups != a.k_groups) return false;
    if (a.ktaps > 1) {      // conv form (gemm_row8c_bf16): LayerNorm-terminated convolutions, optionally with the scalar head; no PE
        if (!a.ln_g || a.pe || a.f16_terms) return false;
    } else if (a.dot_w || !(a.ln_g || a.pe) || a.k_groups > 1 || a.ln_groups > 1) return false;
    if (opts().row8 >= 0) return opts().row8 != 0;
    return ((a.regime_rows ? a.regime_rows : a.R) + 127) / 128 >= 128;
}

// Will launch_gemm run this LayerNorm-terminated k = 1 GEMM on gemm_row4_bf16, and does that kernel's mx form exist for it?  (run_stack asks before it
// decides the format of the A planes; launch_gemm makes the same test.)
bool ffn2_on_row4_mx(const fs2_handle*, const GemmArgs& a) {
    return use_row8(a) && opts().row4 != 0 && row4_epi(a) == 0 && a.Cpad % 128 == 0 && a.Xp != nullptr;
}

// In the 256-row regime the conv kernel runs two workgroups per CU: a launch of T y-tiles per N tile takes ceil(T nN / 512) rounds, and a nearly
// empty last round costs as much as a full one (c3 at 7.87 frames per phoneme: 143 x 8 tiles = 2.2 rounds -> 3).  The smallest tile height (a
// multiple of 32 rows, 160 .. 256) that keeps that number of rounds spreads the rows evenly instead (191 tiles of 192 rows: 3 full rounds of
// tiles that are a quarter shorter).  Results do not depend on the tile height.
inline int conv_bm_balanced(long rows, long nN) {
    const long ypr = std::max<long>(1, 2 * kCus / nN);
    const long rounds = std::max<long>(1, (rows + 256 * ypr - 1) / (256 * ypr));
    const long h = (rows + rounds * ypr - 1) / (rounds * ypr);
    return (int)std::min<long>(256, std::max<long>(160, (h + 31) / 32 * 32));
}
template <int NSPLIT, int ARITH>
hipError_t launch_pl_tall(hipStream_t s, const GemmArgs& a, int bm) {
    if (bm <= 160) return launch_pl_t<NSPLIT, 160, false, ARITH>(s, a);
    if (bm <= 192) return launch_pl_t<NSPLIT, 192, false, ARITH>(s, a);
    if (bm <= 224) return launch_pl_t<NSPLIT, 224, false, ARITH>(s, a);
    return launch_pl_t<NSPLIT, 256, false, ARITH>(s, a);
}

template <int NSPLIT>
hipError_t launch_pl(hipStream_t s, const GemmArgs& a) {
    const int force = opts().bm > 0 ? opts().bm : 0;
    const long rows = a.qk_hi ? a.Rvt : a.R;
    const long nN = (a.N + kB16BN - 1) / kB16BN;
    int bm;
    if (a.ktaps == 1) {
        bm = (force == 128) ? force : 64;     // measured (c3): 64-row tiles win for every k = 1 GEMM (3 workgroups/CU hide the DMA round trips)
        return bm == 128 ? launch_pl_t<NSPLIT, 128, true>(s, a) : launch_pl_t<NSPLIT, 64, true>(s, a);
    }
    bm = force ? force : (nN * ((rows + 255) / 256) >= 512 ? 256 : (nN * ((rows + 127) / 128) >= 400 ? 128 : 64));
    if (bm > 128) {
        if (!force && a.ksplit <= 1 && opts().bal) bm = conv_bm_balanced(opts().bal == 2 ? rows : rows_in_use(a, rows), nN);
        if constexpr (NSPLIT == 3) return launch_pl_tall<3, 0>(s, a, bm);
        else return launch_pl_t<NSPLIT, 256, false>(s, a);
    }
    return bm == 128 ? launch_pl_t<NSPLIT, 128, false>(s, a) : launch_pl_t<NSPLIT, 64, false>(s, a);
}

// fp16 + block-scaled-fp8 form of the conv (gemm_mx.h): the planes kernel on mx planes / the mx weight image
hipError_t launch_mx(hipStream_t s, const GemmArgs& a) {
    const int force = opts().bm > 0 ? opts().bm : 0;
    const long nN = (a.N + kB16BN - 1) / kB16BN;
    int bm = force ? force : (nN * ((a.R + 255) / 256) >= 512 ? 256 : (nN * ((a.R + 127) / 128) >= 400 ? 128 : 64));
    if (bm > 128) return launch_pl_tall<1, 2>(s, a, (force || !opts().bal) ? bm : conv_bm_balanced(opts().bal == 2 ? a.R : rows_in_use(a, a.R), nN));
    return bm == 128 ? launch_pl_t<1, 128, false, 2>(s, a) : launch_pl_t<1, 64, false, 2>(s, a);
}

// fp16 + block-scaled-fp4 form of the conv (gemm_planes.h ARITH = 3): 256-row tiles only -- it runs where the planes-only regime holds (>= 16 k rows: the
// tile-height rule above picks 256 there for every N >= 1024), + 512 bytes of LDS for the A tile's row-scale bytes
hipError_t launch_mx4(hipStream_t s, const GemmArgs& a) {
    static LdsAttr attr;
    // + the A tile's scale bytes (8 per row and cross unit) + two stages of weight block scales; two workgroups per CU must still fit (a first build reserved 32 bytes per
    // row = 78.3 KB per workgroup and ran 25 % slower: one workgroup per CU)
    const size_t lds = pl_lds_bytes<256, false>() + kMx4RowScaleLds + 2048;
    if (a.Cpad != 384) return hipErrorInvalidValue;      // (the LDS stride of the row scales is a compile-time constant: 8 bytes x 3 cross units; run_stack offers mx4 at D = 384 only)
    allow_lds(reinterpret_cast<const void*>(&gemm_pl_bf16<1, 256, false, 3>), pl_lds_bytes<256, false>() + kMx4RowScaleLds + 2048, attr);
    dim3 grid((a.N + kB16BN - 1) / kB16BN, (a.R + 255) / 256, 1);
    hipLaunchKernelGGL((gemm_pl_bf16<1, 256, false, 3>), grid, dim3(256), lds, s, a);
    return hipGetLastError();
}

// fp16-operand form of the conv kernel (FFN w_1 in the mixed modes): NSPLIT MFMAs per fragment pair
template <int NSPLIT>
hipError_t launch_pl_f16(hipStream_t s, const GemmArgs& a) {
    const int force = opts().bm > 0 ? opts().bm : 0;
    const long nN = (a.N + kB16BN - 1) / kB16BN;
    const int bm = force ? force : (nN * ((a.R + 255) / 256) >= 512 ? 256 : (nN * ((a.R + 127) / 128) >= 400 ? 128 : 64));
    if (bm == 256) return launch_pl_t<NSPLIT, 256, false, 1>(s, a);
    return bm == 128 ? launch_pl_t<NSPLIT, 128, false, 1>(s, a) : launch_pl_t<NSPLIT, 64, false, 1>(s, a);
}

// Picks the kernel.  fp32: row-complete tiles when the epilogue needs whole rows and N is small, 128x128 tiles (+ ln_rows) otherwise.
// bf16 / bf16x3 (activation planes in, gemm_planes.h): the row-complete LayerNorm-fused kernel for big k = 1 GEMMs that end in
// a row epilogue, else the BM x 128 tile kernel followed by ln_rows when a row epilogue is needed.
int launch_gemm(fs2_handle* h, hipStream_t s, const char* name, GemmArgs a, int precision = FS2_PREC_FP32) {
    if (a.ktaps - 1 > kMaxHalo) return fail(h, FS2_ERR_UNSUPPORTED, "%s: kernel size %d > %d", name, a.ktaps, kMaxHalo + 1);
    if (a.C % 4 != 0 || a.ldx % 4 != 0) return fail(h, FS2_ERR_UNSUPPORTED, "%s: channels %d / ld %d must be multiples of 4", name, a.C, a.ldx);
    if (h && !a.kpart && h->kp) { a.kpart = h->kp; a.kpart_cap = h->kp_cap; a.ksplit = 3; }
    if (h && !a.regime_rows) a.regime_rows = h->cur_regime;
    const int max_extra_splits = a.ksplit;      // on entry: how many partial buffers the caller allows; from here on a.ksplit = splits in use
    a.ksplit = 1;
    const bool need_rows = a.ln_g || a.dot_w || a.pe;
    const double flops = 2.0 * a.R * (double)a.N * a.C * a.ktaps;
    const double bytes = 4.0 * ((double)a.R * a.C + (double)a.N * a.C * a.ktaps + (double)a.R * a.N);
    hipError_t e;
    if (precision != FS2_PREC_FP32) {
        if (!a.Wb) return fail(h, FS2_ERR_STATE, "%s: no bf16 weight image", name);
        if (a.C % 8 != 0 || a.N % 4 != 0 || (need_rows && a.N > 1024)) return fail(h, FS2_ERR_UNSUPPORTED, "%s: bf16 path needs C %% 8 == 0, N %% 4 == 0 (N <= 1024 with a row epilogue)", name);
        GemmArgs t = a;
        t.W = reinterpret_cast<const float*>(a.Wb);
        // the A operand must exist as split-bf16 planes (Xp) or be convertible into xp_scratch (gemm_planes.h)
        if (!a.Xp && !a.xp_scratch) return fail(h, FS2_ERR_STATE, "%s: no activation planes and no scratch to build them", name);
        if (a.ldy % 4 != 0 || (a.resid && a.ldr % 4 != 0)) return fail(h, FS2_ERR_UNSUPPORTED, "%s: bf16 path needs row strides that are multiples of 4", name);
        const bool row8 = use_row8(a);
        if ((a.residp || (need_rows && !a.Y && a.Yp && !a.dot_w && a.ktaps == 1 && !a.scratch)) && !(row8 && precision == FS2_PREC_BF16X3 && opts().row4 != 0 && row4_epi(a) >= 0))
            return fail(h, FS2_ERR_STATE, "%s: a planes-only launch (residual as planes / no fp32 rows) exists on gemm_row4_bf16 only and this one would not run there", name);
        if (a.yp_col_off && !(row8 && a.ktaps > 1)) return fail(h, FS2_ERR_UNSUPPORTED, "%s: a plane column offset exists in the row-complete conv kernel only", name);
        const bool y_needed = (need_rows && !row8) || (!a.Yp && !a.qk_hi && !(row8 && a.dot_w));      // (row-complete + scalar head: nothing but dot_out leaves)
        if (!t.Y && y_needed) { t.Y = a.scratch; t.ldy = a.N; }
        if (!t.Y && y_needed) return fail(h, FS2_ERR_ARG, "%s: no output or scratch buffer", name);
        if (t.qk_hi && (a.ktaps != 1 || a.att_D % kB16BN != 0 || a.N != 3 * a.att_D)) return fail(h, FS2_ERR_UNSUPPORTED, "%s: fused QKV split needs D %% 128 == 0", name);
        // split-K: on a grid that leaves most CUs idle the kernel is a serial chain of k-steps (one utterance: 108 steps of the FFN
        // conv on 88 workgroups); 2-4 workgroups share the chunks and ln_rows adds their partial sums in a fixed order
        // (deterministic, unlike atomics) and applies the epilogue.  The choice depends on regime_rows (the same number in the host-
        // and the device-driven layout), never on the capacity.
        t.ksplit = 1;
        size_t y_slab = 0;       // 1: Y lives in the first slab of kpart (no fp32 output buffer of the caller's)
        const long rr = a.regime_rows ? a.regime_rows : a.R;
        if (!row8 && !a.qk_hi && a.kpart && !opts().nosplitk && a.N <= 1024 && rr <= kSplitRegime && a.R <= kSplitRows && a.mx != 2) {      // (the mx4 conv walks 9 units per row, not Cpad / 32: it exists unsplit -- its regime never splits unless the row kernels are forced on a small batch)
            const int nchunks = a.Cpad / 32;
            const long wgs = (long)((a.N + kB16BN - 1) / kB16BN) * ((rr + 63) / 64);
            const bool own_y = t.Y != nullptr;
            const size_t ld = own_y ? (size_t)t.ldy : (size_t)a.N;
            for (int cand = 4; cand >= 2; --cand)
                if (nchunks % cand == 0 && (nchunks / cand) * a.ktaps >= 4 && wgs * cand <= 1024 && cand - 1 <= max_extra_splits &&
                    (size_t)(cand - (own_y ? 1 : 0)) * a.R * ld <= a.kpart_cap) { t.ksplit = cand; break; }
            if (t.ksplit > 1 && !own_y) y_slab = 1;
        }
        const bool rows_pass = (need_rows && !row8) || t.ksplit > 1;
        if (y_slab) { t.Y = a.kpart; t.ldy = a.N; }
        if (rows_pass) { t.act_post = 0; t.Yp = nullptr; }      // the row kernel applies the epilogue and writes the planes
        if (t.ksplit > 1) {
            t.relu_pre = 0;                                      // partial sums: ReLU only after they are added (ln_rows)
            t.kpart = a.kpart + y_slab * (size_t)a.R * a.N;
            t.kpart_stride = (size_t)a.R * t.ldy;
        }
        const bool mx_row4 = a.mx && a.ktaps == 1 && precision == FS2_PREC_BF16X3 && ffn2_on_row4_mx(h, a);      // FFN2 + LN2 in the mx arithmetic (gemm_row4.h)
        if ((a.f16_terms || a.mx) && !mx_row4 && (a.ktaps == 1 || need_rows || a.qk_hi)) return fail(h, FS2_ERR_UNSUPPORTED, "%s: the fp16 arithmetic exists for plain convolutions only", name);
        if (a.mx && !mx_row4 && (a.ktaps < 3 || a.C % 128 != 0 || a.N % 128 != 0)) return fail(h, FS2_ERR_UNSUPPORTED, "%s: the mx arithmetic needs a convolution with C %% 128 == 0 and N %% 128 == 0", name);
        if (a.mx == 2 && (!a.x_rowscale || !a.w_rowscale || !a.Xp || t.ksplit > 1)) return fail(h, FS2_ERR_STATE, "%s: the mx4 arithmetic needs mx4 planes with their row scales and the weight image's channel scales", name);

        if (!a.Xp) {
            char nm[112];
            snprintf(nm, sizeof nm, "%s.planes", name);
            Scope sc(h, s, nm, 0.0, 8.0 * a.R * a.Cpad);
            const int64_t n = (int64_t)a.R * (a.Cpad / 4);
            hipLaunchKernelGGL(to_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.X, a.ldx, a.C, a.R, a.Cpad / 32, a.xp_scratch, a.mx ? 2 : (a.f16_terms ? 1 : 0), a.yp_scale);
            t.Xp = a.xp_scratch;
        }
        {
            Scope sc(h, s, name, flops, bytes);
            if (mx_row4) {
                t.W = reinterpret_cast<const float*>(a.Wb);
                e = launch_row4(s, t, 0);
            } else if (a.mx == 2) {
                e = launch_mx4(s, t);
            } else if (a.mx) {
                e = launch_mx(s, t);
            } else if (a.f16_terms) {
                e = a.f16_terms == 3 ? launch_pl_f16<3>(s, t) : (a.f16_terms == 2 ? launch_pl_f16<2>(s, t) : launch_pl_f16<1>(s, t));
            } else if (use_qkv8(t) && precision == FS2_PREC_BF16X3 && opts().qkv4 != 0 && opts().row4 != 0 && a.att_D == 384 && a.Cpad % 64 == 0) {
                e = launch_qkv4(s, t);
            } else if (use_qkv8(t)) {
                if (a.att_D == 384) e = (precision == FS2_PREC_BF16X3) ? launch_qkv8<3, 3>(s, t) : launch_qkv8<1, 3>(s, t);
                else e = (precision == FS2_PREC_BF16X3) ? launch_qkv8<3, 2>(s, t) : launch_qkv8<1, 2>(s, t);
            } else if (row8 && a.ktaps > 1 && a.ln_groups == 2 && a.k_groups <= 1) {      // two stacked layers over one input: N = 512, LayerNorm per N-wave
                e = (precision == FS2_PREC_BF16X3) ? launch_row8c_t<3, 4, 2, 2>(s, t) : launch_row8c_t<1, 4, 2, 2>(s, t);
            } else if (row8 && a.ktaps > 1) {
                if (a.k_groups > 1) { t.N = a.N / a.k_groups; t.ln_groups = 0; }      // grouped conv: a workgroup row per group (grid.y), N = one group's outputs
                if (t.N == 384) e = (precision == FS2_PREC_BF16X3) ? launch_row8c<3, 3>(s, t) : launch_row8c<1, 3>(s, t);
                else e = (precision == FS2_PREC_BF16X3) ? launch_row8c<3, 2>(s, t) : launch_row8c<1, 2>(s, t);
            } else if (row8 && precision == FS2_PREC_BF16X3 && opts().row4 != 0 && row4_epi(t) >= 0) {
                e = launch_row4(s, t, row4_epi(t));
            } else if (row8) {
                if (a.N == 384) e = (precision == FS2_PREC_BF16X3) ? launch_row8<3, 3>(s, t) : launch_row8<1, 3>(s, t);
                else e = (precision == FS2_PREC_BF16X3) ? launch_row8<3, 2>(s, t) : launch_row8<1, 2>(s, t);
            } else e = (precision == FS2_PREC_BF16X3) ? launch_pl<3>(s, t) : launch_pl<1>(s, t);
        }
        if (e == hipSuccess && rows_pass) {
            char nm[112];
            snprintf(nm, sizeof nm, "%s.rows", name);
            Scope sc(h, s, nm, 0.0, 8.0 * a.R * a.N);
            GemmArgs r = a;
            r.Y = t.Y; r.ldy = t.ldy; r.ksplit = t.ksplit; r.kpart = t.kpart; r.kpart_stride = t.kpart_stride;
            hipLaunchKernelGGL(ln_rows, dim3(((size_t)a.R * (a.ln_groups > 1 ? a.ln_groups : 1) + 3) / 4), dim3(256), 0, s, r);
            e = hipGetLastError();
        }
    } else if (need_rows && a.N >= 128 && a.N <= 1024 && a.N % 4 == 0 && (a.Y || a.scratch) && !opts().f32_rows) {
        // fp32, LayerNorm-terminated: 128x128 MFMA tiles + the HBM-bound row kernel (2x faster than the row-complete
        // GEMM, whose 16-rows-per-wave shape re-stages the whole weight matrix for every 64 rows)
        GemmArgs t = a;
        if (!t.Y) { t.Y = a.scratch; t.ldy = a.N; }
        t.act_post = 0;
        {
            Scope sc(h, s, name, flops, bytes);
            e = launch_tile(s, t);
        }
        if (e == hipSuccess) {
            char nm[112];
            snprintf(nm, sizeof nm, "%s.rows", name);
            Scope sc(h, s, nm, 0.0, 8.0 * a.R * a.N);
            GemmArgs r = a;
            r.Y = t.Y; r.ldy = t.ldy;
            hipLaunchKernelGGL(ln_rows, dim3((a.R + 3) / 4), dim3(256), 0, s, r);
            e = hipGetLastError();
        }
    } else {
        Scope sc(h, s, name, flops, bytes);
        if (need_rows || (a.N < 128 && rows_supported(a.N))) {
            if (!rows_supported(a.N)) return fail(h, FS2_ERR_UNSUPPORTED, "%s: row-epilogue GEMM needs N in {80,256,384}, got %d", name, a.N);
            if (a.N == 80) e = launch_rows<5>(s, a);
            else if (a.N == 256) e = launch_rows<16>(s, a);
            else e = launch_rows<24>(s, a);
        } else {
            e = launch_tile(s, a);
        }
    }
    if (e != hipSuccess) return fail(h, FS2_ERR_HIP, "%s launch: %s", name, hipGetErrorString(e));
    return FS2_OK;
}

GemmArgs gemm_args(const Gemm& g, const float* X, int ldx, int R, const int* row_pos, float* Y, int ldy) {
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.X = X; a.ldx = ldx; a.C = g.C; a.W = g.w; a.Cpad = g.Cpad; a.ktaps = g.ktaps; a.N = g.N; a.R = R;
    a.row_pos = row_pos; a.bias = g.bias; a.Y = Y; a.ldy = ldy; a.x_scale = 1.f; a.ln_eps = 1e-5f; a.Wb = g.wb;
    return a;
}

int launch_attention(fs2_handle* h, hipStream_t s, const char* name, const float* qkv, float* ctx, int D, int heads,
                     const DevLayout& dl, int nwork, int mask_q, double flops, int dk_true = 0) {
    const int dk = D / heads;      // (D = heads x padded head dim; dk_true: the model's head dim, for the softmax scale)
    if (!dk_true) dk_true = dk;
    AttnArgs a;
    a.qkv = qkv; a.ld = 3 * D; a.ctx = ctx; a.ldc = D; a.start = dl.start; a.len = dl.len; a.klen = dl.klen;
    a.work = dl.work; a.nwork = dl.dims ? dl.dims + 1 : nullptr; a.nitems = nwork; a.D = D; a.mask_q = mask_q; a.scale = 1.0f / sqrtf((float)dk_true);
    if (nwork == 0) return FS2_OK;
    Scope sc(h, s, name, flops, 0.0);
    dim3 grid(att_grid64(nwork), heads);      // two 64-query workgroups per 128-query work item
    if (dk == 128) {
        static LdsAttr attr;
        allow_lds(reinterpret_cast<const void*>(&attn_f32<128>), attn_lds_bytes<128>(), attr);
        hipLaunchKernelGGL(attn_f32<128>, grid, dim3(256), attn_lds_bytes<128>(), s, a);
    } else if (dk == 192) {
        static LdsAttr attr;
        allow_lds(reinterpret_cast<const void*>(&attn_f32<192>), attn_lds_bytes<192>(), attr);
        hipLaunchKernelGGL(attn_f32<192>, grid, dim3(256), attn_lds_bytes<192>(), s, a);
    } else if (dk == 64) {
        static LdsAttr attr;
        allow_lds(reinterpret_cast<const void*>(&attn_f32<64>), attn_lds_bytes<64>(), attr);
        hipLaunchKernelGGL(attn_f32<64>, grid, dim3(256), attn_lds_bytes<64>(), s, a);
    } else if (dk == 256) {
        static LdsAttr attr;
        allow_lds(reinterpret_cast<const void*>(&attn_f32<256>), attn_lds_bytes<256>(), attr);
        hipLaunchKernelGGL(attn_f32<256>, grid, dim3(256), attn_lds_bytes<256>(), s, a);
    } else {
        return fail(h, FS2_ERR_UNSUPPORTED, "attention head dim %d not in {64,128,192,256} (the model path pads other head dims)", dk);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, FS2_ERR_HIP, "%s launch: %s", name, hipGetErrorString(e));
    return FS2_OK;
}

template <int DK>
hipError_t launch_attn_w32_t(hipStream_t s, dim3 grid, const AttnB16Args& a) {
    static LdsAttr attr;
    allow_lds(reinterpret_cast<const void*>(&attn_w32<DK>), attn_w32_lds_bytes<DK>(), attr);
    hipLaunchKernelGGL((attn_w32<DK>), grid, dim3(256), attn_w32_lds_bytes<DK>(), s, a);
    return hipGetLastError();
}

// attn_w32 (32 queries per wave, one wave per SIMD, 128-query workgroups) serves the split-bf16 launches whose grid fills the chip:
// below kW32MinBlocks 128-query blocks per head a 128-query workgroup per CU leaves CUs idle and doubles the serial chain of an
// utterance's key tiles per wave; the 64-query kernel keeps those (token-level launches, single utterances).  The choice is a
// function of the regime row count (fs2_decode: derived from the phoneme count), like every other kernel-variant choice.
constexpr int kW32MinBlocks = 64;
bool use_attn_w32(int precision, int dk, long rows, unsigned long long qk_lo_bytes, unsigned long long vt_lo_bytes) {
    if (precision != FS2_PREC_BF16X3 || (dk != 128 && dk != 192)) return false;
    if (qk_lo_bytes >= (1ull << 31) || vt_lo_bytes >= (1ull << 31)) return false;      // the lo planes must sit within 2 GB behind the hi planes (32-bit DMA offsets)
    if (opts().w32 >= 0) return opts().w32 != 0;
    return (rows + kAttBlk - 1) / kAttBlk >= kW32MinBlocks;
}

template <int DK, int NSPLIT>
hipError_t launch_attn_b16_t(hipStream_t s, dim3 grid, const AttnB16Args& a) {
    static LdsAttr attr;
    allow_lds(reinterpret_cast<const void*>(&attn_bf16<DK, NSPLIT>), attn_b16_lds_bytes<DK>(), attr);
    hipLaunchKernelGGL((attn_bf16<DK, NSPLIT>), grid, dim3(256), attn_b16_lds_bytes<DK>(), s, a);
    return hipGetLastError();
}

// qkv fp32 [R,3D] -> split planes -> attention.  planes: qk_hi/lo [Rvt][2D], vt_hi/lo [D][Rvt] (Rvt % 32 == 0)
int launch_attention_b16(fs2_handle* h, hipStream_t s, const char* name, const float* qkv, float* ctx, int D, int heads, int R, int Rvt,
                         const DevLayout& dl, int nwork, int mask_q, double flops, int precision, __bf16* qkh, __bf16* qkl, __bf16* vth,
                         __bf16* vtl, void* ctxp = nullptr, int dk_true = 0) {
    const int dk = D / heads;
    if (!dk_true) dk_true = dk;
    if (nwork == 0) return FS2_OK;
    if (qkv != nullptr) {
        char nm[112];
        snprintf(nm, sizeof nm, "%s.split", name);
        Scope sc(h, s, nm, 0.0, 4.0 * R * 3.0 * D * 2);
        static LdsAttr attr;
        allow_lds(reinterpret_cast<const void*>(&qkv_split), 32 * (1024 + 1) * 4, attr);
        QkvSplitArgs q;
        q.qkv = qkv; q.R = R; q.Rvt = Rvt; q.D = D; q.dk = dk; q.scale = 1.4426950408889634f / sqrtf((float)dk_true);   // log2(e)/sqrt(d_k): softmax in base 2
        q.qk_hi = qkh; q.qk_lo = qkl; q.vt_hi = vth; q.vt_lo = vtl;
        if (D > 1024) return fail(h, FS2_ERR_UNSUPPORTED, "attention dim %d > 1024", D);
        hipLaunchKernelGGL(qkv_split, dim3(Rvt / 32), dim3(256), (size_t)32 * (D + 1) * 4, s, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return fail(h, FS2_ERR_HIP, "%s launch: %s", nm, hipGetErrorString(e));
    }
    AttnB16Args a;
    a.qk_hi = qkh; a.qk_lo = qkl; a.ldqk = 2 * D; a.vt_hi = vth; a.vt_lo = vtl; a.Rvt = Rvt; a.ctx = ctx; a.ldc = D;
    a.ctxp = ctxp; a.ctxp_chunks = D / 32;
    a.start = dl.start; a.len = dl.len; a.klen = dl.klen; a.work = dl.work; a.nwork = dl.dims ? dl.dims + 1 : nullptr; a.nitems = nwork; a.D = D; a.mask_q = mask_q;
    const ptrdiff_t qk_d = reinterpret_cast<const char*>(qkl) - reinterpret_cast<const char*>(qkh), vt_d = reinterpret_cast<const char*>(vtl) - reinterpret_cast<const char*>(vth);
    a.qk_lo_bytes = (unsigned)qk_d; a.vt_lo_bytes = (unsigned)vt_d;
    a.slow_count = h ? h->counters : nullptr; a.slow_count2 = (h && h->cur_status) ? h->cur_status + 5 : nullptr;
    Scope sc(h, s, name, flops, 0.0);
    hipError_t e;
    // (the regime row count of the call in progress when there is one: derived from the phoneme count, it is the same number in the host-
    //  and the device-driven layout of a batch, whose R -- rows in use vs row capacity -- differ)
    const long regime = (h && h->cur_regime > 0) ? h->cur_regime : R;
    if (qk_d > 0 && vt_d > 0 && use_attn_w32(precision, dk, regime, (unsigned long long)qk_d, (unsigned long long)vt_d)) {
        dim3 grid(nwork, heads);
        e = dk == 128 ? launch_attn_w32_t<128>(s, grid, a) : launch_attn_w32_t<192>(s, grid, a);
        if (e != hipSuccess) return fail(h, FS2_ERR_HIP, "%s launch: %s", name, hipGetErrorString(e));
        return FS2_OK;
    }
    dim3 grid(att_grid64(nwork), heads);      // two 64-query workgroups per 128-query work item
    const bool x3 = precision == FS2_PREC_BF16X3;
    if (dk == 128) e = x3 ? launch_attn_b16_t<128, 3>(s, grid, a) : launch_attn_b16_t<128, 1>(s, grid, a);
    else if (dk == 192) e = x3 ? launch_attn_b16_t<192, 3>(s, grid, a) : launch_attn_b16_t<192, 1>(s, grid, a);
    else if (dk == 64) e = x3 ? launch_attn_b16_t<64, 3>(s, grid, a) : launch_attn_b16_t<64, 1>(s, grid, a);
    else if (dk == 256) e = x3 ? launch_attn_b16_t<256, 3>(s, grid, a) : launch_attn_b16_t<256, 1>(s, grid, a);
    else return fail(h, FS2_ERR_UNSUPPORTED, "attention head dim %d not in {64,128,192,256} (the model path pads other head dims)", dk);
    if (e != hipSuccess) return fail(h, FS2_ERR_HIP, "%s launch: %s", name, hipGetErrorString(e));
    return FS2_OK;
}

// ------------------------------------------------------------------ layouts
// Attention work list = (utterance, 128-query block) items (kAttBlk: one workgroup of attn_w32, two of the 64-query kernels),
// dispatched in list order.  Two goals:
//  * XCD locality: workgroup i of a launch runs on XCD i % 8 (round-robin dispatch) and every XCD has its own L2.  All query
//    blocks of one utterance stream the same K / V rows, so they are given to ONE XCD: the list is eight interleaved queues
//    (entry 8 i + j = i-th item of queue j).  With the blocks of an utterance spread over all XCDs the K / V planes were
//    fetched from HBM once per XCD: 830 MB per decoder-layer launch at c3 (PMC FETCH_SIZE), the whole 128 us of it.
//  * balance: utterances go to the queues longest first, each to the currently shortest queue (LPT); queues are padded to
//    the same length with (-1, 0) entries that exit at once.
void build_work_list(const std::vector<int>& len, const std::vector<int>& klen, std::vector<int2>& work) {
    const int B = (int)len.size();
    std::vector<int> order(B);
    for (int b = 0; b < B; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return klen[x] > klen[y]; });
    std::vector<int2> q[kXcds];
    for (int b : order) {
        int j = 0;
        for (int t = 1; t < kXcds; ++t) if (q[t].size() < q[j].size()) j = t;
        for (int i = 0; i * kAttBlk < len[b]; ++i) q[j].push_back(make_int2(b, i));
    }
    size_t depth = 0;
    for (int j = 0; j < kXcds; ++j) depth = std::max(depth, q[j].size());
    work.assign(depth * kXcds, make_int2(-1, 0));
    for (int j = 0; j < kXcds; ++j)
        for (size_t i = 0; i < q[j].size(); ++i) work[i * kXcds + j] = q[j][i];
}

void build_layout(HostLayout& L, int B, const std::vector<int>& len, const std::vector<int>& klen, const std::vector<int>& vlen, int gap) {
    L.B = B; L.len = len; L.klen = klen; L.vlen = vlen;
    L.start.resize(B);
    int row = gap;
    for (int b = 0; b < B; ++b) {
        row = round_up(row, kAttAlign);     // aligned starts: 16-byte aligned V^T key tiles (attn_bf16.h)
        L.start[b] = row;
        row += len[b] + gap;
    }
    row += kTailRows;        // zero rows behind the last utterance (attn_bf16.h: V^T vectors that straddle the last key)
    L.R = row;
    L.Rpad = round_up(row, 128);
    build_work_list(len, klen, L.work);
}

size_t layout_dev_ints(const HostLayout& L) { return (size_t)7 * L.B + 2 * (size_t)L.Rpad + 2 * (size_t)L.nwork() + 64 + 16; }

// uploads start/len/klen/vlen/work, then derives row_pos/row_seq on the device
int upload_layout(fs2_handle* h, hipStream_t s, const HostLayout& L, int* dev, DevLayout& D) {
    std::vector<int> host;
    host.reserve(4 * L.B + 2 * L.work.size());
    host.insert(host.end(), L.start.begin(), L.start.end());
    host.insert(host.end(), L.len.begin(), L.len.end());
    host.insert(host.end(), L.klen.begin(), L.klen.end());
    host.insert(host.end(), L.vlen.begin(), L.vlen.end());
    for (auto& w : L.work) { host.push_back(w.x); host.push_back(w.y); }
    const void* src = host.data();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusActive) {
        // Inside a HIP-graph capture the copy becomes a memcpy node that reads its source at every replay: give it pinned host
        // memory that lives as long as the handle (a pageable std::vector is neither capturable nor alive at replay time).
        void* pin = nullptr;
        if (!h) return fail(h, FS2_ERR_STATE, "layout upload inside a graph capture needs a model handle");
        HIP_TRY(h, hipHostMalloc(&pin, host.size() * sizeof(int), hipHostMallocDefault));
        memcpy(pin, host.data(), host.size() * sizeof(int));
        h->graph_pinned.push_back(pin);
        src = pin;
    }
    HIP_TRY(h, hipMemcpyAsync(dev, src, host.size() * sizeof(int), hipMemcpyHostToDevice, s));
    D.start = dev; D.len = dev + L.B; D.klen = dev + 2 * L.B; D.vlen = dev + 3 * L.B;
    D.work = reinterpret_cast<int2*>(dev + 4 * L.B);
    int* rest = dev + 4 * L.B + 2 * L.work.size();
    rest = reinterpret_cast<int*>(align_up(reinterpret_cast<size_t>(rest), 16));
    D.row_pos = rest; D.row_seq = rest + L.Rpad;
    hipLaunchKernelGGL(build_row_meta, dim3((L.Rpad + 255) / 256), dim3(256), 0, s, D.start, D.len, L.B, L.Rpad, D.row_pos, D.row_seq);
    HIP_TRY(h, hipGetLastError());
    return FS2_OK;
}

// device-driven variant: the layout arrays are produced by frame_layout_dev from the device frame counts
int device_layout(fs2_handle* h, hipStream_t s, const HostLayout& L, int* dev, DevLayout& D, const int* olens32, int compat, int masked,
                  int lmax_cap, int pe_rows, int* status) {
    D.start = dev; D.len = dev + L.B; D.klen = dev + 2 * L.B; D.vlen = dev + 3 * L.B;
    int* rank_tmp = dev + 4 * L.B;
    int* woff_tmp = dev + 5 * L.B;
    D.pcum = dev + 6 * L.B;
    int* rest = reinterpret_cast<int*>(align_up(reinterpret_cast<size_t>(dev + 7 * L.B), 16));
    D.dims = rest; rest += 16;
    D.work = reinterpret_cast<int2*>(rest);
    rest += 2 * (size_t)L.nwork();
    rest = reinterpret_cast<int*>(align_up(reinterpret_cast<size_t>(rest), 16));
    D.row_pos = rest; D.row_seq = rest + L.Rpad;
    hipLaunchKernelGGL(frame_layout_dev, dim3(1), dim3(1024), 0, s, olens32, L.B, compat, masked, L.R, L.nwork(), lmax_cap, pe_rows,
                       D.start, D.len, D.klen, D.vlen, rank_tmp, woff_tmp, D.pcum, D.work, D.dims, status, h ? h->gap : kGap);
    hipLaunchKernelGGL(build_row_meta, dim3((L.Rpad + 255) / 256), dim3(256), 0, s, D.start, D.len, L.B, L.Rpad, D.row_pos, D.row_seq);
    HIP_TRY(h, hipGetLastError());
    return FS2_OK;
}

// ------------------------------------------------------------------ FFT block stack
// x0p / x1p: split-bf16 planes of x0 / x1 (gemm_planes.h); xps: planes scratch for activations produced without planes.
// In the bf16 modes the attention context and the FFN hidden layer exist ONLY as planes, in the ctx / hid storage.
struct StackBufs { float *x0, *x1, *qkv, *ctx, *hid; __bf16 *qkh, *qkl, *vth, *vtl; void *x0p, *x1p, *xps; unsigned char* xs4 = nullptr; };      // xs4: row-scale bytes of x1p in the mx4 format

int run_stack_general(fs2_handle* h, hipStream_t s, const char* tag, const Stack& st, int D, int heads, int R, const HostLayout& L, const DevLayout& dl,
                      int mask_q, const StackBufs& b, int prec, bool x0p_ready, int regime_rows);

// x0 holds the input; returns the buffer holding the output (x0 again).
// po: the planes-only residual stream (gemm_row4.h: RES; fs2_decode decides with planes_only_regime): the LayerNorm-fused launches write no fp32
// rows and take their residual from the planes of the producing launch -- x0 / x1 are then never written: the output is b.x0p.
int run_stack(fs2_handle* h, hipStream_t s, const char* tag, const Stack& st, int D, int heads, int R,
              const HostLayout& L, const DevLayout& dl, int mask_q, const StackBufs& b, int prec, bool x0p_ready = false, bool allow_splitk = false,
              int regime_rows = 0, int ffn_terms = 0, bool po = false) {
    if (st.pre_ln || st.concat) return run_stack_general(h, s, tag, st, D, heads, R, L, dl, mask_q, b, prec, x0p_ready, regime_rows);
    const int Dp = st.Dp;      // attention width (= D unless the head dim is padded)
    char nm[96];
    double att_flops = 0;
    for (size_t i = 0; i < L.klen.size(); ++i) att_flops += 4.0 * D * (double)L.klen[i] * std::min(L.len[i], mask_q ? L.klen[i] : L.len[i]);
    // bf16 modes: activations travel between the GEMMs as split-bf16 planes (no conversion work inside the MFMA loops)
    const bool pl = prec != FS2_PREC_FP32;
    void* ctxp = pl ? (void*)b.ctx : nullptr;
    void* hidp = pl ? (void*)b.hid : nullptr;
    if (pl && !x0p_ready) {
        snprintf(nm, sizeof nm, "%s.in.planes", tag);
        Scope sc(h, s, nm, 0.0, 8.0 * R * D);
        const int64_t n = (int64_t)R * (D / 4);
        hipLaunchKernelGGL(to_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, b.x0, D, D, R, D / 32, b.x0p);
    }
    for (size_t li = 0; li < st.layers.size(); ++li) {
        const Layer& ly = st.layers[li];
        int rc;
        snprintf(nm, sizeof nm, "%s.qkv", tag);
        GemmArgs a = gemm_args(ly.qkv, b.x0, D, R, dl.row_pos, b.qkv, 3 * Dp);
        a.Rp = dl.dims; a.regime_rows = regime_rows;
        if (pl) a.Xp = b.x0p;
        const bool fused_split = prec != FS2_PREC_FP32 && Dp % kB16BN == 0;
        if (fused_split) {   // bf16 attention operands straight from the GEMM epilogue (no fp32 QKV round trip)
            a.Y = nullptr; a.qk_hi = b.qkh; a.qk_lo = b.qkl; a.vt_hi = b.vth; a.vt_lo = b.vtl; a.att_D = Dp; a.Rvt = L.Rpad;
            a.q_scale = 1.4426950408889634f / sqrtf((float)st.dk);
        }
        if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
        snprintf(nm, sizeof nm, "%s.attn", tag);
        if (prec == FS2_PREC_FP32) rc = launch_attention(h, s, nm, b.qkv, b.ctx, Dp, heads, dl, L.nwork(), mask_q, att_flops, st.dk);
        else rc = launch_attention_b16(h, s, nm, fused_split ? nullptr : b.qkv, pl ? nullptr : b.ctx, Dp, heads, R, L.Rpad, dl, L.nwork(), mask_q, att_flops, prec, b.qkh, b.qkl, b.vth, b.vtl, ctxp, st.dk);
        if (rc) return rc;
        snprintf(nm, sizeof nm, "%s.out_ln", tag);
        a = gemm_args(ly.out, b.ctx, Dp, R, dl.row_pos, po ? nullptr : b.x1, D);
        a.Rp = dl.dims; a.regime_rows = regime_rows;
        a.ln_g = ly.ln1g; a.ln_b = ly.ln1b; a.ln_eps = 1e-5f;
        if (po) { a.residp = b.x0p; a.residp_chunks = D / 32; }      // (x0p: split-bf16 planes, from the input layer or the previous block's FFN2 + LN2)
        else { a.resid = b.x0; a.ldr = D; }
        const bool mx4l = po && ffn_terms == kFfnMx4 && ly.w1.wm4 && b.xs4 && D == 384;                       // this layer's FFN conv in the mx4 arithmetic (fp4 cross terms; planes-only regime)?
        const bool mxl = pl && (ffn_terms == kFfnMx || ffn_terms == kFfnMx4) && ly.w1.wm;         // ... in the mx arithmetic (also the fallback of mix_mx4 wherever mx4 does not apply)?
        const int f16t = (pl && ffn_terms && ffn_terms != kFfnMx && ffn_terms != kFfnMx4 && ly.w1.ktaps > 1 && ly.w1.wf) ? ffn_terms : 0;      // ... or on fp16 operands?
        if (pl) { a.Xp = ctxp; a.Yp = b.x1p; a.yp_chunks = D / 32; a.yp_f16 = mx4l ? 3 : (mxl ? 2 : (f16t ? 1 : 0)); a.yp_scale = mxl ? exp2f((float)ly.ka) : 1.f; }       // x1p feeds only that conv
        if (mx4l) a.yp_rowscale = b.xs4;
        if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
        // the second FFN GEMM: built first, because WHERE it will run decides the format the hidden layer leaves FFN1 in (mx planes for gemm_row4_bf16's
        // mx form, split-bf16 planes for everything else)
        GemmArgs a2 = gemm_args(ly.w2, b.hid, ly.w1.N, R, dl.row_pos, po ? nullptr : b.x0, D);
        a2.Rp = dl.dims; a2.regime_rows = regime_rows;
        a2.ln_g = ly.ln2g; a2.ln_b = ly.ln2b; a2.ln_eps = 1e-5f;
        if (po) { a2.residp = b.x1p; a2.residp_chunks = D / 32; a2.residp_mx = mx4l ? 2 : (mxl ? 1 : 0); a2.residp_scale = mxl ? exp2f(-(float)(ly.ka + 11)) : 1.f; }      // (x1p: LN1's output in the format the FFN conv wants)
        else { a2.resid = b.x1; a2.ldr = D; }
        if (pl) { a2.Xp = hidp; a2.Yp = b.x0p; a2.yp_chunks = D / 32; }
        const bool mx2 = mxl && ly.w2.wm && opts().ffn2_mx && prec == FS2_PREC_BF16X3 && ffn2_on_row4_mx(h, a2);
        if (mx2) { a2.mx = 1; a2.Wb = ly.w2.wm; a2.mx_scale = scale_byte4(127 - ly.kh - 11); a2.mx_scale_b = scale_byte4(127 - ly.w2.kw); }
        snprintf(nm, sizeof nm, "%s.ffn1", tag);
        a = gemm_args(ly.w1, b.x1, D, R, dl.row_pos, pl ? nullptr : b.hid, ly.w1.N);
        a.Rp = dl.dims;
        a.act_post = 1;
        if (pl) { a.Xp = b.x1p; a.Yp = hidp; a.yp_chunks = round_up(ly.w1.N, 32) / 32; }
        if (mx2) { a.yp_f16 = 2; a.yp_scale = exp2f((float)ly.kh); }
        if (f16t) { a.f16_terms = f16t; a.Wb = ly.w1.wf; }
        if (mx4l) { a.mx = 2; a.Wb = ly.w1.wm4; a.x_rowscale = b.xs4; a.w_rowscale = ly.w1.ws4; }
        else if (mxl) { a.mx = 1; a.Wb = ly.w1.wm; a.mx_scale = scale_byte4(127 - ly.ka - 11); a.mx_scale_b = scale_byte4(127 - ly.w1.kw); }
        if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
        snprintf(nm, sizeof nm, "%s.ffn2_ln", tag);
        if ((rc = launch_gemm(h, s, nm, a2, prec))) return rc;
    }
    return FS2_OK;
}

// standalone LayerNorm over rows (out of place): src [R, ld] -> Y [R, N] fp32 and / or planes
int launch_ln(fs2_handle* h, hipStream_t s, const char* name, const float* src, int ld, int R, int N, const int* row_pos, const float* g,
              const float* bta, float eps, float* Y, void* Yp) {
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.N = N; a.R = R; a.row_pos = row_pos; a.Y = Y; a.ldy = N; a.Ysrc = src; a.ldsrc = ld; a.ln_g = g; a.ln_b = bta; a.ln_eps = eps;
    a.x_scale = 1.f; a.ksplit = 1; a.Yp = Yp; a.yp_chunks = round_up(N, 32) / 32;
    if (N > 1024 || N % 4) return fail(h, FS2_ERR_UNSUPPORTED, "%s: LayerNorm width %d", name, N);
    Scope sc(h, s, name, 0.0, 8.0 * R * N);
    hipLaunchKernelGGL(ln_rows, dim3((R + 3) / 4), dim3(256), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(h, FS2_ERR_HIP, "%s launch: %s", name, hipGetErrorString(e));
    return FS2_OK;
}

// FFT-block stack for the non-default block variants (reference core/encoder.py:53-71,201-202): normalize_before (LayerNorm in front of
// the sub-layers, after_norm at the end) and / or concat_after (x + concat_linear(cat(x, self_attn(x)))).  Same kernels as run_stack,
// without the LayerNorm fusion where the norm no longer follows a GEMM; cat(x, a) . Wc^T is computed as x . Wc[:, :D]^T + a . Wc[:, D:]^T
// (two GEMMs chained through the residual input).  Scratch: xn1 -> hid / xps, attention output -> qkv / qkh, xn2 -> qkv / xps.
int run_stack_general(fs2_handle* h, hipStream_t s, const char* tag, const Stack& st, int D, int heads, int R, const HostLayout& L, const DevLayout& dl,
                      int mask_q, const StackBufs& b, int prec, bool x0p_ready, int regime_rows) {
    char nm[96];
    double att_flops = 0;
    for (size_t i = 0; i < L.klen.size(); ++i) att_flops += 4.0 * D * (double)L.klen[i] * std::min(L.len[i], mask_q ? L.klen[i] : L.len[i]);
    const bool pl = prec != FS2_PREC_FP32;
    const bool pre = st.pre_ln, cat = st.concat;
    const int Dp = st.Dp;
    void* ctxp = pl ? (void*)b.ctx : nullptr;
    void* hidp = pl ? (void*)b.hid : nullptr;
    int rc;
    if (pl && !pre && !x0p_ready) {
        const int64_t n = (int64_t)R * (D / 4);
        hipLaunchKernelGGL(to_planes, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, b.x0, D, D, R, D / 32, b.x0p, 0);
    }
    for (size_t li = 0; li < st.layers.size(); ++li) {
        const Layer& ly = st.layers[li];
        const float* qin = b.x0; const void* qinp = pl ? b.x0p : nullptr;
        if (pre) {
            snprintf(nm, sizeof nm, "%s.ln1", tag);
            if ((rc = launch_ln(h, s, nm, b.x0, D, R, D, dl.row_pos, ly.ln1g, ly.ln1b, 1e-5f, pl ? nullptr : b.hid, pl ? b.xps : nullptr))) return rc;
            qin = b.hid; qinp = pl ? b.xps : nullptr;
        }
        snprintf(nm, sizeof nm, "%s.qkv", tag);
        GemmArgs a = gemm_args(ly.qkv, qin, D, R, dl.row_pos, b.qkv, 3 * Dp);
        a.Rp = dl.dims; a.regime_rows = regime_rows; a.Xp = qinp;
        const bool fused_split = pl && Dp % kB16BN == 0;
        if (fused_split) {
            a.Y = nullptr; a.qk_hi = b.qkh; a.qk_lo = b.qkl; a.vt_hi = b.vth; a.vt_lo = b.vtl; a.att_D = Dp; a.Rvt = L.Rpad;
            a.q_scale = 1.4426950408889634f / sqrtf((float)st.dk);
        }
        if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
        snprintf(nm, sizeof nm, "%s.attn", tag);
        if (!pl) rc = launch_attention(h, s, nm, b.qkv, b.ctx, Dp, heads, dl, L.nwork(), mask_q, att_flops, st.dk);
        else rc = launch_attention_b16(h, s, nm, fused_split ? nullptr : b.qkv, nullptr, Dp, heads, R, L.Rpad, dl, L.nwork(), mask_q, att_flops, prec, b.qkh, b.qkl, b.vth, b.vtl, ctxp, st.dk);
        if (rc) return rc;
        if (cat) {
            // a = linear_out(context) (no residual), then x1 = x0 + xq . Wc_x^T + bc, then x1 += a . Wc_a^T (+ LayerNorm when post-LN)
            float* att = b.qkv; void* attp = pl ? (void*)b.qkh : nullptr;       // both free once the attention kernel has run
            snprintf(nm, sizeof nm, "%s.out", tag);
            a = gemm_args(ly.out, b.ctx, Dp, R, dl.row_pos, pl ? nullptr : att, D);
            a.Rp = dl.dims; a.regime_rows = regime_rows;
            if (pl) { a.Xp = ctxp; a.Yp = attp; a.yp_chunks = D / 32; }
            if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
            snprintf(nm, sizeof nm, "%s.cat_x", tag);
            a = gemm_args(ly.cat_x, qin, D, R, dl.row_pos, b.x1, D);
            a.Rp = dl.dims; a.regime_rows = regime_rows; a.Xp = qinp; a.resid = b.x0; a.ldr = D;
            if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
            snprintf(nm, sizeof nm, "%s.cat_a", tag);
            a = gemm_args(ly.cat_a, att, D, R, dl.row_pos, b.x1, D);
            a.Rp = dl.dims; a.regime_rows = regime_rows; a.Xp = attp; a.resid = b.x1; a.ldr = D;      // in place: every element is read, then written, by one lane
            if (!pre) { a.ln_g = ly.ln1g; a.ln_b = ly.ln1b; a.ln_eps = 1e-5f; if (pl) { a.Yp = b.x1p; a.yp_chunks = D / 32; } }
            if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
        } else {
            snprintf(nm, sizeof nm, "%s.out", tag);
            a = gemm_args(ly.out, b.ctx, Dp, R, dl.row_pos, b.x1, D);
            a.Rp = dl.dims; a.regime_rows = regime_rows; a.resid = b.x0; a.ldr = D;
            if (pl) a.Xp = ctxp;
            if (!pre) { a.ln_g = ly.ln1g; a.ln_b = ly.ln1b; a.ln_eps = 1e-5f; if (pl) { a.Yp = b.x1p; a.yp_chunks = D / 32; } }
            if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
        }
        const float* fin = b.x1; const void* finp = pl ? b.x1p : nullptr;
        if (pre) {
            snprintf(nm, sizeof nm, "%s.ln2", tag);
            if ((rc = launch_ln(h, s, nm, b.x1, D, R, D, dl.row_pos, ly.ln2g, ly.ln2b, 1e-5f, pl ? nullptr : b.qkv, pl ? b.xps : nullptr))) return rc;
            fin = b.qkv; finp = pl ? b.xps : nullptr;
        }
        snprintf(nm, sizeof nm, "%s.ffn1", tag);
        a = gemm_args(ly.w1, fin, D, R, dl.row_pos, pl ? nullptr : b.hid, ly.w1.N);
        a.Rp = dl.dims; a.regime_rows = regime_rows; a.act_post = 1; a.Xp = finp;
        if (pl) { a.Yp = hidp; a.yp_chunks = round_up(ly.w1.N, 32) / 32; }
        if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
        snprintf(nm, sizeof nm, "%s.ffn2", tag);
        a = gemm_args(ly.w2, b.hid, ly.w1.N, R, dl.row_pos, b.x0, D);
        a.Rp = dl.dims; a.regime_rows = regime_rows; a.resid = b.x1; a.ldr = D;
        if (pl) a.Xp = hidp;
        if (!pre) { a.ln_g = ly.ln2g; a.ln_b = ly.ln2b; a.ln_eps = 1e-5f; if (pl) { a.Yp = b.x0p; a.yp_chunks = D / 32; } }
        if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
    }
    if (pre) {      // Encoder.forward: xs = after_norm(xs); downstream consumers read x0 (fp32) and, in the bf16 modes, its planes
        snprintf(nm, sizeof nm, "%s.after_norm", tag);
        if ((rc = launch_ln(h, s, nm, b.x0, D, R, D, dl.row_pos, st.after_g, st.after_b, 1e-5f, b.x0, pl ? b.x0p : nullptr))) return rc;
    }
    return FS2_OK;
}

// conv stack + scalar head (reference variance_predictor.py:46-51 / duration_predictor.py:70-75);
// tmp0/tmp1: [R, chans] scratch; out_rows: [R]
int run_predictor(fs2_handle* h, hipStream_t s, const char* tag, const Predictor& p, const float* X, int ldx, int R,
                  const int* row_pos, float* tmp0, float* tmp1, float* out_rows, int prec, const void* Xp = nullptr, void* xps = nullptr, const int* Rp = nullptr) {
    char nm[96];
    const float* in = X; int ld = ldx;
    float* bufs[2] = {tmp0, tmp1};
    for (size_t l = 0; l < p.conv.size(); ++l) {
        const bool last = (l + 1 == p.conv.size());
        float* out = bufs[l & 1];
        // bf16 modes: the next layer consumes the planes; the fp32 copy is written only where a kernel pair needs it as scratch
        GemmArgs a = gemm_args(p.conv[l], in, ld, R, row_pos, (last || (xps && prec != FS2_PREC_FP32)) ? nullptr : out, p.conv[l].N);
        a.Rp = Rp;
        a.relu_pre = 1; a.ln_g = p.lng[l]; a.ln_b = p.lnb[l]; a.ln_eps = 1e-12f;
        if (last) { a.dot_w = p.lin_w; a.dot_b = p.lin_b; a.dot_out = out_rows; }
        a.scratch = out;
        // planes: the first layer's input from the caller (or built in xps), later layers' from the row kernel of the
        // previous one (it runs after the GEMM that read xps, so the buffer can be reused in place)
        a.Xp = Xp; a.xp_scratch = xps;
        if (xps && !last) { a.Yp = xps; a.yp_chunks = round_up(p.conv[l].N, 32) / 32; }
        if (xps && !last) Xp = xps; else Xp = nullptr;
        snprintf(nm, sizeof nm, "%s.conv%d", tag, (int)l);
        int rc = launch_gemm(h, s, nm, a, prec);
        if (rc) return rc;
        in = out; ld = p.conv[l].N;
    }
    return FS2_OK;
}

// The pitch and the energy predictor of the bf16 modes as one launch per layer (FusedPredictors).  Every output column is accumulated in the
// order of the separate launches; the LayerNorm of a 256-column group may be summed in another order than the two-wave form (1e-7).
int run_predictors_fused(fs2_handle* h, hipStream_t s, const FusedPredictors& v, const float* X, int ldx, int R, const int* row_pos, const int* Rp,
                         const void* Xp, void* xps, float* vp, float* vs, float* e_rows, float* p_rows, int prec) {
    const int chans = v.c0.N / 2;
    int rc;
    const long regime = h->cur_regime ? h->cur_regime : R;
    const bool row8 = opts().row8 >= 0 ? opts().row8 != 0 : (regime + 127) / 128 >= 128;
    if (row8 && (chans == 256 || chans == 384)) {
        // big grids: layer 0 as two row-complete launches (the stacked 512-column form exists at 128-row tiles only -- 128 accumulator
        // registers -- and pays a nearly empty second round where the 192-row form does not: c3 0.169 ms against 2 x 0.07), each filling its half
        // of the stacked plane rows
        const Predictor* ps[2] = {&h->energy, &h->pitch};
        for (int g = 0; g < 2; ++g) {
            GemmArgs a = gemm_args(ps[g]->conv[0], X, ldx, R, row_pos, nullptr, chans);
            a.Rp = Rp; a.relu_pre = 1; a.ln_g = ps[g]->lng[0]; a.ln_b = ps[g]->lnb[0]; a.ln_eps = 1e-12f;
            a.Xp = Xp; a.xp_scratch = xps; a.Yp = vp; a.yp_chunks = v.c0.N / 32; a.yp_col_off = g * chans; a.scratch = vs;
            if ((rc = launch_gemm(h, s, g ? "pitch.conv0" : "energy.conv0", a, prec))) return rc;
        }
    } else {
        GemmArgs a = gemm_args(v.c0, X, ldx, R, row_pos, nullptr, v.c0.N);
        a.Rp = Rp; a.relu_pre = 1; a.ln_g = v.ln0g; a.ln_b = v.ln0b; a.ln_eps = 1e-12f; a.ln_groups = 2;
        a.Xp = Xp; a.xp_scratch = xps; a.Yp = vp; a.yp_chunks = v.c0.N / 32; a.scratch = vs;
        if ((rc = launch_gemm(h, s, "var.conv0", a, prec))) return rc;
    }
    GemmArgs a = gemm_args(v.c1, nullptr, v.c0.N, R, row_pos, nullptr, v.c1.N);
    a.Rp = Rp; a.relu_pre = 1; a.ln_g = v.ln1g; a.ln_b = v.ln1b; a.ln_eps = 1e-12f; a.ln_groups = 2; a.k_groups = 2;
    a.Xp = vp; a.xp_row_chunks = v.c0.N / 32; a.scratch = vs;
    a.dot_w = v.lin_w; a.dot_b = v.lin_b; a.dot_out = e_rows; a.dot_gstride = (int)(p_rows - e_rows);
    return launch_gemm(h, s, "var.conv1", a, prec);
}

// ------------------------------------------------------------------ weight loading
// src [rows, cols] -> dst with every block of dk rows (by_cols: columns) moved to a block of dkp (dst pre-zeroed)
__global__ void pad_heads(const float* src, int rows, int cols, int dk, int dkp, int by_cols, float* dst) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
    if (by_cols) dst[(size_t)r * ((cols / dk) * dkp) + (c / dk) * dkp + c % dk] = src[i];
    else dst[((size_t)(r / dk) * dkp + r % dk) * cols + c] = src[i];
}

struct Loader {
    fs2_handle* h; hipStream_t s;
    std::map<std::string, const fs2_tensor_desc*> m;
    int rc = FS2_OK;
    unsigned* absmax_scratch = nullptr;
    std::vector<fs2_tensor_desc*> synth;      // head-padded copies of projection weights (padded_tensor): descriptors + device memory, released with the loader
    std::vector<std::string*> synth_names;
    std::vector<void*> synth_mem;
    ~Loader() {
        if (absmax_scratch) hipFree(absmax_scratch);
        if (!synth_mem.empty()) hipStreamSynchronize(s);
        for (void* p : synth_mem) hipFree(p);
        for (auto* d : synth) delete d;
        for (auto* n : synth_names) delete n;
    }
    // Registers "<name>#pad": the tensor `name` ([D, C] weight, or [D] bias when C == 0) with every head's dk rows (pad_cols: columns)
    // moved to a block of dkp, zeros in between.  Returns the new name ("" on error).
    std::string padded_tensor(const std::string& name, int heads, int dk, int dkp, int D, int C, bool pad_cols) {
        const fs2_tensor_desc* d = C ? get(name, {D, pad_cols ? heads * dk : C}) : get(name, {D});
        if (!d) return "";
        const int Dp = heads * dkp;
        const size_t n = C ? (pad_cols ? (size_t)D * Dp : (size_t)Dp * C) : (size_t)Dp;
        float* p = nullptr;
        if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) { if (!rc) rc = fail(h, FS2_ERR_HIP, "hipMalloc failed"); return ""; }
        synth_mem.push_back(p);
        hipMemsetAsync(p, 0, n * sizeof(float), s);
        const int rows = pad_cols ? D : heads * dk, cols = C ? (pad_cols ? heads * dk : C) : 1;
        hipLaunchKernelGGL(pad_heads, dim3((unsigned)(((size_t)rows * cols + 255) / 256)), dim3(256), 0, s, (const float*)d->data, rows, cols, dk, dkp, pad_cols ? 1 : 0, p);
        auto* nd = new fs2_tensor_desc(*d);
        auto* nn = new std::string(name + "#pad");
        nd->name = nn->c_str(); nd->data = p;
        if (C) { nd->shape[0] = pad_cols ? D : Dp; nd->shape[1] = pad_cols ? Dp : C; } else nd->shape[0] = Dp;
        synth.push_back(nd); synth_names.push_back(nn);
        m[*nn] = nd;
        return *nn;
    }
    const fs2_tensor_desc* get(const std::string& name, std::initializer_list<int64_t> shape) {
        auto it = m.find(name);
        if (it == m.end()) { if (!rc) rc = fail(h, FS2_ERR_WEIGHT, "missing tensor %s", name.c_str()); return nullptr; }
        const fs2_tensor_desc* d = it->second;
        bool ok = d->ndim == (int)shape.size();
        int i = 0;
        for (int64_t v : shape) { if (ok && d->shape[i] != v) ok = false; ++i; }
        if (!ok) { if (!rc) rc = fail(h, FS2_ERR_WEIGHT, "tensor %s has the wrong shape", name.c_str()); return nullptr; }
        return d;
    }
    float* dalloc(size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float)) != hipSuccess) { if (!rc) rc = fail(h, FS2_ERR_HIP, "hipMalloc of %zu floats failed", n); return nullptr; }
        h->allocs.push_back(p);
        return (float*)p;
    }
    float* copy(const std::string& name, std::initializer_list<int64_t> shape) {
        const fs2_tensor_desc* d = get(name, shape);
        if (!d) return nullptr;
        size_t n = 1;
        for (int64_t v : shape) n *= (size_t)v;
        float* p = dalloc(n);
        if (p && hipMemcpyAsync(p, d->data, n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess && !rc) rc = fail(h, FS2_ERR_HIP, "copy of %s failed", name.c_str());
        return p;
    }
    // two reference tensors of n elements each, back to back (parameters of stacked layers)
    float* copy2(const std::string& a_, const std::string& b_, std::initializer_list<int64_t> shape) {
        const fs2_tensor_desc *da = get(a_, shape), *db = get(b_, shape);
        if (!da || !db) return nullptr;
        size_t n = 1;
        for (int64_t v : shape) n *= (size_t)v;
        float* p = dalloc(2 * n);
        if (p) {
            hipMemcpyAsync(p, da->data, n * sizeof(float), hipMemcpyDeviceToDevice, s);
            hipMemcpyAsync(p + n, db->data, n * sizeof(float), hipMemcpyDeviceToDevice, s);
        }
        return p;
    }
    // parts: weights stacked along N (q|k|v); each [n_i, C, k] (k omitted for Linear)
    // col_off / src_cols: take columns [col_off, col_off + C) of a Linear weight stored [Neach, src_cols] (concat_linear halves)
    Gemm gemm(std::vector<std::string> wnames, std::vector<std::string> bnames, int Neach, int C, int k, bool linear,
              const std::string& bn_prefix = "", bool f16_image = false, int col_off = 0, int src_cols = 0, bool mx_k1 = false) {
        Gemm g;
        const int parts = (int)wnames.size();
        g.N = Neach * parts; g.C = C; g.ktaps = k; g.Cpad = round_up(C, kBK);
        const int Npad = round_up(g.N, 128);
        g.w = dalloc((size_t)Npad * k * g.Cpad);
        if (!g.w) return g;
        hipMemsetAsync(g.w, 0, (size_t)Npad * k * g.Cpad * sizeof(float), s);
        const float *bg = nullptr, *bb = nullptr, *bm = nullptr, *bv = nullptr;
        if (!bn_prefix.empty()) {
            const fs2_tensor_desc *dg = get(bn_prefix + ".weight", {Neach}), *db = get(bn_prefix + ".bias", {Neach}),
                                  *dm = get(bn_prefix + ".running_mean", {Neach}), *dv = get(bn_prefix + ".running_var", {Neach});
            if (!dg || !db || !dm || !dv) return g;
            bg = (const float*)dg->data; bb = (const float*)db->data; bm = (const float*)dm->data; bv = (const float*)dv->data;
        }
        const int nchunks = g.Cpad / 32;
        const size_t wb_elems = (size_t)Npad * k * nchunks * 64;
        {
            void* pb = nullptr;
            if (hipMalloc(&pb, wb_elems * 2) != hipSuccess) { if (!rc) rc = fail(h, FS2_ERR_HIP, "hipMalloc of bf16 weights failed"); return g; }
            h->allocs.push_back(pb);
            g.wb = pb;
            hipMemsetAsync(pb, 0, wb_elems * 2, s);
        }
        if (((f16_image && k > 1 && !linear) || (mx_k1 && k == 1 && !src_cols)) && parts == 1 && g.N % 128 == 0 && C % 128 == 0 && bn_prefix.empty()) {      // "mx" mode: its weight image (gemm_mx.h)
            const fs2_tensor_desc* d = (k == 1 && linear) ? get(wnames[0], {Neach, C}) : get(wnames[0], {Neach, C, k});
            if (!d) return g;
            if (!absmax_scratch && hipMalloc((void**)&absmax_scratch, 16) != hipSuccess) { if (!rc) rc = fail(h, FS2_ERR_HIP, "hipMalloc failed"); return g; }
            g.kw = fp8_scale_exponent(device_absmax(s, (const float*)d->data, (int64_t)Neach * C * k, absmax_scratch));
            const size_t bytes = mx_image_bytes(Npad, C, k);
            void* pm = nullptr;
            if (hipMalloc(&pm, bytes) != hipSuccess) { if (!rc) rc = fail(h, FS2_ERR_HIP, "hipMalloc of the mx weight image failed"); return g; }
            h->allocs.push_back(pm);
            g.wm = pm;
            hipLaunchKernelGGL(repack_weight_mx, dim3((unsigned)((bytes / 2 + 255) / 256)), dim3(256), 0, s, (const float*)d->data, g.N, C, k, Npad, g.kw,
                               reinterpret_cast<unsigned short*>(pm));
            if (k > 1) {      // the mx4 image of the same convolution (fp4 cross terms, one scale byte per output channel)
                const size_t b4 = mx4_image_bytes(Npad, C, k), bs = mx4_scale_image_bytes(Npad, C, k);
                void *p4 = nullptr, *ps = nullptr;
                if (hipMalloc(&p4, b4) != hipSuccess || hipMalloc(&ps, bs) != hipSuccess) { if (!rc) rc = fail(h, FS2_ERR_HIP, "hipMalloc of the mx4 weight image failed"); return g; }
                h->allocs.push_back(p4); h->allocs.push_back(ps);
                g.wm4 = p4; g.ws4 = reinterpret_cast<unsigned char*>(ps);
                hipMemsetAsync(ps, 127, bs, s);
                hipLaunchKernelGGL(repack_weight_mx4, dim3((unsigned)((b4 / 4 + 255) / 256)), dim3(256), 0, s, (const float*)d->data, g.N, C, k, Npad, g.ws4,
                                   reinterpret_cast<unsigned*>(p4));
            }
        }
        if (f16_image) {
            void* pf = nullptr;
            if (hipMalloc(&pf, wb_elems * 2) != hipSuccess) { if (!rc) rc = fail(h, FS2_ERR_HIP, "hipMalloc of fp16 weights failed"); return g; }
            h->allocs.push_back(pf);
            g.wf = pf;
            hipMemsetAsync(pf, 0, wb_elems * 2, s);
        }
        for (int p = 0; p < parts; ++p) {
            const fs2_tensor_desc* d = src_cols ? get(wnames[p], {Neach, src_cols}) : (linear ? get(wnames[p], {Neach, C}) : get(wnames[p], {Neach, C, k}));
            if (!d) return g;
            const float* wsrc = (const float*)d->data + col_off;
            {
                const int64_t tb = (int64_t)Neach * k * nchunks * 32;
                hipLaunchKernelGGL(repack_weight_bf16, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, s, wsrc, Neach, C, k,
                                   Neach, nchunks, bg, bv, 1e-5f, reinterpret_cast<__bf16*>(g.wb) + (size_t)p * Neach * k * nchunks * 64, 0, src_cols);
                if (g.wf)
                    hipLaunchKernelGGL(repack_weight_bf16, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, s, wsrc, Neach, C, k,
                                       Neach, nchunks, bg, bv, 1e-5f, reinterpret_cast<__bf16*>(g.wf) + (size_t)p * Neach * k * nchunks * 64, 1, src_cols);
            }
            const int64_t total = (int64_t)Neach * k * g.Cpad;
            hipLaunchKernelGGL(repack_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, wsrc, Neach, C, k,
                               Neach, g.Cpad, bg, bv, 1e-5f, g.w + (size_t)p * Neach * k * g.Cpad, src_cols);
        }
        if (!bn_prefix.empty() && k > 1 && parts == 1 && g.N % 128 == 0 && C % 128 == 0) {
            // mx image of a BatchNorm-folded convolution (the Postnet's 512 -> 512 layers), from the library's own folded fp32 image; its operand is a tanh output:
            // |x| <= 1 is the a-priori bound of the static activation scale (kPostKa)
            if (!absmax_scratch && hipMalloc((void**)&absmax_scratch, 16) != hipSuccess) { if (!rc) rc = fail(h, FS2_ERR_HIP, "hipMalloc failed"); return g; }
            g.kw = fp8_scale_exponent(device_absmax(s, g.w, (int64_t)Npad * k * g.Cpad, absmax_scratch));
            const size_t bytes = mx_image_bytes(Npad, C, k);
            void* pm = nullptr;
            if (hipMalloc(&pm, bytes) != hipSuccess) { if (!rc) rc = fail(h, FS2_ERR_HIP, "hipMalloc of the mx weight image failed"); return g; }
            h->allocs.push_back(pm);
            g.wm = pm;
            hipLaunchKernelGGL(repack_weight_mx, dim3((unsigned)((bytes / 2 + 255) / 256)), dim3(256), 0, s, g.w, g.N, C, k, Npad, g.kw, reinterpret_cast<unsigned short*>(pm), g.Cpad);
        }
        if (!bnames.empty()) {
            g.bias = dalloc(g.N);
            if (!g.bias) return g;
            for (int p = 0; p < parts; ++p) {
                const fs2_tensor_desc* d = get(bnames[p], {Neach});
                if (!d) return g;
                hipMemcpyAsync(g.bias + (size_t)p * Neach, d->data, Neach * sizeof(float), hipMemcpyDeviceToDevice, s);
            }
        } else if (bg) {
            g.bias = dalloc(g.N);
            if (!g.bias) return g;
            hipLaunchKernelGGL(bn_fold_bias, dim3((Neach + 255) / 256), dim3(256), 0, s, bg, bb, bm, bv, 1e-5f, Neach, g.bias);
        }
        return g;
    }
};

void load_stack(Loader& L, Stack& st, const std::string& pre, int nlayers, int D, int heads, int units, int k, const std::string& pe_pre,
                bool scaled, bool pre_ln, bool concat) {
    st.layers.resize(nlayers);
    st.dk = D / heads;
    const int dkp = padded_head_dim(st.dk);
    st.Dp = heads * dkp;
    st.pre_ln = pre_ln; st.concat = concat;
    if (pre_ln) { st.after_g = L.copy(pre + ".after_norm.weight", {D}); st.after_b = L.copy(pre + ".after_norm.bias", {D}); }
    for (int i = 0; i < nlayers; ++i) {
        const std::string p = pre + ".encoders_." + std::to_string(i);
        Layer& ly = st.layers[i];
        if (st.Dp == D) {
            ly.qkv = L.gemm({p + ".self_attn.linear_q.weight", p + ".self_attn.linear_k.weight", p + ".self_attn.linear_v.weight"},
                            {p + ".self_attn.linear_q.bias", p + ".self_attn.linear_k.bias", p + ".self_attn.linear_v.bias"}, D, D, 1, true);
            ly.out = L.gemm({p + ".self_attn.linear_out.weight"}, {p + ".self_attn.linear_out.bias"}, D, D, 1, true);
        } else {      // head dim padded to a kernel size: zero rows in W_q / W_k / W_v (and their biases), zero columns in W_out
            std::vector<std::string> wn, bn;
            for (const char* t : {"q", "k", "v"}) {
                wn.push_back(L.padded_tensor(p + ".self_attn.linear_" + t + ".weight", heads, st.dk, dkp, D, D, false));
                bn.push_back(L.padded_tensor(p + ".self_attn.linear_" + t + ".bias", heads, st.dk, dkp, D, 0, false));
            }
            const std::string wo = L.padded_tensor(p + ".self_attn.linear_out.weight", heads, st.dk, dkp, D, D, true);
            if (L.rc) return;
            ly.qkv = L.gemm(wn, bn, st.Dp, D, 1, true);
            ly.out = L.gemm({wo}, {p + ".self_attn.linear_out.bias"}, D, st.Dp, 1, true);
        }
        ly.w1 = L.gemm({p + ".feed_forward.w_1.weight"}, {p + ".feed_forward.w_1.bias"}, units, D, k, L.h->cfg.ffn_kernel == 1 && L.m.count(p + ".feed_forward.w_1.weight") && L.m[p + ".feed_forward.w_1.weight"]->ndim == 2,
                        "", /*f16_image=*/k > 1);
        {
            const bool lin2 = L.m.count(p + ".feed_forward.w_2.weight") && L.m[p + ".feed_forward.w_2.weight"]->ndim == 2;
            // (mx image of w_2 where the one-wave-per-SIMD row kernel exists for this width: gemm_row4.h is instantiated for N = 384)
            ly.w2 = L.gemm({p + ".feed_forward.w_2.weight"}, {p + ".feed_forward.w_2.bias"}, D, units, 1, lin2, "", false, 0, 0, /*mx_k1=*/k > 1 && D == 384);
        }
        if (concat) {
            ly.cat_x = L.gemm({p + ".concat_linear.weight"}, {p + ".concat_linear.bias"}, D, D, 1, true, "", false, /*col_off=*/0, /*src_cols=*/2 * D);
            ly.cat_a = L.gemm({p + ".concat_linear.weight"}, {}, D, D, 1, true, "", false, /*col_off=*/D, /*src_cols=*/2 * D);
        }
        ly.ln1g = L.copy(p + ".norm1.weight", {D}); ly.ln1b = L.copy(p + ".norm1.bias", {D});
        if (ly.w1.wm && ly.ln1g && ly.ln1b && L.absmax_scratch) {      // a-priori bound of the LayerNorm output that feeds the FFN conv
            const float gm = device_absmax(L.s, ly.ln1g, D, L.absmax_scratch), bm = device_absmax(L.s, ly.ln1b, D, L.absmax_scratch);
            const float xmax = std::sqrt((float)D) * gm + bm;
            ly.ka = fp8_scale_exponent(xmax);
            if (ly.w2.wm) {      // the hidden layer's bound: l1 norms of w_1's rows (the fp32 tensor the caller handed in is still there during the load)
                const fs2_tensor_desc* dw = L.m.count(p + ".feed_forward.w_1.weight") ? L.m[p + ".feed_forward.w_1.weight"] : nullptr;
                if (dw) ly.kh = fp8_scale_exponent(device_row_l1_bound(L.s, (const float*)dw->data, ly.w1.bias, units, D * k, xmax, L.absmax_scratch));
                else ly.w2.wm = nullptr;
            }
        }
        else ly.w2.wm = nullptr;
        ly.ln2g = L.copy(p + ".norm2.weight", {D}); ly.ln2b = L.copy(p + ".norm2.bias", {D});
    }
    auto it = L.m.find(pe_pre + ".pe");
    if (it == L.m.end() || it->second->ndim != 3 || it->second->shape[2] != D) {
        if (!L.rc) L.rc = fail(L.h, FS2_ERR_WEIGHT, "missing/odd positional table %s.pe", pe_pre.c_str());
        return;
    }
    st.pe_rows = (int)it->second->shape[1];
    st.pe = L.copy(pe_pre + ".pe", {1, st.pe_rows, D});
    st.alpha = scaled ? L.copy(pe_pre + ".alpha", {}) : nullptr;
}

void load_predictor(Loader& L, Predictor& p, const std::string& pre, int nlayers, int idim, int chans, int k) {
    p.conv.clear(); p.lng.clear(); p.lnb.clear();
    for (int l = 0; l < nlayers; ++l) {
        const std::string c = pre + ".conv." + std::to_string(l);
        p.conv.push_back(L.gemm({c + ".0.weight"}, {c + ".0.bias"}, chans, l == 0 ? idim : chans, k, false));
        p.lng.push_back(L.copy(c + ".2.layer_norm.weight", {chans}));
        p.lnb.push_back(L.copy(c + ".2.layer_norm.bias", {chans}));
    }
    p.lin_w = L.copy(pre + ".linear.weight", {1, chans});
    p.lin_b = L.copy(pre + ".linear.bias", {1});
}

void free_weights(fs2_handle* h) {
    for (void* p : h->allocs) hipFree(p);
    h->allocs.clear();
    h->loaded = false;
}

int check_batch(fs2_handle* h, const fs2_batch& b) {
    if (b.B <= 0 || b.Tmax <= 0 || !b.ilens) return fail(h, FS2_ERR_ARG, "batch: B=%d Tmax=%d ilens=%p", b.B, b.Tmax, (const void*)b.ilens);
    for (int i = 0; i < b.B; ++i)
        if (b.ilens[i] <= 0 || b.ilens[i] > b.Tmax) return fail(h, FS2_ERR_ARG, "ilens[%d]=%lld outside [1,%d]", i, (long long)b.ilens[i], b.Tmax);
    if (b.precision < FS2_PREC_FP32 || b.precision > FS2_PREC_MIX_MX4) return fail(h, FS2_ERR_ARG, "unknown precision mode %d", b.precision);
    if (b.regime_tokens < 0 || b.regime_utterances < 0 || (b.regime_tokens > 0) != (b.regime_utterances > 0))
        return fail(h, FS2_ERR_ARG, "batch: regime_tokens=%lld / regime_utterances=%d must be given together (0 / 0 = this call's own batch)", (long long)b.regime_tokens, b.regime_utterances);
    return FS2_OK;
}

// The row counts the kernel-variant choices are functions of (fs2.h: fs2_batch.regime_*): ESTIMATES of the packed rows of the batch the variants
// are chosen for -- this call's own, or the larger one the caller names -- at token level (one row per phoneme) and at frame level (8 frames per
// phoneme; LJSpeech: 7.9), plus the alignment and gap rows of every utterance.  Functions of (phonemes, utterances) only: the host knows both in
// the host- and in the device-driven layout, and a shard that names the whole batch gets the whole batch's numbers.  Any value gives correct
// results; the same value gives the same kernels, hence the same bits.
inline long regime_tokens_of(const fs2_batch& b, bool token_level) {
    if (b.regime_tokens > 0) return (long)b.regime_tokens;
    long n = 0;
    for (int i = 0; i < b.B; ++i) n += (token_level && b.compat_padded) ? (long)b.Tmax : (long)b.ilens[i];      // (padded-batch semantics: the encoder runs on Tmax rows per utterance)
    return n;
}
inline int regime_rows_of(const fs2_batch& b, bool token_level) {
    const long utts = b.regime_utterances > 0 ? b.regime_utterances : b.B;
    return (int)std::min<long>((token_level ? 1 : 8) * regime_tokens_of(b, token_level) + utts * (kGap + kAttAlign) + kGap, INT32_MAX - 256);
}

void token_layout(const fs2_batch& b, HostLayout& L, int gap) {
    std::vector<int> len(b.B), klen(b.B), vlen(b.B);
    for (int i = 0; i < b.B; ++i) {
        vlen[i] = klen[i] = (int)b.ilens[i];
        len[i] = b.compat_padded ? b.Tmax : (int)b.ilens[i];
    }
    build_layout(L, b.B, len, klen, vlen, gap);
}

struct TokenPlan {   // offsets inside the token workspace
    size_t total;
};

// carve the token workspace; if ws == nullptr only sizes it
size_t carve_tokens(const fs2_config& c, const fs2_batch& b, const HostLayout& L, void* ws, size_t cap, int** meta, StackBufs* sb,
                    float** p0, float** p1, float** dlog_rows, int64_t** dint, int** cum, int** olens32, bool* ok, float** kp = nullptr, size_t* kp_cap = nullptr) {
    Bump bp(ws, cap);
    const size_t R = L.Rpad;
    int* m = bp.take<int>(layout_dev_ints(L));
    float* x0 = bp.take<float>(R * c.adim);
    float* x1 = bp.take<float>(R * c.adim);
    const size_t eDp = att_width(c.adim, c.aheads);      // attention width: heads x padded head dim
    float* qkv = bp.take<float>(R * 3 * eDp);
    float* ctx = bp.take<float>(R * eDp);
    float* hid = bp.take<float>(R * (size_t)round_up(c.eunits, 32));     // bf16 modes: hidden layer as planes (32-channel chunks)
    __bf16* qkh = bp.take<__bf16>(R * 2 * eDp);
    __bf16* qkl = bp.take<__bf16>(R * 2 * eDp);
    __bf16* vth = bp.take<__bf16>(R * eDp);
    __bf16* vtl = bp.take<__bf16>(R * eDp);
    const size_t pw = round_up(c.adim, 32);
    float* x0p = bp.take<float>(R * pw);
    float* x1p = bp.take<float>(R * pw);
    float* xps = bp.take<float>(R * (size_t)round_up(std::max(c.adim, c.dur_chans), 32));
    float* q0 = bp.take<float>(R * c.dur_chans);
    float* q1 = bp.take<float>(R * c.dur_chans);
    float* dl = bp.take<float>(R);
    int64_t* di = bp.take<int64_t>((size_t)b.B * b.Tmax);
    int* cu = bp.take<int>((size_t)b.B * b.Tmax);
    int* o32 = bp.take<int>(b.B);
    const size_t kcap = (size_t)4 * std::min<size_t>(R, kSplitRows) * 1024;
    float* kpb = bp.take<float>(kcap);
    if (kp) *kp = kpb;
    if (kp_cap) *kp_cap = kcap;
    if (meta) *meta = m;
    if (sb) { sb->x0 = x0; sb->x1 = x1; sb->qkv = qkv; sb->ctx = ctx; sb->hid = hid; sb->qkh = qkh; sb->qkl = qkl; sb->vth = vth; sb->vtl = vtl; sb->x0p = x0p; sb->x1p = x1p; sb->xps = xps; }
    if (p0) *p0 = q0;
    if (p1) *p1 = q1;
    if (dlog_rows) *dlog_rows = dl;
    if (dint) *dint = di;
    if (cum) *cum = cu;
    if (olens32) *olens32 = o32;
    if (ok) *ok = bp.ok();
    return align_up(bp.off, 256);
}

struct FrameBufs {
    int* meta; float* hfr; float *t0, *t1, *e_rows, *p_rows; StackBufs sb; float *before, *after; int *qe, *qp, *lri;
    float *vp = nullptr, *vs = nullptr;      // fused pitch + energy predictors: planes of the stacked hidden layer [R][2 chans], fp32 scratch [R, 2 chans]
    float* kp; size_t kp_cap;
    // reduction_factor r > 1: the Postnet runs on r rows per decoder row (its own row metadata and ping-pong buffers)
    int* meta2 = nullptr; float *pa2 = nullptr, *pb2 = nullptr, *xps2 = nullptr;
};

size_t carve_frames(const fs2_config& c, const HostLayout& L, void* ws, size_t cap, FrameBufs* fb, bool* ok) {
    Bump bp(ws, cap);
    const size_t R = L.Rpad;
    FrameBufs f;
    f.meta = bp.take<int>(layout_dev_ints(L));
    f.hfr = bp.take<float>(R * c.adim);
    f.t0 = bp.take<float>(R * c.var_chans);
    f.t1 = bp.take<float>(R * c.var_chans);
    f.e_rows = bp.take<float>(R);
    f.p_rows = bp.take<float>(R);
    f.vp = bp.take<float>(R * 2 * (size_t)c.var_chans);
    f.vs = bp.take<float>(R * 2 * (size_t)c.var_chans);
    f.sb.x0 = bp.take<float>(R * c.ddim);
    f.sb.x1 = bp.take<float>(R * c.ddim);
    const size_t dDp = att_width(c.ddim, c.aheads);
    f.sb.qkv = bp.take<float>(R * 3 * dDp);
    f.sb.ctx = bp.take<float>(R * dDp);
    f.sb.hid = bp.take<float>(R * (size_t)std::max(round_up(c.dunits, 32), 2 * c.postnet_chans));
    f.sb.qkh = bp.take<__bf16>(R * 2 * dDp);
    f.sb.qkl = bp.take<__bf16>(R * 2 * dDp);
    f.sb.vth = bp.take<__bf16>(R * dDp);
    f.sb.vtl = bp.take<__bf16>(R * dDp);
    f.sb.x0p = bp.take<float>(R * (size_t)round_up(std::max(c.ddim, c.postnet_chans), 32));
    f.sb.x1p = bp.take<float>(R * (size_t)round_up(std::max(std::max(c.adim, c.ddim), c.postnet_chans), 32));    // also holds the length-regulator output's planes
    f.sb.xps = bp.take<float>(R * (size_t)round_up(std::max(std::max(c.adim, c.ddim), std::max(std::max(c.var_chans, c.postnet_chans), c.odim)), 32));
    f.sb.xs4 = bp.take<unsigned char>(R * 32 + 64);      // mx4: one scale byte per (row, 16-channel slot of a cross unit), 32 per row
    const size_t rf = (size_t)std::max(c.reduction_factor, 1);
    f.before = bp.take<float>(R * rf * c.odim);
    f.after = bp.take<float>(R * rf * c.odim);
    if (rf > 1) {
        f.meta2 = bp.take<int>(4 * (size_t)L.B + 2 * R * rf + 64);
        f.pa2 = bp.take<float>(R * rf * c.postnet_chans);
        f.pb2 = bp.take<float>(R * rf * c.postnet_chans);
        f.xps2 = bp.take<float>(R * rf * (size_t)round_up(std::max(c.postnet_chans, c.odim), 32));
    }
    f.qe = bp.take<int>(R);
    f.qp = bp.take<int>(R);
    f.lri = bp.take<int>(R);
    f.kp_cap = (size_t)4 * std::min<size_t>(R, kSplitRows) * 1024;
    f.kp = bp.take<float>(f.kp_cap);
    if (fb) *fb = f;
    if (ok) *ok = bp.ok();
    return align_up(bp.off, 256);
}

void frame_layout(const fs2_batch& b, const int64_t* olens, int masked, HostLayout& L, int gap) {
    std::vector<int> len(b.B), klen(b.B), vlen(b.B);
    int mx = 0;
    for (int i = 0; i < b.B; ++i) mx = std::max(mx, (int)olens[i]);
    for (int i = 0; i < b.B; ++i) {
        vlen[i] = (int)olens[i];
        len[i] = b.compat_padded ? mx : vlen[i];
        klen[i] = b.compat_padded ? (masked ? vlen[i] : mx) : vlen[i];
    }
    build_layout(L, b.B, len, klen, vlen, gap);
}

template <typename T>
int unpack(fs2_handle* h, hipStream_t s, const T* src, int W, const int* start, const int* limit, int B, int Lout, T* dst, T fill,
           const int* ovf = nullptr, bool* ovf_handled = nullptr) {
    const int64_t total = (int64_t)B * Lout * W;
    if (ovf_handled) *ovf_handled = false;
    if (total == 0) return FS2_OK;
    if constexpr (std::is_same<T, float>::value) {
        if (W % 4 == 0 && fill == 0.f && (reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) % 16 == 0) {
            hipLaunchKernelGGL(unpack_rows4, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, src, W / 4, start, limit, B, Lout, dst, ovf);
            if (ovf_handled) *ovf_handled = true;
            HIP_TRY(h, hipGetLastError());
            return FS2_OK;
        }
    }
    hipLaunchKernelGGL(unpack_rows<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, W, start, limit, B, Lout, dst, fill);
    HIP_TRY(h, hipGetLastError());
    return FS2_OK;
}

}  // namespace

// =====================================================================================================
extern "C" {

int32_t fs2_abi_version(void) { return FS2_ABI_VERSION; }

// every caller-filled struct starts with its own sizeof as the caller's header saw it: a binding built against another revision
// of include/fs2.h is refused here instead of having its fields read at the wrong offsets
#define ABI_CHECK(h, what, ptr, type)                                                                                         \
    do {                                                                                                                      \
        if ((ptr)->struct_size != (uint32_t)sizeof(type))                                                                     \
            return fail(h, FS2_ERR_ARG, "%s: " #type ".struct_size is %u but this library (ABI %d) expects %zu: the binding " \
                        "does not match include/fs2.h", what, (unsigned)(ptr)->struct_size, FS2_ABI_VERSION, sizeof(type));     \
    } while (0)

int fs2_create(const fs2_config* cfg, fs2_handle** out) {
    if (!cfg || !out) return fail(nullptr, FS2_ERR_ARG, "fs2_create: null argument");
    *out = nullptr;
    ABI_CHECK(nullptr, "fs2_create", cfg, fs2_config);
    if (cfg->reduction_factor < 1 || cfg->reduction_factor > 8) return fail(nullptr, FS2_ERR_UNSUPPORTED, "reduction_factor %d outside [1, 8]", cfg->reduction_factor);
    if (cfg->aheads <= 0 || cfg->adim % cfg->aheads || cfg->ddim % cfg->aheads) return fail(nullptr, FS2_ERR_ARG, "adim/ddim not divisible by aheads");
    if (!padded_head_dim(cfg->adim / cfg->aheads) || !padded_head_dim(cfg->ddim / cfg->aheads))
        return fail(nullptr, FS2_ERR_UNSUPPORTED, "attention head dims above 256 (adim / aheads = %d, ddim / aheads = %d) are not implemented", cfg->adim / cfg->aheads, cfg->ddim / cfg->aheads);
    if (!cfg->decoder_input_layer && cfg->ddim != cfg->adim) return fail(nullptr, FS2_ERR_ARG, "decoder_input_layer = 0 needs ddim == adim");
    if (cfg->n_bins != cfg->adim) return fail(nullptr, FS2_ERR_ARG, "n_bins (%d) must equal adim (%d): the reference feeds one_hot(256) into Linear(adim, adim)", cfg->n_bins, cfg->adim);
    if (cfg->ffn_kernel % 2 == 0 || cfg->dur_kernel % 2 == 0 || cfg->var_kernel % 2 == 0 || (cfg->postnet_layers > 0 && cfg->postnet_filts % 2 == 0))
        return fail(nullptr, FS2_ERR_UNSUPPORTED, "even convolution kernel sizes are not supported");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, FS2_ERR_HIP, "no HIP device is visible: libfs2_hip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, FS2_ERR_ARG, "device %d out of range (%d visible)", cfg->device, ndev);
    {
        DeviceGuard g(cfg->device);      // only proves the device can be made current; the caller's current device is restored
        if (g.err != hipSuccess) return fail(nullptr, FS2_ERR_HIP, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(g.err));
    }
    fs2_handle* h = new fs2_handle();
    h->cfg = *cfg;
    {
        DeviceGuard g(cfg->device);
        if (hipMalloc((void**)&h->counters, 16 * sizeof(int)) != hipSuccess || hipMemset(h->counters, 0, 16 * sizeof(int)) != hipSuccess) {
            delete h;
            return fail(nullptr, FS2_ERR_HIP, "hipMalloc of the handle's counters failed");
        }
    }
    {   // zero rows between packed utterances: the largest conv halo of this model (launch_gemm refuses kernels > kMaxHalo + 1 taps)
        int p = std::max(std::max(cfg->ffn_kernel, cfg->dur_kernel), cfg->var_kernel);
        if (cfg->postnet_layers > 0) p = std::max(p, cfg->postnet_filts);
        h->gap = std::min(kGap, std::max(kMinGap, (p - 1) / 2));
    }
    *out = h;
    return FS2_OK;
}

void fs2_destroy(fs2_handle* h) {
    if (!h) return;
    DeviceGuard g(h->cfg.device);
    free_weights(h);
    if (h->counters) hipFree(h->counters);
    for (auto& r : h->recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    for (void* p : h->graph_pinned) hipHostFree(p);
    delete h;
}

const char* fs2_last_error(const fs2_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int fs2_load_weights(fs2_handle* h, const fs2_tensor_desc* t, int32_t n, void* stream) {
    if (!h || !t || n <= 0) return fail(h, FS2_ERR_ARG, "fs2_load_weights: bad arguments");
    DeviceGuard g(h->cfg.device);
    HIP_TRY(h, g.err);
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(h, hipStreamSynchronize(s));   // nothing may still be reading the old weights
    free_weights(h);
    h->encoded = false;
    const fs2_config& c = h->cfg;
    Loader L{h, s, {}, FS2_OK};
    for (int i = 0; i < n; ++i) if (t[i].name) L.m[t[i].name] = &t[i];

    h->enc_embed = L.copy("encoder.embed.0.weight", {c.idim, c.adim});
    load_stack(L, h->enc, "encoder", c.elayers, c.adim, c.aheads, c.eunits, c.ffn_kernel, "encoder.embed.1", c.use_scaled_pos_enc,
               c.enc_normalize_before != 0, c.enc_concat_after != 0);
    load_stack(L, h->dec, "decoder", c.dlayers, c.ddim, c.aheads, c.dunits, c.ffn_kernel, c.decoder_input_layer ? "decoder.embed.4" : "decoder.embed.0",
               c.use_scaled_pos_enc, c.dec_normalize_before != 0, c.dec_concat_after != 0);
    load_predictor(L, h->dur, "duration_predictor", c.dur_layers, c.adim, c.dur_chans, c.dur_kernel);
    load_predictor(L, h->energy, "energy_predictor.predictor", c.var_layers, c.adim, c.var_chans, c.var_kernel);
    load_predictor(L, h->pitch, "pitch_predictor.predictor", c.var_layers, c.adim, c.var_chans, c.var_kernel);
    h->var2 = FusedPredictors();
    if (c.var_layers == 2 && c.var_chans % 128 == 0 && c.adim % 32 == 0) {
        const std::string e = "energy_predictor.predictor", q = "pitch_predictor.predictor";
        FusedPredictors& v = h->var2;
        v.c0 = L.gemm({e + ".conv.0.0.weight", q + ".conv.0.0.weight"}, {e + ".conv.0.0.bias", q + ".conv.0.0.bias"}, c.var_chans, c.adim, c.var_kernel, false);
        v.c1 = L.gemm({e + ".conv.1.0.weight", q + ".conv.1.0.weight"}, {e + ".conv.1.0.bias", q + ".conv.1.0.bias"}, c.var_chans, c.var_chans, c.var_kernel, false);
        v.ln0g = L.copy2(e + ".conv.0.2.layer_norm.weight", q + ".conv.0.2.layer_norm.weight", {c.var_chans});
        v.ln0b = L.copy2(e + ".conv.0.2.layer_norm.bias", q + ".conv.0.2.layer_norm.bias", {c.var_chans});
        v.ln1g = L.copy2(e + ".conv.1.2.layer_norm.weight", q + ".conv.1.2.layer_norm.weight", {c.var_chans});
        v.ln1b = L.copy2(e + ".conv.1.2.layer_norm.bias", q + ".conv.1.2.layer_norm.bias", {c.var_chans});
        v.lin_w = L.copy2(e + ".linear.weight", q + ".linear.weight", {1, c.var_chans});
        v.lin_b = L.copy2(e + ".linear.bias", q + ".linear.bias", {1});
        v.ok = !L.rc;
    }
    h->ebins = L.copy("energy_predictor.energy_bins", {c.n_bins - 1});
    h->pbins = L.copy("pitch_predictor.pitch_bins", {c.n_bins - 1});
    h->Te = L.dalloc((size_t)c.n_bins * c.adim);
    h->Tp = L.dalloc((size_t)c.n_bins * c.adim);
    {
        const fs2_tensor_desc *we = L.get("energy_embed.weight", {c.adim, c.n_bins}), *be = L.get("energy_embed.bias", {c.adim});
        const fs2_tensor_desc *wp = L.get("pitch_embed.weight", {c.adim, c.n_bins}), *bp = L.get("pitch_embed.bias", {c.adim});
        if (we && be && wp && bp && h->Te && h->Tp) {
            const int tot = c.adim * c.n_bins;
            hipLaunchKernelGGL(onehot_table, dim3((tot + 255) / 256), dim3(256), 0, s, (const float*)we->data, (const float*)be->data, c.adim, c.n_bins, h->Te);
            hipLaunchKernelGGL(onehot_table, dim3((tot + 255) / 256), dim3(256), 0, s, (const float*)wp->data, (const float*)bp->data, c.adim, c.n_bins, h->Tp);
        }
    }
    if (c.decoder_input_layer) {
        h->dec_in = L.gemm({"decoder.embed.0.weight"}, {"decoder.embed.0.bias"}, c.ddim, c.adim, 1, true);
        h->dec_in_lng = L.copy("decoder.embed.1.weight", {c.ddim});
        h->dec_in_lnb = L.copy("decoder.embed.1.bias", {c.ddim});
    }
    h->feat = L.gemm({"feat_out.weight"}, {"feat_out.bias"}, c.odim * std::max(c.reduction_factor, 1), c.ddim, 1, true);
    h->post.clear();
    for (int l = 0; l < c.postnet_layers; ++l) {
        const std::string p = "postnet.postnet." + std::to_string(l);
        const int ic = (l == 0) ? c.odim : c.postnet_chans;
        const int oc = (l == c.postnet_layers - 1) ? c.odim : c.postnet_chans;
        h->post.push_back(L.gemm({p + ".0.weight"}, {}, oc, ic, c.postnet_filts, false, c.use_batch_norm ? p + ".1" : std::string()));
    }
    if (L.rc) { free_weights(h); return L.rc; }
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipStreamSynchronize(s));   // source tensors may be released by the caller after return
    h->loaded = true;
    return FS2_OK;
}

int fs2_set_profiling(fs2_handle* h, int32_t on) {
    if (!h) return FS2_ERR_ARG;
    h->prof = on != 0;
    for (auto& r : h->recs) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    h->recs.clear();
    return FS2_OK;
}

int fs2_set_profile_filter(fs2_handle* h, const char* name) {
    if (!h) return FS2_ERR_ARG;
    h->prof_filter = name ? name : "";
    return FS2_OK;
}

int fs2_get_profile(fs2_handle* h, const char** names, float* ms, double* flops, double* bytes, int32_t cap) {
    if (!h) return FS2_ERR_ARG;
    int n = 0;
    for (auto& r : h->recs) {
        if (n >= cap) break;
        hipEventSynchronize(r.e1);
        float t = 0.f;
        hipEventElapsedTime(&t, r.e0, r.e1);
        if (names) names[n] = r.name.c_str();
        if (ms) ms[n] = t;
        if (flops) flops[n] = r.flops;
        if (bytes) bytes[n] = r.bytes;
        ++n;
    }
    return n;
}

size_t fs2_token_workspace_bytes(const fs2_handle* h, const fs2_batch* b) {
    if (!h || !b || b->B <= 0 || !b->ilens) return 0;
    HostLayout L;
    token_layout(*b, L, h->gap);
    return carve_tokens(h->cfg, *b, L, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int fs2_encode(fs2_handle* h, void* stream, const fs2_encode_io* io) {
    if (!h || !io) return fail(h, FS2_ERR_ARG, "fs2_encode: null argument");
    ABI_CHECK(h, "fs2_encode", io, fs2_encode_io);
    if (!h->loaded) return fail(h, FS2_ERR_STATE, "fs2_encode: weights not loaded");
    int rc = check_batch(h, io->batch);
    if (rc) return rc;
    if (!io->xs || !io->olens || !io->workspace) return fail(h, FS2_ERR_ARG, "fs2_encode: xs/olens/workspace must be given");
    if (io->duration_alpha < 0.f) return fail(h, FS2_ERR_ARG, "fs2_encode: duration_alpha %g must be > 0 (0 = 1.0)", (double)io->duration_alpha);
    DeviceGuard g(h->cfg.device);
    HIP_TRY(h, g.err);
    hipStream_t s = (hipStream_t)stream;
    const fs2_config& c = h->cfg;
    const fs2_batch& b = io->batch;
    const int prec = base_precision(b.precision), ffn_terms = ffn_f16_terms(b.precision);
    h->encoded = false;
    token_layout(b, h->tok, h->gap);
    const HostLayout& L = h->tok;
    if (L.len.size() && *std::max_element(L.len.begin(), L.len.end()) > h->enc.pe_rows)
        return fail(h, FS2_ERR_ARG, "sequence longer than the positional table (%d rows): extend `pe` and reload", h->enc.pe_rows);
    int* meta; StackBufs sb; float *p0, *p1, *dlog_rows; int64_t* dint; int* cum; int* o32; bool ok;
    carve_tokens(c, b, L, io->workspace, io->workspace_bytes, &meta, &sb, &p0, &p1, &dlog_rows, &dint, &cum, &o32, &ok, &h->kp, &h->kp_cap);
    if (!ok) return fail(h, FS2_ERR_WORKSPACE, "fs2_encode: workspace too small");
    const int tok_regime = regime_rows_of(b, /*token_level=*/true);
    h->cur_regime = tok_regime; h->cur_status = nullptr;
    if ((rc = upload_layout(h, s, L, meta, h->dtok))) return rc;
    const DevLayout& dl = h->dtok;
    const bool enc_pl = prec != FS2_PREC_FP32;   // activations also travel as planes (x0p holds those of x0 before and after the stack)
    {
        Scope sc(h, s, "enc.embed", 0, 4.0 * L.R * c.adim * 2);
        hipLaunchKernelGGL(embed_pe, dim3((L.R + 3) / 4), dim3(256), 0, s, io->xs, b.Tmax, h->enc_embed, c.idim, c.adim, h->enc.pe,
                           h->enc.alpha, c.use_scaled_pos_enc ? 1.0f : sqrtf((float)c.adim), dl.row_pos, dl.row_seq, L.R, sb.x0,
                           enc_pl && c.adim % 32 == 0 ? sb.x0p : nullptr);
        HIP_TRY(h, hipGetLastError());
    }
    if ((rc = run_stack(h, s, "enc", h->enc, c.adim, c.aheads, L.R, L, dl, /*mask_q=*/1, sb, prec, /*x0p_ready=*/enc_pl && c.adim % 32 == 0, /*allow_splitk=*/true, /*regime_rows=*/tok_regime, ffn_terms))) return rc;
    if ((rc = run_predictor(h, s, "dur", h->dur, sb.x0, c.adim, L.R, dl.row_pos, p0, p1, dlog_rows, prec, enc_pl ? sb.x0p : nullptr, sb.xps))) return rc;
    {
        Scope sc(h, s, "dur.post", 0, 0);
        const int n = b.B * b.Tmax;
        hipLaunchKernelGGL(dur_finalize, dim3((n + 255) / 256), dim3(256), 0, s, dlog_rows, dl.start, dl.vlen, b.B, b.Tmax, io->d_log, dint, io->d_int);
        HIP_TRY(h, hipGetLastError());
        hipLaunchKernelGGL(dur_scan, dim3(b.B), dim3(256), 0, s, io->ds ? io->ds : dint, b.Tmax, dl.vlen, cum, io->olens, o32,
                           io->duration_alpha > 0.f ? io->duration_alpha : 1.f, io->xs, c.idim);
        HIP_TRY(h, hipGetLastError());
    }
    if (io->enc_out) {
        if ((rc = unpack<float>(h, s, sb.x0, c.adim, dl.start, dl.vlen, b.B, b.Tmax, io->enc_out, 0.f))) return rc;
    }
    h->enc_final = sb.x0; h->cum = cum; h->o32 = o32; h->enc_B = b.B; h->enc_Tmax = b.Tmax; h->enc_compat = b.compat_padded;
    h->enc_ntok = 0;
    for (int i = 0; i < b.B; ++i) h->enc_ntok += (long)b.ilens[i];
    h->enc_ws = io->workspace;
    h->encoded = true;
    return FS2_OK;
}

size_t fs2_frame_workspace_bytes(const fs2_handle* h, const fs2_batch* b, const int64_t* olens) {
    if (!h || !b || !olens || b->B <= 0) return 0;
    HostLayout L;
    frame_layout(*b, olens, 0, L, h->gap);
    return carve_frames(h->cfg, L, nullptr, 0, nullptr, nullptr);
}

// device-driven layout: every utterance start is rounded up to kAttAlign rows and followed by kGap zero rows
int64_t fs2_row_capacity(const fs2_batch* b, int64_t total_frames_bound) {
    if (!b || b->B <= 0 || total_frames_bound <= 0) return 0;
    return round_up((int)std::min<int64_t>(total_frames_bound + (int64_t)b->B * (kGap + kAttAlign) + kGap, INT32_MAX - 256), 128);
}

void capacity_layout(const fs2_batch& b, int64_t row_capacity, int lmax_cap, HostLayout& L) {
    L.B = b.B; L.R = (int)row_capacity; L.Rpad = round_up((int)row_capacity, 128);
    // eight LPT queues: none is longer than the mean plus one utterance's items
    L.work_cap = kXcds * (((int)(row_capacity / kAttBlk) + b.B + kXcds - 1) / kXcds + lmax_cap / kAttBlk + 2);
    L.start.clear(); L.len.clear(); L.klen.clear(); L.vlen.clear(); L.work.clear();
}

size_t fs2_frame_workspace_bytes_cap(const fs2_handle* h, const fs2_batch* b, int64_t row_capacity, int32_t lmax_cap) {
    if (!h || !b || b->B <= 0 || row_capacity <= 0 || row_capacity > INT32_MAX - 256 || lmax_cap <= 0) return 0;
    HostLayout L;
    capacity_layout(*b, row_capacity, lmax_cap, L);
    return carve_frames(h->cfg, L, nullptr, 0, nullptr, nullptr);
}

int fs2_decode(fs2_handle* h, void* stream, const fs2_decode_io* io) {
    if (!h || !io) return fail(h, FS2_ERR_ARG, "fs2_decode: null argument");
    ABI_CHECK(h, "fs2_decode", io, fs2_decode_io);
    if (!h->encoded) return fail(h, FS2_ERR_STATE, "fs2_decode called without a preceding fs2_encode");
    const fs2_batch& b = io->batch;
    int rc = check_batch(h, b);
    if (rc) return rc;
    if (b.B != h->enc_B || b.Tmax != h->enc_Tmax || b.compat_padded != h->enc_compat || io->token_workspace != h->enc_ws)
        return fail(h, FS2_ERR_STATE, "fs2_decode: batch does not match the preceding fs2_encode");
    const bool devlay = io->olens == nullptr && io->row_capacity > 0;
    if ((!io->olens && !devlay) || !io->workspace || (!io->after && !io->after_packed))
        return fail(h, FS2_ERR_ARG, "fs2_decode: olens (or row_capacity) / workspace / after (or after_packed) must be given");
    const fs2_config& c = h->cfg;
    if (devlay) {
        if (!io->status) return fail(h, FS2_ERR_ARG, "fs2_decode: the device-driven layout needs a status buffer");
        if (io->row_capacity > INT32_MAX - 256 || io->Lmax <= 0) return fail(h, FS2_ERR_ARG, "row_capacity %lld / Lmax %d", (long long)io->row_capacity, io->Lmax);
    } else {
        int mx = 0;
        for (int i = 0; i < b.B; ++i) {
            if (io->olens[i] == -1) return fail(h, FS2_ERR_ARG, "utterance %d holds a phoneme id outside [0, %d) (fs2_encode marks it with the frame count -1)", i, h->cfg.idim);
            if (io->olens[i] <= 0) return fail(h, FS2_ERR_ARG, "olens[%d]=%lld", i, (long long)io->olens[i]);
            mx = std::max(mx, (int)io->olens[i]);
        }
        if (io->Lmax < mx) return fail(h, FS2_ERR_ARG, "Lmax %d < longest utterance %d", io->Lmax, mx);
        if (mx > h->dec.pe_rows) return fail(h, FS2_ERR_ARG, "utterance of %d frames exceeds the positional table (%d rows): extend `pe` and reload", mx, h->dec.pe_rows);
    }
    DeviceGuard g(c.device);
    HIP_TRY(h, g.err);
    hipStream_t s = (hipStream_t)stream;
    const int prec = base_precision(b.precision), ffn_terms = ffn_f16_terms(b.precision);
    HostLayout L;
    if (devlay) capacity_layout(b, io->row_capacity, io->Lmax, L);
    else frame_layout(b, io->olens, io->masked, L, h->gap);
    FrameBufs f; bool ok;
    carve_frames(c, L, io->workspace, io->workspace_bytes, &f, &ok);
    if (!ok) return fail(h, FS2_ERR_WORKSPACE, "fs2_decode: workspace too small");
    h->kp = f.kp; h->kp_cap = f.kp_cap;
    DevLayout dl;
    if (devlay) {
        if ((rc = device_layout(h, s, L, f.meta, dl, h->o32, b.compat_padded, io->masked, io->Lmax, h->dec.pe_rows, io->status))) return rc;
    } else if ((rc = upload_layout(h, s, L, f.meta, dl))) return rc;
    const int R = L.R;
    // Kernel variants with different summation orders (LayerNorm fused into the row-complete GEMM or not) are chosen from a row
    // count.  The exact one is unknown to the host in the device-driven mode, and the capacity differs from it, so both modes use
    // the same ESTIMATE instead (regime_rows_of): 8 frames per phoneme plus the per-utterance alignment rows -- of this batch, or of
    // the larger batch the caller says this one is a shard of (fs2_batch.regime_*).
    const int regime_rows = regime_rows_of(b, /*token_level=*/false);
    h->cur_regime = regime_rows; h->cur_status = devlay ? io->status : nullptr;
    // bf16 modes: the length-regulator output (and later its sum with the pitch / energy embeddings) is also written as planes,
    // in x1p (free until the decoder stack's first LayerNorm): the A operand of both variance predictors and of the decoder input layer
    void* hfr_planes = (prec != FS2_PREC_FP32 && c.adim % 32 == 0) ? f.sb.x1p : nullptr;
    {   // length regulator
        Scope sc(h, s, "lr.expand", 0, 4.0 * R * c.adim * 2);
        hipLaunchKernelGGL(lr_expand, dim3((R + 3) / 4), dim3(256), 0, s, h->enc_final, c.adim, h->dtok.start, h->dtok.vlen, h->cum, b.Tmax,
                           dl.row_pos, dl.row_seq, 0, dl.vlen, R, f.hfr, f.lri, hfr_planes);
        HIP_TRY(h, hipGetLastError());
    }
    const bool need_e = (io->es == nullptr) || io->e_out, need_p = (io->ps == nullptr) || io->p_out;
    if (need_e && need_p && prec != FS2_PREC_FP32 && h->var2.ok && hfr_planes && opts().fuse_var) {
        if ((rc = run_predictors_fused(h, s, h->var2, f.hfr, c.adim, R, dl.row_pos, dl.dims, hfr_planes, f.sb.xps, f.vp, f.vs, f.e_rows, f.p_rows, prec))) return rc;
    } else {
        if (need_e && (rc = run_predictor(h, s, "energy", h->energy, f.hfr, c.adim, R, dl.row_pos, f.t0, f.t1, f.e_rows, prec, hfr_planes, f.sb.xps, dl.dims))) return rc;
        if (need_p && (rc = run_predictor(h, s, "pitch", h->pitch, f.hfr, c.adim, R, dl.row_pos, f.t0, f.t1, f.p_rows, prec, hfr_planes, f.sb.xps, dl.dims))) return rc;
    }
    {
        Scope sc(h, s, "var.embed", 0, 4.0 * R * c.adim * 4);
        hipLaunchKernelGGL(bucket_embed, dim3((R + 3) / 4), dim3(256), 0, s, f.hfr, c.adim, dl.row_pos, dl.row_seq, R, io->es, io->es_stride,
                           io->ps, io->ps_stride, f.e_rows, f.p_rows, h->ebins, h->pbins, c.n_bins - 1, h->Te, h->Tp, f.qe, f.qp, hfr_planes);
        HIP_TRY(h, hipGetLastError());
    }
    const bool dec_pl = prec != FS2_PREC_FP32;
    // Planes-only residual stream (round 6): where the decoder's LayerNorm-fused launches run on gemm_row4_bf16 they write planes and nothing else and
    // read their residual from planes (gemm_row4.h: RES) -- the fp32 rows x0 / x1 are not written at all (0.9 GB per c3 step, 11 GB per c4 step of HBM
    // writes).  Not in the fp16 two- / one-term modes (their LN1 output is a fp16 hi + lo pair, a format the residual reader does not have).
    const bool po = dec_pl && planes_only_regime(c, h->dec, prec, regime_rows) && (ffn_terms == 0 || ffn_terms == kFfnMx || ffn_terms == kFfnMx4);
    if (c.decoder_input_layer) {   // decoder input layer: Linear -> LN -> ReLU -> + alpha * pe   (reference encoder.py:118-125)
        GemmArgs a = gemm_args(h->dec_in, f.hfr, c.adim, R, dl.row_pos, po ? nullptr : f.sb.x0, c.ddim);
        a.Rp = dl.dims; a.regime_rows = regime_rows;
        a.ln_g = h->dec_in_lng; a.ln_b = h->dec_in_lnb; a.ln_eps = 1e-5f; a.act_post = 1;
        a.pe = h->dec.pe; a.pe_ld = c.ddim; a.pe_alpha = h->dec.alpha; a.x_scale = c.use_scaled_pos_enc ? 1.f : sqrtf((float)c.ddim);
        a.Xp = hfr_planes; a.xp_scratch = f.sb.xps;
        if (dec_pl) { a.Yp = f.sb.x0p; a.yp_chunks = c.ddim / 32; }
        if ((rc = launch_gemm(h, s, "dec.in", a, prec))) return rc;
    } else {                       // TorchScript twin: the decoder input is just x (* sqrt(d)) + alpha * pe   (encoder.py:138-141)
        Scope sc(h, s, "dec.in.pe", 0.0, 8.0 * R * c.ddim);
        HIP_TRY(h, hipMemcpyAsync(f.sb.x0, f.hfr, (size_t)R * c.ddim * sizeof(float), hipMemcpyDeviceToDevice, s));
        GemmArgs a;
        memset(&a, 0, sizeof a);
        a.N = c.ddim; a.R = R; a.row_pos = dl.row_pos; a.Y = f.sb.x0; a.ldy = c.ddim;
        a.pe = h->dec.pe; a.pe_ld = c.ddim; a.pe_alpha = h->dec.alpha; a.x_scale = c.use_scaled_pos_enc ? 1.f : sqrtf((float)c.ddim);
        if (dec_pl) { a.Yp = f.sb.x0p; a.yp_chunks = c.ddim / 32; }
        hipLaunchKernelGGL(ln_rows, dim3((R + 3) / 4), dim3(256), 0, s, a);
        HIP_TRY(h, hipGetLastError());
    }
    const int mask_q = (b.compat_padded && io->masked) ? 1 : 0;
    if ((rc = run_stack(h, s, "dec", h->dec, c.ddim, c.aheads, R, L, dl, mask_q, f.sb, prec, /*x0p_ready=*/dec_pl, /*allow_splitk=*/false, regime_rows, ffn_terms, po))) return rc;
    if (po && io->dec_out) {      // the API hands the decoder output out as fp32 rows: rebuilt from the planes (hi + lo) only when asked for
        Scope sc(h, s, "dec.out.rows", 0.0, 8.0 * R * c.ddim);
        const int64_t n = (int64_t)R * (c.ddim / 4);
        hipLaunchKernelGGL(planes_to_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, f.sb.x0p, c.ddim / 32, R, c.ddim, f.sb.x0, c.ddim);
        HIP_TRY(h, hipGetLastError());
    }
    // reduction_factor r (reference fastspeech.py:153,228-230): feat_out emits r mel frames per decoder frame; its [R, odim r] output
    // IS the [R r, odim] row image the Postnet runs on (decoder row i -> rows i r .. i r + r - 1, gap rows stay zero rows), so only the
    // row metadata is rebuilt at the finer rate.  r = 1 uses the decoder's own metadata and buffers.
    const int rf = std::max(c.reduction_factor, 1);
    const int R2 = R * rf;
    DevLayout dl2 = dl;
    if (rf > 1) {
        int* m = f.meta2;
        dl2.start = m; dl2.len = m + b.B; dl2.vlen = m + 2 * b.B; dl2.klen = dl2.len;
        int* rest = reinterpret_cast<int*>(align_up(reinterpret_cast<size_t>(m + 3 * b.B), 16));
        dl2.dims = dl.dims ? rest : nullptr; rest += 16;
        const int Rpad2 = L.Rpad * rf;
        dl2.row_pos = rest; dl2.row_seq = rest + Rpad2;
        hipLaunchKernelGGL(scale_layout, dim3((b.B + 255) / 256), dim3(256), 0, s, dl.start, dl.len, dl.vlen, dl.dims, b.B, rf, dl2.start, dl2.len, dl2.vlen, dl2.dims);
        hipLaunchKernelGGL(build_row_meta, dim3((Rpad2 + 255) / 256), dim3(256), 0, s, dl2.start, dl2.len, b.B, Rpad2, dl2.row_pos, dl2.row_seq);
        HIP_TRY(h, hipGetLastError());
    }
    // planes hand-off through the tail: decoder -> feat_out (fp32 mel + planes of it) -> Postnet convs ping-pong x0p / x1p
    const bool post_pl = dec_pl && c.postnet_layers > 1 && rf == 1;     // (r > 1: feat_out's planes would be in the decoder-row layout)
    {
        GemmArgs a = gemm_args(h->feat, f.sb.x0, c.ddim, R, dl.row_pos, f.before, c.odim * rf);
        a.Rp = dl.dims;
        if (dec_pl) a.Xp = f.sb.x0p;
        if (post_pl) { a.Yp = f.sb.xps; a.yp_chunks = round_up(c.odim, 32) / 32; }
        if ((rc = launch_gemm(h, s, "feat_out", a, prec))) return rc;
    }
    const float* mel_after = f.before;
    if (c.postnet_layers > 0) {
        float* pa = rf > 1 ? f.pa2 : f.sb.hid;
        float* pb = rf > 1 ? f.pb2 : f.sb.hid + (size_t)L.Rpad * c.postnet_chans;
        const float* in = f.before; int ld = c.odim;
        for (int l = 0; l < c.postnet_layers; ++l) {
            const bool last = (l == c.postnet_layers - 1);
            float* out = last ? f.after : ((l & 1) ? pb : pa);
            GemmArgs a = gemm_args(h->post[l], in, ld, R2, dl2.row_pos, out, h->post[l].N);
            a.Rp = dl2.dims;
            if (!last) a.act_post = 2; else { a.resid = f.before; a.ldr = c.odim; }
            if (post_pl) {
                a.Xp = (l == 0) ? f.sb.xps : ((l & 1) ? f.sb.x0p : f.sb.x1p);
                if (!last) { a.Y = nullptr; a.Yp = (l & 1) ? f.sb.x1p : f.sb.x0p; a.yp_chunks = round_up(h->post[l].N, 32) / 32; }
                // round 6: the 512 -> 512 layers in the mx arithmetic of the mixed modes (fp16 main term + e4m3 cross terms, 2 MFMA-equivalents instead of 3; simulated:
                // mel +6e-6, tools/arith_sim_postnet.py): a layer whose image exists takes mx planes, so the layer in front of it writes them (tanh output: static scale 2^8)
                const bool post_mx = (ffn_terms == kFfnMx || ffn_terms == kFfnMx4) && opts().post_mx;
                if (post_mx && !last && h->post[l + 1].wm) { a.yp_f16 = 2; a.yp_scale = exp2f((float)kPostKa); }
                if (post_mx && l > 0 && h->post[l].wm) { a.mx = 1; a.Wb = h->post[l].wm; a.mx_scale = scale_byte4(127 - kPostKa - 11); a.mx_scale_b = scale_byte4(127 - h->post[l].kw); }
            } else {
                a.xp_scratch = rf > 1 ? f.xps2 : f.sb.xps;
            }
            char nm[32]; snprintf(nm, sizeof nm, "postnet.%d", l);
            if ((rc = launch_gemm(h, s, nm, a, prec))) return rc;
            in = out; ld = h->post[l].N;
        }
        mel_after = f.after;
    }
    {
        Scope sc(h, s, "unpack", 0, 4.0 * R * c.odim * 4);
        const int* lim_len = dl.len;    // every stored row (pads carry real values in compat mode)
        const int* lim_msk = (b.compat_padded && !io->masked) ? dl.len : dl.vlen;
        // device-driven layout: the kernels that write the mel outputs look at the overflow flags themselves and write NaN when a
        // capacity was too small (nobody can mistake the outputs of such a call for silence); poison_on_overflow covers the rest
        const int* ovf = devlay ? dl.dims + 2 : nullptr;
        bool after_done = false, before_done = false, packed_done = false;
        // (mel outputs hold Lmax * reduction_factor frames per utterance)
        if (io->after && (rc = unpack<float>(h, s, mel_after, c.odim, dl2.start, dl2.len, b.B, io->Lmax * rf, io->after, 0.f, ovf, &after_done))) return rc;
        if (io->before && (rc = unpack<float>(h, s, f.before, c.odim, dl2.start, dl2.len, b.B, io->Lmax * rf, io->before, 0.f, ovf, &before_done))) return rc;
        if (io->e_out && (rc = unpack<float>(h, s, f.e_rows, 1, dl.start, lim_msk, b.B, io->Lmax, io->e_out, 0.f))) return rc;
        if (io->p_out && (rc = unpack<float>(h, s, f.p_rows, 1, dl.start, lim_msk, b.B, io->Lmax, io->p_out, 0.f))) return rc;
        if (io->qe && (rc = unpack<int>(h, s, f.qe, 1, dl.start, lim_len, b.B, io->Lmax, io->qe, -1))) return rc;
        if (io->qp && (rc = unpack<int>(h, s, f.qp, 1, dl.start, lim_len, b.B, io->Lmax, io->qp, -1))) return rc;
        if (io->lr_index && (rc = unpack<int>(h, s, f.lri, 1, dl.start, dl.vlen, b.B, io->Lmax, io->lr_index, -1))) return rc;
        if (io->dec_out && (rc = unpack<float>(h, s, f.sb.x0, c.ddim, dl.start, lim_len, b.B, io->Lmax, io->dec_out, 0.f))) return rc;
        if (io->after_packed) {
            if (c.odim % 4) return fail(h, FS2_ERR_UNSUPPORTED, "after_packed needs odim %% 4 == 0");
            const int* dcum;
            if (devlay) {
                dcum = dl.pcum;      // built by frame_layout_dev; the caller's buffer holds row_capacity rows (>= the frames)
            } else {
                std::vector<int> cum(b.B);
                int run = 0;
                for (int i = 0; i < b.B; ++i) { cum[i] = run; run += (int)io->olens[i]; }
                int* up = f.qe;    // the bucket-index rows are already unpacked: reuse their storage for the offsets
                HIP_TRY(h, hipMemcpyAsync(up, cum.data(), cum.size() * sizeof(int), hipMemcpyHostToDevice, s));
                dcum = up;
            }
            // (r > 1: the rows are mel frames -- the Postnet's layout dl2, r rows per decoder row -- and the offsets r times the decoder-frame sums)
            const int64_t n = (int64_t)R2 * (c.odim / 4);
            hipLaunchKernelGGL(pack_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mel_after, c.odim, dl2.row_pos, dl2.row_seq, dl2.vlen, dcum, R2, io->after_packed,
                               (devlay && R == io->row_capacity) ? ovf : (const int*)nullptr, rf);
            HIP_TRY(h, hipGetLastError());
            packed_done = devlay && R == io->row_capacity;
        }
        const bool need_poison = devlay && ((io->after && !after_done) || (io->before && !before_done) || (io->after_packed && !packed_done));
        if (need_poison) {
            hipLaunchKernelGGL(poison_on_overflow, dim3(256), dim3(256), 0, s, dl.dims, io->after, (io->after && !after_done) ? (int64_t)b.B * io->Lmax * rf * c.odim : (int64_t)0,
                               io->after_packed, (io->after_packed && !packed_done) ? io->row_capacity * rf * c.odim : (int64_t)0, io->before,
                               (io->before && !before_done) ? (int64_t)b.B * io->Lmax * rf * c.odim : (int64_t)0);
            HIP_TRY(h, hipGetLastError());
        }
    }
    return FS2_OK;
}

// ---------------------------------------------------------------------------------- single operators
#define OP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_); return FS2_ERR_HIP; } } while (0)

// temporary device memory of an operator entry point: released on every exit path (after the stream drained)
struct DevTmp {
    hipStream_t s;
    std::vector<void*> ptrs;
    explicit DevTmp(hipStream_t s_) : s(s_) {}
    ~DevTmp() { if (!ptrs.empty()) hipStreamSynchronize(s); for (void* p : ptrs) hipFree(p); }
    hipError_t alloc(void** out, size_t bytes) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 16));
        if (e == hipSuccess) { ptrs.push_back(p); *out = p; }
        return e;
    }
};

int fs2_op_conv_gemm(void* stream, const fs2_op_gemm_args* o) {
    if (!o) return fail(nullptr, FS2_ERR_ARG, "fs2_op_conv_gemm: null argument");
    ABI_CHECK(nullptr, "fs2_op_conv_gemm", o, fs2_op_gemm_args);
    if (!o->x || !o->w || o->R <= 0) return fail(nullptr, FS2_ERR_ARG, "fs2_op_conv_gemm: bad arguments");
    if (o->precision < FS2_PREC_FP32 || o->precision > FS2_PREC_MIX_MX4) return fail(nullptr, FS2_ERR_ARG, "unknown precision %d", o->precision);
    const bool mx = o->precision == FS2_PREC_MIX_MX || o->precision == FS2_PREC_MIX_MX4;   // THIS operator in the fp16 + block-scaled-fp8 arithmetic (convolutions, C % 128 == 0; the fp4 form exists inside the model only: its operand comes with row scales from a LayerNorm epilogue)
    const int f16t = mx ? 1 : ffn_f16_terms(o->precision);      // mixed modes: THIS operator on fp16 operands with 2 / 1 MFMAs per fragment pair
    hipStream_t s = (hipStream_t)stream;
    Gemm g; g.N = o->N; g.C = o->C; g.ktaps = o->ktaps; g.Cpad = round_up(o->C, kBK);
    const int Npad = round_up(o->N, 128);
    const int nchunks = g.Cpad / 32;
    const size_t wn = (size_t)Npad * o->ktaps * g.Cpad;
    DevTmp tmp(s);
    OP_TRY(tmp.alloc((void**)&g.w, wn * sizeof(float)));
    OP_TRY(tmp.alloc((void**)&g.wb, wn * 4));
    float* scratch = nullptr;
    OP_TRY(tmp.alloc((void**)&scratch, (size_t)o->R * o->N * sizeof(float)));
    void* xps = nullptr;          // planes of x for the bf16 modes (gemm_planes.h)
    if (o->precision != FS2_PREC_FP32) OP_TRY(tmp.alloc((void**)&xps, (size_t)o->R * g.Cpad * sizeof(float)));
    int* rp = nullptr;
    hipMemsetAsync(g.w, 0, wn * sizeof(float), s);
    hipMemsetAsync(g.wb, 0, wn * 4, s);
    const int64_t total = (int64_t)o->N * o->ktaps * g.Cpad;
    hipLaunchKernelGGL(repack_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, o->w, o->N, o->C, o->ktaps, o->N, g.Cpad,
                       (const float*)nullptr, (const float*)nullptr, 0.f, g.w);
    const int64_t tb = (int64_t)o->N * o->ktaps * nchunks * 32;
    hipLaunchKernelGGL(repack_weight_bf16, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, s, o->w, o->N, o->C, o->ktaps, o->N, nchunks,
                       (const float*)nullptr, (const float*)nullptr, 0.f, reinterpret_cast<__bf16*>(g.wb), f16t ? 1 : 0);
    g.bias = const_cast<float*>(o->bias);
    GemmArgs a = gemm_args(g, o->x, o->C, o->R, nullptr, o->y, o->N);
    a.scratch = scratch;
    if (o->row_valid) {   // row_valid (0/1) -> row_pos (-1 / 0)
        OP_TRY(tmp.alloc((void**)&rp, (size_t)o->R * sizeof(int)));
        std::vector<int> hv(o->R);
        OP_TRY(hipMemcpyAsync(hv.data(), o->row_valid, (size_t)o->R * sizeof(int), hipMemcpyDeviceToHost, s));
        OP_TRY(hipStreamSynchronize(s));
        for (auto& v : hv) v = v ? 0 : -1;
        OP_TRY(hipMemcpyAsync(rp, hv.data(), (size_t)o->R * sizeof(int), hipMemcpyHostToDevice, s));
        a.row_pos = rp;
    }
    a.resid = o->resid; a.ldr = o->N; a.relu_pre = o->relu_pre; a.ln_g = o->ln_gamma; a.ln_b = o->ln_beta; a.ln_eps = o->ln_eps;
    a.act_post = o->act_post; a.dot_w = o->dot_w; a.dot_b = o->dot_b; a.dot_out = o->dot_out;
    a.xp_scratch = xps;
    a.f16_terms = mx ? 0 : f16t;
    if (mx) {
        if (o->ktaps < 3 || Npad != o->N || o->C % 128 != 0) return fail(nullptr, FS2_ERR_UNSUPPORTED, "the mx arithmetic needs a convolution with C %% 128 == 0 and N %% 128 == 0");
        unsigned* scr = nullptr;
        OP_TRY(tmp.alloc((void**)&scr, 16));
        const int ka = fp8_scale_exponent(device_absmax(s, o->x, (int64_t)o->R * o->C, scr));
        const int kw = fp8_scale_exponent(device_absmax(s, o->w, (int64_t)o->N * o->C * o->ktaps, scr));
        void* wm = nullptr;
        const size_t bytes = mx_image_bytes(Npad, o->C, o->ktaps);
        OP_TRY(tmp.alloc(&wm, bytes));
        hipLaunchKernelGGL(repack_weight_mx, dim3((unsigned)((bytes / 2 + 255) / 256)), dim3(256), 0, s, o->w, o->N, o->C, o->ktaps, Npad, kw, reinterpret_cast<unsigned short*>(wm));
        a.mx = 1; a.Wb = wm; a.yp_scale = exp2f((float)ka); a.mx_scale = scale_byte4(127 - ka - 11); a.mx_scale_b = scale_byte4(127 - kw);
    }
    return launch_gemm(nullptr, s, "op.conv_gemm", a, base_precision(o->precision));      // (tmp drains the stream and frees)
}

int fs2_op_attention(void* stream, const float* qkv, float* ctx, int32_t D, int32_t heads, int32_t B, const int32_t* seq_start,
                     const int32_t* seq_len, const int32_t* seq_klen, int32_t mask_q, int32_t precision) {
    if (!qkv || !ctx || B <= 0) return fail(nullptr, FS2_ERR_ARG, "fs2_op_attention: bad arguments");
    if (precision < FS2_PREC_FP32 || precision > FS2_PREC_BF16) return fail(nullptr, FS2_ERR_ARG, "unknown precision %d", precision);
    hipStream_t s = (hipStream_t)stream;
    std::vector<int> host;
    std::vector<int2> work;
    int R = 0;
    for (int b = 0; b < B; ++b) {
        for (int q = 0; q * kAttBlk < seq_len[b]; ++q) work.push_back(make_int2(b, q));
        R = std::max(R, seq_start[b] + seq_len[b]);
        if (precision != FS2_PREC_FP32 && seq_start[b] % kAttAlign) return fail(nullptr, FS2_ERR_ARG, "bf16 attention needs sequence starts aligned to %d rows", kAttAlign);
    }
    host.insert(host.end(), seq_start, seq_start + B);
    host.insert(host.end(), seq_len, seq_len + B);
    host.insert(host.end(), seq_klen, seq_klen + B);
    for (auto& w : work) { host.push_back(w.x); host.push_back(w.y); }
    DevTmp tmp(s);
    int* dev = nullptr;
    OP_TRY(tmp.alloc((void**)&dev, host.size() * sizeof(int) + 16));
    OP_TRY(hipMemcpyAsync(dev, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, s));
    DevLayout dl;
    dl.start = dev; dl.len = dev + B; dl.klen = dev + 2 * B; dl.work = reinterpret_cast<int2*>(dev + 3 * B);
    int rc;
    if (precision == FS2_PREC_FP32) {
        rc = launch_attention(nullptr, s, "op.attention", qkv, ctx, D, heads, dl, (int)work.size(), mask_q, 0.0);
    } else {
        const int Rvt = round_up(R + kTailRows, 128);      // (8-key V^T vectors that straddle the last key stay inside the planes)
        __bf16* planes = nullptr;
        OP_TRY(tmp.alloc((void**)&planes, (size_t)Rvt * D * 6 * sizeof(__bf16)));
        __bf16 *qkh = planes, *qkl = planes + (size_t)Rvt * 2 * D, *vth = qkl + (size_t)Rvt * 2 * D, *vtl = vth + (size_t)Rvt * D;
        if (opts().op_att_planes > 0 && D % 32 == 0) {
            // the model's output form: the kernels write the context as planes only (attn_w32: the LDS-staged epilogue), converted here
            void* ctxp = nullptr;
            OP_TRY(tmp.alloc(&ctxp, (size_t)Rvt * D * 4));
            OP_TRY(hipMemsetAsync(ctxp, 0, (size_t)Rvt * D * 4, s));
            rc = launch_attention_b16(nullptr, s, "op.attention", qkv, nullptr, D, heads, R, Rvt, dl, (int)work.size(), mask_q, 0.0, precision, qkh, qkl, vth, vtl, ctxp);
            if (rc == FS2_OK) {
                const int64_t n = (int64_t)R * (D / 4);
                hipLaunchKernelGGL(planes_to_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ctxp, D / 32, R, D, ctx, D);
                OP_TRY(hipGetLastError());
            }
        } else {
            rc = launch_attention_b16(nullptr, s, "op.attention", qkv, ctx, D, heads, R, Rvt, dl, (int)work.size(), mask_q, 0.0, precision, qkh, qkl, vth, vtl);
        }
    }
    return rc;      // (tmp drains the stream and frees)
}

int fs2_op_length_regulate(void* stream, const float* hs, const int64_t* ds, const int64_t* ilens_host, int32_t B, int32_t Tmax,
                           int32_t D, int32_t Lmax, float alpha, float* out, int32_t* index, int64_t* olens) {
    if (!hs || !ds || !ilens_host || !out || B <= 0 || D % 4 || !(alpha > 0.f)) return fail(nullptr, FS2_ERR_ARG, "fs2_op_length_regulate: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    std::vector<int> host(2 * B);
    for (int b = 0; b < B; ++b) { host[b] = b * Tmax; host[B + b] = (int)ilens_host[b]; }
    DevTmp tmp(s);
    int* dev = nullptr;
    OP_TRY(tmp.alloc((void**)&dev, ((size_t)3 * B + (size_t)B * Tmax) * sizeof(int)));
    OP_TRY(hipMemcpyAsync(dev, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, s));
    int *tok_start = dev, *ilen = dev + B, *o32 = dev + 2 * B, *cum = dev + 3 * B;
    hipLaunchKernelGGL(dur_scan, dim3(B), dim3(256), 0, s, ds, Tmax, ilen, cum, olens, o32, alpha);
    const int R = B * Lmax;
    hipLaunchKernelGGL(lr_expand, dim3((R + 3) / 4), dim3(256), 0, s, hs, D, tok_start, ilen, cum, Tmax, (const int*)nullptr,
                       (const int*)nullptr, Lmax, o32, R, out, index);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, FS2_ERR_HIP, "length_regulate: %s", hipGetErrorString(e));
    return FS2_OK;
}

int fs2_op_unpack_rows(void* stream, const float* src, int32_t W, int32_t B, const int32_t* starts, const int32_t* lens, int32_t Lout,
                       float* dst) {
    if (!src || !dst || !starts || !lens || B <= 0 || W <= 0 || Lout <= 0) return fail(nullptr, FS2_ERR_ARG, "fs2_op_unpack_rows: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    // per-process scratch for the two small index arrays (grown on demand, never freed: a few KB)
    static int* dev = nullptr; static int cap = 0;
    if (2 * B > cap) {
        if (dev) hipFree(dev);
        cap = std::max(2 * B, 4096);
        OP_TRY(hipMalloc((void**)&dev, (size_t)cap * sizeof(int)));
    }
    std::vector<int> host(2 * B);
    for (int i = 0; i < B; ++i) { host[i] = starts[i]; host[B + i] = lens[i]; }
    OP_TRY(hipMemcpyAsync(dev, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, s));
    const int64_t total = (int64_t)B * Lout * W;
    if (W % 4 == 0 && (reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) % 16 == 0)
        hipLaunchKernelGGL(unpack_rows4, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, src, W / 4, dev, dev + B, B, Lout, dst);
    else
        hipLaunchKernelGGL(unpack_rows<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, W, dev, dev + B, B, Lout, dst, 0.f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, FS2_ERR_HIP, "unpack_rows: %s", hipGetErrorString(e));
    return FS2_OK;
}

int fs2_op_unpack_rows_dev(void* stream, const float* src, int32_t W, int32_t B, const int32_t* starts_dev, const int32_t* lens_dev,
                           int32_t Lout, float* dst) {
    if (!src || !dst || !starts_dev || !lens_dev || B <= 0 || W <= 0 || Lout <= 0) return fail(nullptr, FS2_ERR_ARG, "fs2_op_unpack_rows_dev: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)B * Lout * W;
    if (W % 4 == 0 && (reinterpret_cast<size_t>(src) | reinterpret_cast<size_t>(dst)) % 16 == 0)
        hipLaunchKernelGGL(unpack_rows4, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, s, src, W / 4, starts_dev, lens_dev, B, Lout, dst);
    else
        hipLaunchKernelGGL(unpack_rows<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, W, starts_dev, lens_dev, B, Lout, dst, 0.f);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, FS2_ERR_HIP, "unpack_rows: %s", hipGetErrorString(e));
    return FS2_OK;
}

int fs2_op_transpose(void* stream, const float* src, int64_t N, int32_t W, float* dst) {
    if (!src || !dst || N < 0 || W <= 0) return fail(nullptr, FS2_ERR_ARG, "fs2_op_transpose: bad arguments");
    if (N == 0) return FS2_OK;
    hipLaunchKernelGGL(transpose_rows, dim3((unsigned)((N + 31) / 32), (W + 31) / 32), dim3(256), 0, (hipStream_t)stream, src, N, W, dst);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, FS2_ERR_HIP, "transpose: %s", hipGetErrorString(e));
    return FS2_OK;
}

int fs2_op_duration(void* stream, const float* d_log, int64_t n, int64_t* d) {
    if (!d_log || !d || n < 0) return fail(nullptr, FS2_ERR_ARG, "fs2_op_duration: bad arguments");
    if (n == 0) return FS2_OK;
    hipLaunchKernelGGL(duration_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_log, n, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, FS2_ERR_HIP, "duration: %s", hipGetErrorString(e));
    return FS2_OK;
}

int fs2_set_option(const char* name, int32_t value) {
    if (!name) return fail(nullptr, FS2_ERR_ARG, "fs2_set_option: null name");
    Options& o = opts();
    const std::string n = name;
    if (n == "FS2_BM") o.bm = value;
    else if (n == "FS2_ROW8") o.row8 = value;
    else if (n == "FS2_QKV8") o.qkv8 = value;
    else if (n == "FS2_NOSPLITK") o.nosplitk = value > 0;
    else if (n == "FS2_F32_ROWS") o.f32_rows = value > 0;
    else if (n == "FS2_MT8") o.mt8 = value;
    else if (n == "FS2_QKV_SPLIT") o.qkv_split = value;
    else if (n == "FS2_OP_ATT_PLANES") o.op_att_planes = value;
    else if (n == "FS2_FUSE_VAR") o.fuse_var = value != 0;
    else if (n == "FS2_BAL") o.bal = value < 0 ? 0 : value;
    else if (n == "FS2_ATTN_W32") o.w32 = value;
    else if (n == "FS2_ROW4") o.row4 = value;
    else if (n == "FS2_MT4") o.mt4 = value;
    else if (n == "FS2_FFN2_MX") o.ffn2_mx = value != 0;
    else if (n == "FS2_QKV4") o.qkv4 = value;
    else if (n == "FS2_POST_MX") o.post_mx = value != 0;
    else return fail(nullptr, FS2_ERR_ARG, "fs2_set_option: unknown option %s", name);
    return FS2_OK;
}

int fs2_get_option(const char* name, int32_t* value) {
    if (!name || !value) return fail(nullptr, FS2_ERR_ARG, "fs2_get_option: null argument");
    const Options& o = opts();
    const std::string n = name;
    if (n == "FS2_AUDIT_CLEAN") *value = audit_clean() ? 1 : 0;
    else if (n == "attn_w32_active") *value = o.w32 != 0;
    else if (n == "row4_active") *value = o.row4 != 0;
    else if (n == "qkv4_active") *value = o.qkv4 != 0 && o.row4 != 0;
    else if (n == "FS2_BM") *value = o.bm;
    else if (n == "FS2_ROW8") *value = o.row8;
    else if (n == "FS2_QKV8") *value = o.qkv8;
    else if (n == "FS2_NOSPLITK") *value = o.nosplitk;
    else if (n == "FS2_F32_ROWS") *value = o.f32_rows;
    else if (n == "FS2_MT8") *value = o.mt8;
    else if (n == "FS2_QKV_SPLIT") *value = o.qkv_split;
    else if (n == "FS2_OP_ATT_PLANES") *value = o.op_att_planes;
    else if (n == "FS2_FUSE_VAR") *value = o.fuse_var;
    else if (n == "FS2_BAL") *value = o.bal;
    else if (n == "FS2_ATTN_W32") *value = o.w32;
    else if (n == "FS2_ROW4") *value = o.row4;
    else if (n == "FS2_MT4") *value = o.mt4;
    else if (n == "FS2_FFN2_MX") *value = o.ffn2_mx;
    else if (n == "FS2_QKV4") *value = o.qkv4;
    else if (n == "FS2_POST_MX") *value = o.post_mx;
    else return fail(nullptr, FS2_ERR_ARG, "fs2_get_option: unknown option %s", name);
    return FS2_OK;
}

int fs2_get_counter(fs2_handle* h, void* stream, const char* name, int64_t* value, int32_t reset) {
    if (!h || !name || !value) return fail(h, FS2_ERR_ARG, "fs2_get_counter: null argument");
    if (std::string(name) != "attn_slow_path_waves") return fail(h, FS2_ERR_ARG, "fs2_get_counter: unknown counter %s", name);
    DeviceGuard g(h->cfg.device);
    HIP_TRY(h, g.err);
    hipStream_t s = (hipStream_t)stream;
    int v = 0;
    HIP_TRY(h, hipMemcpyAsync(&v, h->counters, sizeof(int), hipMemcpyDeviceToHost, s));
    if (reset) HIP_TRY(h, hipMemsetAsync(h->counters, 0, sizeof(int), s));
    HIP_TRY(h, hipStreamSynchronize(s));
    *value = v;
    return FS2_OK;
}

int fs2_op_bucketize(void* stream, const float* x, int64_t n, const float* bins, int32_t nb, int32_t* idx) {
    if (!x || !bins || !idx || n < 0) return fail(nullptr, FS2_ERR_ARG, "fs2_op_bucketize: bad arguments");
    if (n == 0) return FS2_OK;
    hipLaunchKernelGGL(bucketize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, bins, nb, idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, FS2_ERR_HIP, "bucketize: %s", hipGetErrorString(e));
    return FS2_OK;
}

}  // extern "C"
