// libfs2_torch.so -- the dispatcher op behind the TorchScript twin (reference utils/fastspeech2_script.py:201-219, export_torchscript.py:46-58),
// registered from C++ so that an exported archive runs in ANY process that has libtorch and this library -- no Python, no import of the
// fastspeech2_amd package:
//
//     torch.ops.load_library(".../fastspeech2_amd/libfs2_torch.so");  m = torch.jit.load("fs2_twin.pt");  mel = m(ids)
//     (C++: dlopen / link the library, torch::jit::load(...).forward({ids}))
//
//     fs2::twin_inference(Tensor x, Tensor flat_weights, str config_json) -> Tensor
//
// x: [T] int64 phoneme ids; flat_weights: every float32 tensor of the twin's state dict, concatenated; config_json: the hyper-parameters
// (the fs2_config fields) plus the manifest {"tensors": [[name, [shape...]], ...]} that says how to cut the flat buffer, written by
// fastspeech2_amd/fastspeech2_script.py at export time.  The op builds a libfs2_hip handle per (weight buffer, config) -- an LRU of FS2_TWIN_CACHE entries (default 2) --
// and runs fs2_encode -> frame count read-back -> fs2_decode on torch's current HIP stream with torch-allocated workspaces, exactly what
// FeedForwardTransformer.inference() does through ctypes.  Inputs on another device are moved to the weights' device (the reference traces
// with a CPU example input, export_torchscript.py:53-56); weights on the CPU are refused: there is no CPU path.
#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fs2.h"

namespace {

// ---------------------------------------------------------------- a JSON reader for what fastspeech2_script.py writes
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    double num = 0; bool b = false; std::string str; std::vector<Json> arr; std::map<std::string, Json> obj;
    const Json* find(const std::string& k) const { auto it = obj.find(k); return it == obj.end() ? nullptr : &it->second; }
};
struct JsonParser {
    const std::string& s; size_t i = 0;
    explicit JsonParser(const std::string& s_) : s(s_) {}
    // every read of s[i] goes through at(): a truncated or malformed document is a TORCH_CHECK error, never a read past the buffer
    char at() const { TORCH_CHECK(i < s.size(), "fs2::twin_inference: truncated config_json (offset ", i, ")"); return s[i]; }
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
    static void put_utf8(std::string& o, unsigned cp) {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) { o.push_back((char)(0xE0 | (cp >> 12))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else { o.push_back((char)(0xF0 | (cp >> 18))); o.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); o.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); o.push_back((char)(0x80 | (cp & 0x3F))); }
    }
    unsigned hex4() {
        TORCH_CHECK(i + 4 <= s.size(), "config_json: truncated \\u escape");
        unsigned v = 0;
        for (int k = 0; k < 4; ++k) {
            const char c = s[i++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else TORCH_CHECK(false, "config_json: bad \\u escape");
        }
        return v;
    }
    Json parse(int depth = 0) {
        TORCH_CHECK(depth < 64, "config_json: nested too deeply");
        ws();
        Json j;
        const char c = at();
        if (c == '{') {
            j.kind = Json::Obj; ++i; ws();
            if (at() == '}') { ++i; return j; }
            for (;;) {
                ws(); Json k = parse(depth + 1); TORCH_CHECK(k.kind == Json::Str, "config_json: object key is not a string");
                ws(); TORCH_CHECK(at() == ':', "config_json: ':' expected"); ++i;
                j.obj[k.str] = parse(depth + 1); ws();
                if (at() == ',') { ++i; continue; }
                TORCH_CHECK(at() == '}', "config_json: '}' expected"); ++i; return j;
            }
        }
        if (c == '[') {
            j.kind = Json::Arr; ++i; ws();
            if (at() == ']') { ++i; return j; }
            for (;;) {
                j.arr.push_back(parse(depth + 1)); ws();
                if (at() == ',') { ++i; continue; }
                TORCH_CHECK(at() == ']', "config_json: ']' expected"); ++i; return j;
            }
        }
        if (c == '"') {
            j.kind = Json::Str; ++i;
            for (;;) {
                const char ch = at();          // (an unterminated string ends in the TORCH_CHECK of at())
                ++i;
                if (ch == '"') return j;
                if (ch != '\\') { j.str.push_back(ch); continue; }
                const char e = at();
                ++i;
                switch (e) {
                    case '"': case '\\': case '/': j.str.push_back(e); break;
                    case 'n': j.str.push_back('\n'); break;
                    case 't': j.str.push_back('\t'); break;
                    case 'r': j.str.push_back('\r'); break;
                    case 'b': j.str.push_back('\b'); break;
                    case 'f': j.str.push_back('\f'); break;
                    case 'u': {      // json.dumps writes every non-ASCII character of an hp value this way
                        unsigned cp = hex4();
                        // a surrogate is only ever half of a pair: a lone one (high without a following \u low, or a low on its own) would
                        // become a three-byte sequence that is not UTF-8 (round-4 advisor finding)
                        TORCH_CHECK(!(cp >= 0xDC00 && cp < 0xE000), "config_json: unpaired low surrogate");
                        if (cp >= 0xD800 && cp < 0xDC00) {
                            TORCH_CHECK(i + 1 < s.size() && s[i] == '\\' && s[i + 1] == 'u', "config_json: unpaired high surrogate");
                            i += 2;
                            const unsigned lo = hex4();
                            TORCH_CHECK(lo >= 0xDC00 && lo < 0xE000, "config_json: unpaired high surrogate");
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        put_utf8(j.str, cp);
                        break;
                    }
                    default: TORCH_CHECK(false, "config_json: unknown escape \\", std::string(1, e));
                }
            }
        }
        if (!s.compare(i, 4, "true")) { j.kind = Json::Bool; j.b = true; i += 4; return j; }
        if (!s.compare(i, 5, "false")) { j.kind = Json::Bool; i += 5; return j; }
        if (!s.compare(i, 4, "null")) { i += 4; return j; }
        // a number, parsed in place (std::string is NUL-terminated: strtod stops at the terminator at the latest; no copy of the tail)
        const char* b = s.c_str() + i;
        char* end = nullptr;
        j.kind = Json::Num; j.num = std::strtod(b, &end);
        TORCH_CHECK(end != b, "config_json: unexpected character '", std::string(1, c), "' at offset ", i);
        i += (size_t)(end - b);
        return j;
    }
};
double num_of(const Json& o, const char* key, double dflt, bool required = false) {
    const Json* v = o.find(key);
    if (!v || v->kind == Json::Null) { TORCH_CHECK(!required, "fs2::twin_inference: config_json lacks '", key, "'"); return dflt; }
    return v->kind == Json::Bool ? (v->b ? 1.0 : 0.0) : v->num;
}

struct Entry {
    fs2_handle* h = nullptr;
    at::Tensor flat;                  // keeps the storage the handle was loaded from alive (and pins its data_ptr: the cache key stays unique)
    const void* ptr = nullptr; int64_t numel = 0; uint32_t version = 0; int dev = 0; std::string config;
    int idim = 0, odim = 0, adim = 0, rf = 1, precision = FS2_PREC_FP32, pe_rows_enc = 0, pe_rows_dec = 0;
    std::vector<std::pair<std::string, std::vector<int64_t>>> tensors;
    std::map<std::string, at::Tensor> grown;      // positional tables recomputed for longer utterances (reference embedding.py:50-56 extend_pe)
};
std::mutex g_mu;
// The cache is reached through a pointer that is never deleted: its entries hold device tensors and library handles, and static
// destruction at process exit may run after the HIP runtime / torch's caching allocator have been torn down.  Eviction (below) is
// what frees an entry: fs2_destroy for the handle's device allocations, then the tensors.
std::list<Entry>& lru() { static std::list<Entry>* l = new std::list<Entry>(); return *l; }      // front = most recent
size_t max_entries() {      // FS2_TWIN_CACHE (1 .. 16, default 2): every entry is a full copy of the weights on the device
    static const size_t n = [] { const char* e = getenv("FS2_TWIN_CACHE"); const long v = e ? atol(e) : 2; return (size_t)std::min<long>(std::max<long>(v, 1), 16); }();
    return n;
}
void check_abi() {      // a stale libfs2_torch.so next to a newer libfs2_hip.so (or the reverse) must not get as far as a struct mismatch
    static const bool ok = [] {
        const uint32_t v = fs2_abi_version();
        TORCH_CHECK(v == FS2_ABI_VERSION, "libfs2_torch.so was built against include/fs2.h ABI revision ", FS2_ABI_VERSION,
                    " but libfs2_hip.so reports revision ", v, ": rebuild both (python __graft_entry__.py)");
        return true;
    }();
    (void)ok;
}

void check(int rc, fs2_handle* h, const char* what) {
    TORCH_CHECK(rc == FS2_OK, "libfs2_hip: ", what, " failed (", rc, "): ", fs2_last_error(h));
}

at::Tensor sinusoid_table(int64_t n, int64_t d, const at::Device& dev) {      // embedding.py:57-66, as the Python module builds it
    auto pos = at::arange(0, n, at::TensorOptions().dtype(at::kFloat)).unsqueeze(1);
    auto div = at::exp(at::arange(0, d, 2, at::TensorOptions().dtype(at::kFloat)) * (float)(-(std::log(10000.0) / (double)d)));
    auto pe = at::zeros({n, d}, at::TensorOptions().dtype(at::kFloat));
    pe.slice(1, 0, d, 2).copy_(at::sin(pos * div));
    pe.slice(1, 1, d, 2).copy_(at::cos(pos * div));
    return pe.unsqueeze(0).to(dev).contiguous();
}

void load_weights(Entry& e, hipStream_t stream) {
    std::vector<fs2_tensor_desc> descs;
    const float* base = e.flat.data_ptr<float>();
    int64_t off = 0;
    for (auto& t : e.tensors) {
        int64_t n = 1;
        for (auto v : t.second) n *= v;
        fs2_tensor_desc d;
        memset(&d, 0, sizeof d);
        d.name = t.first.c_str(); d.ndim = (int32_t)t.second.size();
        TORCH_CHECK(d.ndim <= 4, "fs2::twin_inference: tensor ", t.first, " has more than 4 dimensions");
        for (int k = 0; k < d.ndim; ++k) d.shape[k] = t.second[k];
        auto g = e.grown.find(t.first);
        if (g != e.grown.end()) {
            d.data = g->second.data_ptr<float>();
            for (int k = 0; k < d.ndim; ++k) d.shape[k] = g->second.size(k);
        } else d.data = base + off;
        descs.push_back(d);
        off += n;
    }
    TORCH_CHECK(off == e.numel, "fs2::twin_inference: the flat weight buffer has ", e.numel, " elements, the manifest describes ", off);
    check(fs2_load_weights(e.h, descs.data(), (int32_t)descs.size(), stream), e.h, "fs2_load_weights");
}

Entry& get_entry(const at::Tensor& flat, const std::string& config, hipStream_t stream) {
    const void* ptr = flat.data_ptr();
    const uint32_t ver = flat._version();
    const int dev = flat.get_device();
    check_abi();
    std::list<Entry>& g_lru = lru();
    for (auto it = g_lru.begin(); it != g_lru.end(); ++it)
        if (it->ptr == ptr && it->numel == flat.numel() && it->version == ver && it->dev == dev && it->config == config) {
            g_lru.splice(g_lru.begin(), g_lru, it);
            return g_lru.front();
        }
    JsonParser jp(config);
    const Json root = jp.parse();
    const Json* hp = root.find("hp");
    TORCH_CHECK(hp && hp->find("model"), "fs2::twin_inference: config_json lacks hp.model");
    const Json& m = *hp->find("model");
    Entry e;
    e.flat = flat; e.ptr = ptr; e.numel = flat.numel(); e.version = ver; e.dev = dev; e.config = config;
    e.idim = (int)num_of(root, "idim", 0, true); e.odim = (int)num_of(root, "odim", 0, true);
    const Json* prec = root.find("precision");
    if (prec && prec->kind == Json::Str) {
        static const std::map<std::string, int> names = {{"fp32", FS2_PREC_FP32}, {"bf16x3", FS2_PREC_BF16X3}, {"bf16", FS2_PREC_BF16},
                                                         {"mix_f16x2", FS2_PREC_MIX_F16X2}, {"mix_f16x1", FS2_PREC_MIX_F16X1}, {"mix_mx", FS2_PREC_MIX_MX}};
        auto it = names.find(prec->str);
        TORCH_CHECK(it != names.end(), "fs2::twin_inference: unknown precision '", prec->str, "'");
        e.precision = it->second;
    }
    const Json* ts = root.find("tensors");
    TORCH_CHECK(ts && ts->kind == Json::Arr, "fs2::twin_inference: config_json lacks the tensor manifest (re-export with this version of fastspeech2_amd)");
    for (const Json& t : ts->arr) {
        TORCH_CHECK(t.kind == Json::Arr && t.arr.size() == 2 && t.arr[0].kind == Json::Str, "fs2::twin_inference: malformed manifest entry");
        std::vector<int64_t> shape;
        for (const Json& d : t.arr[1].arr) shape.push_back((int64_t)d.num);
        e.tensors.emplace_back(t.arr[0].str, shape);
        if (t.arr[0].str == "encoder.embed.1.pe" && shape.size() == 3) e.pe_rows_enc = (int)shape[1];
        if (t.arr[0].str == "decoder.embed.0.pe" && shape.size() == 3) e.pe_rows_dec = (int)shape[1];
    }
    // the twin's architecture (utils/fastspeech2_script.py:112-145): decoder at adim, positional encoding as its only input layer
    fs2_config c;
    memset(&c, 0, sizeof c);
    c.struct_size = sizeof c;
    c.idim = e.idim; c.odim = e.odim;
    c.adim = e.adim = (int)num_of(m, "adim", 0, true); c.aheads = (int)num_of(m, "aheads", 0, true);
    c.elayers = (int)num_of(m, "elayers", 0, true); c.eunits = (int)num_of(m, "eunits", 0, true);
    c.ddim = c.adim; c.dlayers = (int)num_of(m, "dlayers", 0, true); c.dunits = (int)num_of(m, "dunits", 0, true);
    const Json* plt = m.find("positionwise_layer_type");
    const bool conv = !plt || plt->str == "conv1d";
    c.ffn_kernel = conv ? (int)num_of(m, "positionwise_conv_kernel_size", 1) : 1;
    c.dur_layers = (int)num_of(m, "duration_predictor_layers", 2); c.dur_chans = (int)num_of(m, "duration_predictor_chans", 256);
    c.dur_kernel = (int)num_of(m, "duration_predictor_kernel_size", 3);
    c.var_layers = 2; c.var_chans = 256; c.var_kernel = 3; c.n_bins = 256;      // hard-wired in the reference (variance_predictor.py:125,198)
    c.postnet_layers = (int)num_of(m, "postnet_layers", 0); c.postnet_chans = (int)num_of(m, "postnet_chans", 256);
    c.postnet_filts = (int)num_of(m, "postnet_filts", 5); c.use_batch_norm = (int)num_of(m, "use_batch_norm", 1);
    c.use_scaled_pos_enc = (int)num_of(m, "use_scaled_pos_enc", 1); c.reduction_factor = (int)num_of(m, "reduction_factor", 1);
    c.device = dev; c.decoder_input_layer = 0;
    e.rf = c.reduction_factor;
    c.enc_normalize_before = (int)num_of(m, "encoder_normalize_before", 0); c.dec_normalize_before = (int)num_of(m, "decoder_normalize_before", 0);
    c.enc_concat_after = (int)num_of(m, "encoder_concat_after", 0); c.dec_concat_after = (int)num_of(m, "decoder_concat_after", 0);
    check(fs2_create(&c, &e.h), nullptr, "fs2_create");
    try {
        load_weights(e, stream);
    } catch (...) {
        fs2_destroy(e.h);
        throw;
    }
    while (g_lru.size() >= max_entries()) {
        // (said once: a process that alternates between more scripted models than the cache holds re-uploads the weights on every call)
        TORCH_WARN_ONCE("fs2::twin_inference: evicting a cached model (FS2_TWIN_CACHE = ", max_entries(), " handles per process); set FS2_TWIN_CACHE "
                        "to the number of scripted models this process alternates between");
        fs2_destroy(g_lru.back().h); g_lru.pop_back();
    }
    g_lru.push_front(std::move(e));
    return g_lru.front();
}

at::Tensor twin_inference(const at::Tensor& x_in, const at::Tensor& flat_weights, std::string config_json) {
    TORCH_CHECK(flat_weights.is_cuda(),
                "fs2::twin_inference runs on an MI355X only: the module's weights are on ", flat_weights.device(),
                " (no CPU fallback): move the exported module to a HIP device first, e.g. torch.jit.load(path, map_location='cuda') "
                "or module.cuda()");
    TORCH_CHECK(flat_weights.scalar_type() == at::kFloat && flat_weights.is_contiguous(), "fs2::twin_inference: flat_weights must be contiguous float32");
    TORCH_CHECK(x_in.dim() == 1 && x_in.numel() > 0, "fs2::twin_inference: x must be a non-empty [T] tensor of phoneme ids");
    const at::Device dev = flat_weights.device();
    at::Tensor xs = x_in.to(dev, at::kLong).contiguous().unsqueeze(0);      // (a CPU example input, as export_torchscript.py traces with, is moved)
    const int64_t T = xs.size(1);
    c10::hip::HIPGuard guard(dev.index());
    hipStream_t stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    std::lock_guard<std::mutex> lock(g_mu);
    Entry& e = get_entry(flat_weights, config_json, stream);
    auto grow = [&](const char* name, int& rows, int64_t need) {
        if (need <= rows) return false;
        const int64_t n = std::max<int64_t>(need, 2 * (int64_t)rows);
        e.grown[name] = sinusoid_table(n, e.adim, dev);
        rows = (int)n;
        return true;
    };
    if (grow("encoder.embed.1.pe", e.pe_rows_enc, T)) load_weights(e, stream);
    const auto bytes = at::TensorOptions().dtype(at::kByte).device(dev);
    int64_t ilen = T;
    fs2_batch batch;
    memset(&batch, 0, sizeof batch);
    batch.B = 1; batch.Tmax = (int32_t)T; batch.ilens = &ilen; batch.compat_padded = 0; batch.precision = e.precision;
    at::Tensor after;
    for (int attempt = 0; attempt < 2; ++attempt) {
        at::Tensor tok_ws = at::empty({(int64_t)fs2_token_workspace_bytes(e.h, &batch)}, bytes);
        at::Tensor olens_dev = at::empty({1}, at::TensorOptions().dtype(at::kLong).device(dev));
        fs2_encode_io eio;
        memset(&eio, 0, sizeof eio);
        eio.struct_size = sizeof eio; eio.batch = batch; eio.xs = xs.data_ptr<int64_t>(); eio.olens = olens_dev.data_ptr<int64_t>();
        eio.workspace = tok_ws.data_ptr(); eio.workspace_bytes = (size_t)tok_ws.numel(); eio.duration_alpha = 1.f;
        check(fs2_encode(e.h, stream, &eio), e.h, "fs2_encode");
        int64_t frames = olens_dev.cpu().item<int64_t>();      // the one host sync of the path (the frame count shapes the result)
        if (grow("decoder.embed.0.pe", e.pe_rows_dec, frames)) { load_weights(e, stream); continue; }      // longer table: redo the encoder
        at::Tensor frm_ws = at::empty({(int64_t)fs2_frame_workspace_bytes(e.h, &batch, &frames)}, bytes);
        after = at::empty({1, frames * e.rf, (int64_t)e.odim}, at::TensorOptions().dtype(at::kFloat).device(dev));      // r mel frames per decoder frame
        fs2_decode_io dio;
        memset(&dio, 0, sizeof dio);
        dio.struct_size = sizeof dio; dio.batch = batch; dio.olens = &frames; dio.Lmax = (int32_t)frames; dio.after = after.data_ptr<float>();
        dio.token_workspace = tok_ws.data_ptr(); dio.workspace = frm_ws.data_ptr(); dio.workspace_bytes = (size_t)frm_ws.numel();
        check(fs2_decode(e.h, stream, &dio), e.h, "fs2_decode");
        break;
    }
    return after.squeeze(0);
}

}  // namespace

TORCH_LIBRARY(fs2, m) { m.def("twin_inference(Tensor x, Tensor flat_weights, str config_json) -> Tensor"); }
TORCH_LIBRARY_IMPL(fs2, CompositeExplicitAutograd, m) { m.impl("twin_inference", twin_inference); }
