"""ctypes binding of libfs2_hip.so (C ABI in include/fs2.h).

There is no CPU or eager-PyTorch fallback: if the shared library is missing or cannot be
loaded, importing the product path raises immediately (``Fs2LibraryError``).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfs2_hip.so")
CSRC = os.path.join(_HERE, "csrc")

FS2_PREC_FP32, FS2_PREC_BF16X3, FS2_PREC_BF16, FS2_PREC_MIX_F16X2, FS2_PREC_MIX_F16X1, FS2_PREC_MIX_MX, FS2_PREC_MIX_MX4 = 0, 1, 2, 3, 4, 5, 6
PRECISIONS = {"fp32": FS2_PREC_FP32, "bf16x3": FS2_PREC_BF16X3, "bf16": FS2_PREC_BF16, "mix_f16x2": FS2_PREC_MIX_F16X2,
              "mix_f16x1": FS2_PREC_MIX_F16X1, "mix_mx": FS2_PREC_MIX_MX, "mix_mx4": FS2_PREC_MIX_MX4}


class Fs2LibraryError(RuntimeError):
    pass


class Fs2Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libfs2_hip error %d: %s" % (code, msg))
        self.code = code


class _Sized(C.Structure):
    """ABI structs that start with ``struct_size`` (include/fs2.h): filled in here, so call sites construct the mirrors with the
    header's remaining fields only; the library refuses a struct whose size differs from its own sizeof."""

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class Config(_Sized):
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in (
        "idim", "odim", "adim", "aheads", "elayers", "eunits", "ddim", "dlayers", "dunits", "ffn_kernel",
        "dur_layers", "dur_chans", "dur_kernel", "var_layers", "var_chans", "var_kernel", "n_bins",
        "postnet_layers", "postnet_chans", "postnet_filts", "use_batch_norm", "use_scaled_pos_enc",
        "reduction_factor", "device", "decoder_input_layer", "enc_normalize_before", "dec_normalize_before", "enc_concat_after",
        "dec_concat_after")]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class Batch(C.Structure):
    """fs2_batch.  ``regime_tokens`` / ``regime_utterances`` (0 / 0 = this batch's own): the batch the kernel variants are chosen for --
    a shard of a larger batch names the whole batch and is then computed bit-identically to the one-call run of the whole batch."""
    _fields_ = [("B", C.c_int32), ("Tmax", C.c_int32), ("ilens", C.POINTER(C.c_int64)),
                ("compat_padded", C.c_int32), ("precision", C.c_int32), ("regime_tokens", C.c_int64), ("regime_utterances", C.c_int32)]


class EncodeIO(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("batch", Batch), ("xs", C.c_void_p), ("ds", C.c_void_p), ("d_log", C.c_void_p),
                ("d_int", C.c_void_p), ("olens", C.c_void_p), ("enc_out", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("duration_alpha", C.c_float)]


class DecodeIO(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("batch", Batch), ("olens", C.POINTER(C.c_int64)), ("Lmax", C.c_int32), ("masked", C.c_int32),
                ("es", C.c_void_p), ("ps", C.c_void_p), ("es_stride", C.c_int32), ("ps_stride", C.c_int32),
                ("before", C.c_void_p), ("after", C.c_void_p), ("e_out", C.c_void_p), ("p_out", C.c_void_p),
                ("qe", C.c_void_p), ("qp", C.c_void_p), ("lr_index", C.c_void_p), ("dec_out", C.c_void_p),
                ("token_workspace", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("after_packed", C.c_void_p), ("row_capacity", C.c_int64), ("status", C.c_void_p)]


class OpGemmArgs(_Sized):
    _fields_ = [("struct_size", C.c_uint32), ("R", C.c_int32), ("C", C.c_int32), ("N", C.c_int32), ("ktaps", C.c_int32), ("precision", C.c_int32),
                ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("resid", C.c_void_p),
                ("relu_pre", C.c_int32), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
                ("act_post", C.c_int32), ("dot_w", C.c_void_p), ("dot_b", C.c_void_p), ("dot_out", C.c_void_p),
                ("y", C.c_void_p), ("row_valid", C.c_void_p)]


# every symbol include/fs2.h declares (tests check the library exports all of them)
ABI_VERSION = 4      # FS2_ABI_VERSION of the include/fs2.h these mirrors were written against (checked in lib())

EXPORTS = ["fs2_abi_version", "fs2_create", "fs2_destroy", "fs2_last_error", "fs2_load_weights", "fs2_token_workspace_bytes",
           "fs2_encode", "fs2_frame_workspace_bytes", "fs2_row_capacity", "fs2_frame_workspace_bytes_cap", "fs2_decode", "fs2_set_profiling", "fs2_set_profile_filter", "fs2_get_profile",
           "fs2_op_conv_gemm", "fs2_op_attention", "fs2_op_length_regulate", "fs2_op_unpack_rows", "fs2_op_unpack_rows_dev", "fs2_op_transpose", "fs2_op_bucketize", "fs2_op_duration", "fs2_set_option", "fs2_get_option", "fs2_get_counter"]

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"]


AUDIT_PATH = os.path.join(_HERE, "libfs2_hip.audit.json")


def _sha16(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    """Compile csrc/fs2_runtime.hip for gfx950 into fastspeech2_amd/libfs2_hip.so (in-tree) and audit the device assembly of exactly
    this binary (_audit.py: the kernels whose accumulators are literal registers -- attn_w32, gemm_row4_bf16); the outcome is recorded
    next to the library (libfs2_hip.audit.json, tied to the binary's hash).  A violation raises: such a library must not ship."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(_HERE, "_audit.py")]
    hdr = os.path.join(_HERE, "..", "include", "fs2.h")
    if not force and os.path.exists(LIB_PATH):
        newest = max(os.path.getmtime(p) for p in srcs + [hdr])
        if os.path.getmtime(LIB_PATH) >= newest and audit_record() is not None:
            return LIB_PATH
    import glob
    import json
    import tempfile
    from . import _audit
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + ["-save-temps", os.path.join(CSRC, "fs2_runtime.hip"), "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    with tempfile.TemporaryDirectory(prefix="fs2_build_") as td:      # (-save-temps drops the intermediates into the working directory)
        subprocess.run(cmd, check=True, cwd=td)
        asm = glob.glob(os.path.join(td, "*gfx950*.s"))
        if len(asm) != 1:
            raise Fs2LibraryError("build: expected one gfx950 assembly listing from -save-temps, found %s" % asm)
        kernels = _audit.audit_file(asm[0])
    bad = {k: v["violations"] for k, v in kernels.items() if v["violations"]}
    rec = dict(so_sha16=_sha16(LIB_PATH), hipcc=subprocess.run([hipcc, "--version"], capture_output=True, text=True).stdout.strip().split("\n")[0],
               kernels={k: {a: b for a, b in v.items() if a != "violations"} for k, v in kernels.items()},
               violations=sum(len(v) for v in bad.values()),
               clean=not bad and all(sum(1 for v in kernels.values() if v["kind"] == k) >= n for k, n in _audit.EXPECTED_KERNELS.items()))
    with open(AUDIT_PATH, "w") as f:
        json.dump(rec, f, indent=1)
    if bad:
        raise Fs2LibraryError("ISA audit of %s failed (the library stays on disk with FS2_ATTN_W32 / FS2_ROW4 forced off, see libfs2_hip.audit.json):\n%s"
                              % (LIB_PATH, "\n".join("%s: %s" % (k, v[:3]) for k, v in bad.items())))
    return LIB_PATH


def audit_record():
    """The audit record of the library on disk, or None if there is none / it belongs to another binary."""
    import json
    try:
        with open(AUDIT_PATH) as f:
            rec = json.load(f)
        return rec if rec.get("so_sha16") == _sha16(LIB_PATH) else None
    except (OSError, ValueError):
        return None


TORCH_OP_PATH = os.path.join(_HERE, "libfs2_torch.so")


def build_torch_op(force=False, verbose=False):
    """Compile csrc/fs2_torch_op.cpp (the C++ dispatcher op fs2::twin_inference behind the TorchScript twin) against this interpreter's
    libtorch and libfs2_hip.so into fastspeech2_amd/libfs2_torch.so (in-tree; rpath $ORIGIN finds libfs2_hip.so next to it)."""
    src = os.path.join(CSRC, "fs2_torch_op.cpp")
    hdr = os.path.join(_HERE, "..", "include", "fs2.h")
    build(force=False)
    # (rebuilt when its source, the header OR libfs2_hip.so is newer: the op links against the library's ABI, and checks
    #  fs2_abi_version() against its own FS2_ABI_VERSION at first use)
    if not force and os.path.exists(TORCH_OP_PATH) and os.path.getmtime(TORCH_OP_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr),
                                                                                             os.path.getmtime(LIB_PATH)):
        return TORCH_OP_PATH
    import torch
    tdir = os.path.dirname(torch.__file__)
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi, "-Wno-deprecated-declarations",
           "-I", os.path.join(tdir, "include"), "-I", os.path.join(tdir, "include", "torch", "csrc", "api", "include"), "-I", "/opt/rocm/include",
           src, "-o", TORCH_OP_PATH, "-L", os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
           "-L", _HERE, "-lfs2_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return TORCH_OP_PATH


_lib = None


def lib():
    """Load the shared library (once).  Raises Fs2LibraryError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Fs2LibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:
        raise Fs2LibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
    vp, i32, i64p = C.c_void_p, C.c_int32, C.POINTER(C.c_int64)
    try:
        L.fs2_abi_version.restype = C.c_int32
        got = int(L.fs2_abi_version())
    except AttributeError:
        got = None
    if got != ABI_VERSION:
        raise Fs2LibraryError("%s implements ABI revision %s, this binding revision %d: rebuild it with "
                              "`python -c 'import __graft_entry__ as g; g.build()'`" % (LIB_PATH, got, ABI_VERSION))
    L.fs2_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.fs2_create.restype = C.c_int
    L.fs2_destroy.argtypes = [vp]
    L.fs2_destroy.restype = None
    L.fs2_last_error.argtypes = [vp]
    L.fs2_last_error.restype = C.c_char_p
    L.fs2_load_weights.argtypes = [vp, C.POINTER(TensorDesc), i32, vp]
    L.fs2_load_weights.restype = C.c_int
    L.fs2_token_workspace_bytes.argtypes = [vp, C.POINTER(Batch)]
    L.fs2_token_workspace_bytes.restype = C.c_size_t
    L.fs2_encode.argtypes = [vp, vp, C.POINTER(EncodeIO)]
    L.fs2_encode.restype = C.c_int
    L.fs2_frame_workspace_bytes.argtypes = [vp, C.POINTER(Batch), i64p]
    L.fs2_frame_workspace_bytes.restype = C.c_size_t
    L.fs2_row_capacity.argtypes = [C.POINTER(Batch), C.c_int64]
    L.fs2_row_capacity.restype = C.c_int64
    L.fs2_frame_workspace_bytes_cap.argtypes = [vp, C.POINTER(Batch), C.c_int64, i32]
    L.fs2_frame_workspace_bytes_cap.restype = C.c_size_t
    L.fs2_decode.argtypes = [vp, vp, C.POINTER(DecodeIO)]
    L.fs2_decode.restype = C.c_int
    L.fs2_set_profiling.argtypes = [vp, i32]
    L.fs2_set_profiling.restype = C.c_int
    L.fs2_set_profile_filter.argtypes = [vp, C.c_char_p]
    L.fs2_set_profile_filter.restype = C.c_int
    L.fs2_get_profile.argtypes = [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), i32]
    L.fs2_get_profile.restype = C.c_int
    L.fs2_op_conv_gemm.argtypes = [vp, C.POINTER(OpGemmArgs)]
    L.fs2_op_conv_gemm.restype = C.c_int
    L.fs2_op_attention.argtypes = [vp, vp, vp, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32, i32]
    L.fs2_op_attention.restype = C.c_int
    L.fs2_op_length_regulate.argtypes = [vp, vp, vp, i64p, i32, i32, i32, i32, C.c_float, vp, vp, vp]
    L.fs2_op_length_regulate.restype = C.c_int
    L.fs2_op_unpack_rows.argtypes = [vp, vp, i32, i32, C.POINTER(i32), C.POINTER(i32), i32, vp]
    L.fs2_op_unpack_rows.restype = C.c_int
    L.fs2_op_unpack_rows_dev.argtypes = [vp, vp, i32, i32, vp, vp, i32, vp]
    L.fs2_op_unpack_rows_dev.restype = C.c_int
    L.fs2_op_transpose.argtypes = [vp, vp, C.c_int64, i32, vp]
    L.fs2_op_transpose.restype = C.c_int
    L.fs2_op_bucketize.argtypes = [vp, vp, C.c_int64, vp, i32, vp]
    L.fs2_op_bucketize.restype = C.c_int
    L.fs2_op_duration.argtypes = [vp, vp, C.c_int64, vp]
    L.fs2_op_duration.restype = C.c_int
    L.fs2_set_option.argtypes = [C.c_char_p, i32]
    L.fs2_set_option.restype = C.c_int
    L.fs2_get_option.argtypes = [C.c_char_p, C.POINTER(i32)]
    L.fs2_get_option.restype = C.c_int
    L.fs2_get_counter.argtypes = [vp, vp, C.c_char_p, C.POINTER(C.c_int64), i32]
    L.fs2_get_counter.restype = C.c_int
    # The kernels with literal-register accumulators run only in a library whose ISA was audited (build()): the LIBRARY looks for the record of
    # its own hash when it is first used (fs2_runtime.hip: audit_clean) and otherwise starts with attn_w32 / gemm_row4_bf16 switched off -- for
    # every consumer, not only this binding; the compiler-scheduled kernels take their place (slower, never silently wrong).  Here: say so.
    v = i32(0)
    L.fs2_get_option(b"FS2_AUDIT_CLEAN", C.byref(v))
    if not v.value:
        import warnings
        rec = audit_record()
        warnings.warn("%s has no clean ISA audit record (%s): attn_w32 and gemm_row4_bf16 are switched off; rebuild with "
                      "`python -c 'import __graft_entry__ as g; g.build()'`" % (LIB_PATH, "missing or stale" if rec is None else "violations"))
    _lib = L
    return L


def get_option(name):
    """Value of a kernel-choice switch, or of a read-only fact about this binary: "FS2_AUDIT_CLEAN", "attn_w32_active", "row4_active",
    "qkv4_active" (include/fs2.h: fs2_get_option)."""
    v = C.c_int32(0)
    check(lib().fs2_get_option(name.encode(), C.byref(v)))
    return int(v.value)


def kernel_state():
    """What a benchmark line should say about the binary that produced it: the audit state and which hand-scheduled kernels may run."""
    return {k: bool(get_option(n)) for k, n in (("audit_clean", "FS2_AUDIT_CLEAN"), ("attn_w32_active", "attn_w32_active"),
                                                ("row4_active", "row4_active"), ("qkv4_active", "qkv4_active"))}


def set_option(name, value):
    """Kernel-choice switch for A/B measurements and tests (include/fs2.h: fs2_set_option); -1 = automatic."""
    check(lib().fs2_set_option(name.encode(), int(value)))


def check(code, handle=None):
    if code != 0:
        msg = lib().fs2_last_error(handle)
        raise Fs2Error(code, msg.decode() if msg else "?")
