"""Thin test-side wrappers around the fs2_op_* C-ABI entry points (tensors in, tensors out)."""
import ctypes as C

import torch

from fastspeech2_amd import _lib


def _st(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _p(t):
    return t.data_ptr() if t is not None else None


def conv_gemm(x, w, bias=None, resid=None, relu_pre=False, ln=None, ln_eps=1e-5, act_post=0, dot=None,
              row_valid=None, precision="fp32"):
    """x [R,C]; w [N,C,k] or [N,C].  Returns (y [R,N], dot_out [R] or None)."""
    L = _lib.lib()
    R, Cc = x.shape
    N = w.shape[0]
    k = w.shape[2] if w.dim() == 3 else 1
    x, w = x.contiguous(), w.contiguous()
    y = torch.full((R, N), float("nan"), device=x.device)
    dot_out = torch.full((R,), float("nan"), device=x.device) if dot is not None else None
    keep = [t.contiguous() if t is not None else None for t in
            (bias, resid, ln[0] if ln else None, ln[1] if ln else None, dot[0] if dot else None, dot[1] if dot else None,
             row_valid.int() if row_valid is not None else None)]
    a = _lib.OpGemmArgs(R, Cc, N, k, _lib.PRECISIONS[precision], _p(x), _p(w), _p(keep[0]), _p(keep[1]), int(relu_pre),
                        _p(keep[2]), _p(keep[3]), float(ln_eps), int(act_post), _p(keep[4]), _p(keep[5]), _p(dot_out),
                        _p(y), _p(keep[6]))
    with torch.cuda.device(x.device):
        _lib.check(L.fs2_op_conv_gemm(_st(x.device), C.byref(a)))
    torch.cuda.synchronize()
    return y, dot_out


def attention(qkv, D, heads, starts, lens, klens, mask_q, precision="fp32"):
    L = _lib.lib()
    B = len(starts)
    qkv = qkv.contiguous()
    ctx = torch.zeros(qkv.shape[0], D, device=qkv.device)
    arr = lambda v: (C.c_int32 * B)(*[int(i) for i in v])
    with torch.cuda.device(qkv.device):
        _lib.check(L.fs2_op_attention(_st(qkv.device), _p(qkv), _p(ctx), D, heads, B, arr(starts), arr(lens), arr(klens),
                                      int(mask_q), _lib.PRECISIONS[precision]))
    torch.cuda.synchronize()
    return ctx


def length_regulate(hs, ds, ilens, Lmax, alpha=1.0):
    L = _lib.lib()
    B, Tmax, D = hs.shape
    hs, ds = hs.contiguous(), ds.contiguous().long()
    out = torch.full((B, Lmax, D), float("nan"), device=hs.device)
    idx = torch.full((B, Lmax), -7, dtype=torch.int32, device=hs.device)
    olens = torch.zeros(B, dtype=torch.int64, device=hs.device)
    il = (C.c_int64 * B)(*[int(i) for i in ilens])
    with torch.cuda.device(hs.device):
        _lib.check(L.fs2_op_length_regulate(_st(hs.device), _p(hs), _p(ds), il, B, Tmax, D, Lmax, float(alpha), _p(out), _p(idx), _p(olens)))
    torch.cuda.synchronize()
    return out, idx, olens


def bucketize(x, bins):
    L = _lib.lib()
    x = x.contiguous().float()
    idx = torch.full(x.shape, -1, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.fs2_op_bucketize(_st(x.device), _p(x), x.numel(), _p(bins.contiguous()), bins.numel(), _p(idx)))
    torch.cuda.synchronize()
    return idx


def duration(d_log):
    """clamp(round_half_even(exp(y) - 1), 0) -> int64 through fs2_op_duration."""
    L = _lib.lib()
    y = d_log.contiguous().float()
    d = torch.full(y.shape, -1, dtype=torch.int64, device=y.device)
    with torch.cuda.device(y.device):
        _lib.check(L.fs2_op_duration(_st(y.device), _p(y), y.numel(), _p(d)))
    torch.cuda.synchronize()
    return d
