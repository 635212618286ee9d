#!/usr/bin/env python3
"""CPU simulation of reduced-MFMA arithmetic for the P.V product of attention (second half), operator level,
numpy only.  P = softmax probabilities relative to the running row maximum (<= 1: the attn_bf16 form), V = a LayerNorm-scale
activation.  Schemes (MFMA-equivalents per product at bf16 rate; an e4m3 K = 128 MFMA runs at twice that rate):
  bf16x3      P = ph + pl, V = vh + vl (bf16):  ph.vh + ph.vl + pl.vh                                   3     (what attn_bf16 / attn_w32 do)
  f16x1       fp16(P) . fp16(V)                                                                         1
  f16 + mx    fp16(P).fp16(V) + e4m3(rP 2^a).fp16(V) 2^-a + fp16(P).e4m3(rV 2^b) 2^-b                   2     (static scales a, b; the `mix_mx` trick)
  f16 + mx8   the same with BOTH operands of the cross terms in e4m3 (what v_mfma_scale_f32_*_f8f6f4 takes)  2
Prints the error of the context against float64, relative to max |ctx|, for score distributions of increasing peakedness.

  python tools/arith_sim_attention.py
"""
import numpy as np


def to_bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def to_f16(x):
    return x.astype(np.float16).astype(np.float32)


def to_e4m3(x):
    """round to nearest e4m3 (bias 7, max 448, subnormals of 2^-9), saturating"""
    x = np.clip(x.astype(np.float64), -448.0, 448.0)
    a = np.abs(x)
    e = np.floor(np.log2(np.maximum(a, 2.0 ** -9)))
    e = np.maximum(e, -6.0)                       # subnormal range shares the exponent of 2^-6
    q = 2.0 ** (e - 3)                            # 3 mantissa bits
    return (np.sign(x) * np.round(a / q) * q).astype(np.float32)


def run(L, dk, temp, rs):
    q = rs.standard_normal((L, dk)).astype(np.float32)
    k = rs.standard_normal((L, dk)).astype(np.float32)
    v = (rs.standard_normal((L, dk)) * 1.0).astype(np.float32)
    s = (q @ k.T).astype(np.float64) / np.sqrt(dk) * temp
    p = np.exp(s - s.max(-1, keepdims=True))      # <= 1, unnormalised (the kernels normalise at the end, in fp32)
    l = p.sum(-1, keepdims=True)
    ref = (p @ v.astype(np.float64)) / l
    p32 = p.astype(np.float32)
    out = {}
    ph = to_bf16(p32); pl = to_bf16(p32 - ph); vh = to_bf16(v); vl = to_bf16(v - vh)
    f64 = lambda a, b: a.astype(np.float64) @ b.astype(np.float64)
    out["bf16x3"] = (f64(ph, vh) + f64(ph, vl) + f64(pl, vh)) / l
    p16 = to_f16(p32); v16 = to_f16(v)
    out["f16x1"] = f64(p16, v16) / l
    rp, rv = p32 - p16, v - v16
    a = 2.0 ** np.floor(np.log2(448.0 / max(np.abs(rp).max(), 1e-30)))      # static scales: the largest residual lands on the top binade
    b = 2.0 ** np.floor(np.log2(448.0 / max(np.abs(rv).max(), 1e-30)))
    rp8, rv8 = to_e4m3(rp * a) / a, to_e4m3(rv * b) / b
    out["f16 + mx"] = (f64(p16, v16) + f64(rp8, v16) + f64(p16, rv8)) / l
    # both operands of the cross terms in e4m3 (per-tensor scales for fp16(P) <= 1 and fp16(V) as well)
    cp = 2.0 ** np.floor(np.log2(448.0 / max(np.abs(p16).max(), 1e-30)))
    cv = 2.0 ** np.floor(np.log2(448.0 / max(np.abs(v16).max(), 1e-30)))
    p8, v8 = to_e4m3(p16 * cp) / cp, to_e4m3(v16 * cv) / cv
    out["f16 + mx8"] = (f64(p16, v16) + f64(rp8, v8) + f64(p8, rv8)) / l
    m = np.abs(ref).max()
    return {k_: (np.abs(o - ref).max() / m, np.sqrt(((o - ref) ** 2).mean()) / m) for k_, o in out.items()}, float(1.0 / (p / l).max(-1).mean())


def run_qk(L, dk, temp, rs):
    """the same split on Q.K^T (Q pre-scaled by log2 e / sqrt d_k as the QKV epilogue leaves it): error of the context when only the scores are computed in the scheme"""
    q = (rs.standard_normal((L, dk)) * np.sqrt(temp)).astype(np.float32)
    k = (rs.standard_normal((L, dk)) * np.sqrt(temp)).astype(np.float32)
    v = rs.standard_normal((L, dk)).astype(np.float32)
    qs = (q * (1.4426950408889634 / np.sqrt(dk))).astype(np.float32)
    f64 = lambda a, b: a.astype(np.float64) @ b.astype(np.float64)

    def ctx_of(s):
        p = np.exp2(s - s.max(-1, keepdims=True))
        return (p @ v.astype(np.float64)) / p.sum(-1, keepdims=True)
    s_ref = f64(qs, k.T)
    ref = ctx_of(s_ref)
    out = {}
    qh = to_bf16(qs); ql = to_bf16(qs - qh); kh = to_bf16(k); kl = to_bf16(k - kh)
    out["bf16x3"] = ctx_of(f64(qh, kh.T) + f64(qh, kl.T) + f64(ql, kh.T))
    q16, k16 = to_f16(qs), to_f16(k)
    rq, rk = qs - q16, k - k16
    out["f16x1"] = ctx_of(f64(q16, k16.T))
    sc = lambda x: 2.0 ** np.floor(np.log2(448.0 / np.abs(x).max()))
    rq8, rk8 = to_e4m3(rq * sc(rq)) / sc(rq), to_e4m3(rk * sc(rk)) / sc(rk)
    q8, k8 = to_e4m3(q16 * sc(q16)) / sc(q16), to_e4m3(k16 * sc(k16)) / sc(k16)
    out["f16 + mx"] = ctx_of(f64(q16, k16.T) + f64(rq8, k16.T) + f64(q16, rk8.T))
    out["f16 + mx8"] = ctx_of(f64(q16, k16.T) + f64(rq8, k8.T) + f64(q8, rk8.T))
    m = np.abs(ref).max()
    return {k_: (np.abs(o - ref).max() / m, np.sqrt(((o - ref) ** 2).mean()) / m) for k_, o in out.items()}, float(np.abs(s_ref).max())


def main():
    rs = np.random.RandomState(0)
    print("P.V arithmetic, context error relative to max|ctx| (max | rms); L = 1024 keys, d_k = 192")
    print("%-34s %-22s %-22s %-22s %-22s" % ("scores", "bf16x3 (3 MFMA)", "f16x1 (1)", "f16 + mx (2)", "f16 + mx8 (2)"))
    for temp, name in ((1.0, "N(0,1): ~flat rows"), (3.0, "x3: a few keys dominate"), (8.0, "x8: near one-hot rows")):
        r, eff = run(1024, 192, temp, rs)
        print("%-34s " % ("%s (1/mean max p = %.1f)" % (name, eff)) + " ".join("%.1e | %.1e     " % r[k] for k in ("bf16x3", "f16x1", "f16 + mx", "f16 + mx8")))
    print()
    print("Q.K^T in the scheme (P.V exact), context error relative to max|ctx| (max | rms)")
    for temp in (1.0, 3.0, 8.0):
        r, smax = run_qk(1024, 192, temp, rs)
        print("%-34s " % ("max |score| = %.0f (log2 domain)" % smax) + " ".join("%.1e | %.1e     " % r[k] for k in ("bf16x3", "f16x1", "f16 + mx", "f16 + mx8")))


if __name__ == "__main__":
    main()
