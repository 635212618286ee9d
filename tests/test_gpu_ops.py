"""Per-kernel parity tests through the C ABI (fs2_op_*), each against a plain fp32 PyTorch statement of
the same reference op computed on the CPU.  Tolerances: 5 x the worst error measured on MI355X per arithmetic mode (GEMM_TOL / ATT_TOL below); integer
outputs bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def test_the_library_under_test_is_the_audited_one_with_every_hand_scheduled_kernel_allowed():
    """First GPU test of the suite: libfs2_hip.so found a clean ISA-audit record of its own hash, so attn_w32 /
    gemm_row4_bf16 / the QKV passes on it are allowed to run.  Without it every parity test below would still pass -- on attn_bf16 /
    gemm_row8_bf16 -- and every timing would be of other kernels; a benchmark line carries the same four flags (`kernels`)."""
    _dev()
    from fastspeech2_amd import _lib
    st = _lib.kernel_state()
    assert st == dict(audit_clean=True, attn_w32_active=True, row4_active=True, qkv4_active=True), st
    rec = _lib.audit_record()
    assert rec is not None and rec["clean"] and rec["violations"] == 0


def _rand(rs, *shape, scale=1.0):
    return torch.from_numpy(rs.uniform(-scale, scale, size=shape).astype(np.float32))


def _ref_conv(x, w, bias):
    """x [R,C] one sequence, zero padded; w [N,C,k]."""
    if w.dim() == 2:
        return F.linear(x, w, bias)
    k = w.shape[-1]
    return F.conv1d(x.t().unsqueeze(0), w, bias, padding=(k - 1) // 2)[0].t()


CASES = [
    # R,   C,    N,    k, bias, resid, relu_pre, ln_eps, act, dot,  name
    (200, 256, 256, 3, True, False, True, 1e-12, 0, True, "predictor conv + relu + LN + linear head"),
    (333, 256, 256, 3, True, False, True, 1e-12, 0, False, "predictor conv layer 0"),
    (300, 384, 1024, 9, True, False, False, None, 1, False, "decoder FFN conv k9 + relu (tile kernel)"),
    (130, 1024, 384, 1, True, True, False, 1e-5, 0, False, "FFN w_2 + residual + LN (rows NT=24)"),
    (257, 256, 256, 1, True, True, False, 1e-5, 0, False, "out-proj + residual + LN (rows NT=16)"),
    (190, 256, 768, 1, True, False, False, None, 0, False, "QKV projection (tile kernel)"),
    (210, 80, 256, 5, True, False, False, None, 2, False, "postnet layer 0: C=80 k5 tanh"),
    (210, 256, 80, 5, True, True, False, None, 0, False, "postnet last: N=80 + residual (rows NT=5)"),
    (100, 384, 80, 1, True, False, False, None, 0, False, "feat_out 384->80"),
    (70, 256, 384, 1, True, False, False, 1e-5, 1, False, "decoder input layer Linear+LN+ReLU"),
    (64, 256, 1024, 1, True, False, False, None, 1, False, "linear FFN (k=1) + relu, exact tile rows"),
    (290, 256, 384, 5, True, True, False, 1e-5, 1, False, "conv k5 + residual + LN + relu, N=384 (row-complete conv form)"),
]


# fp32: exact-fp32 MFMA chain vs CPU sum order.  bf16x3: operands carry ~16 mantissa bits.  bf16: 8 bits
# (reported, not a parity mode).
# Tolerances = 5 x the worst error measured on MI355X over all cases (gpurun_out/measured_errors.jsonl, round 2: fp32 5.3e-6,
# bf16x3 2.8e-5, bf16 1.4e-2), so that a regression of the arithmetic cannot hide below the 1e-3 end-to-end bar.
GEMM_TOL = {"fp32": 2.5e-5, "bf16x3": 1.5e-4, "bf16": 7e-2}


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[c[-1] for c in CASES])
def test_conv_gemm(case, precision):
    from tests import ops_binding as ops
    R, C, N, k, has_bias, has_res, relu_pre, ln_eps, act, has_dot, _ = case
    rs = np.random.RandomState(R + C + N + k)
    dev = _dev()
    x = _rand(rs, R, C)
    w = _rand(rs, N, C, k, scale=1.0 / np.sqrt(C * k)) if k > 1 else _rand(rs, N, C, scale=1.0 / np.sqrt(C))
    bias = _rand(rs, N, scale=0.5) if has_bias else None
    resid = _rand(rs, R, N) if has_res else None
    g, bt = (1.0 + _rand(rs, N, scale=0.2), _rand(rs, N, scale=0.2)) if ln_eps is not None else (None, None)
    dw, db = (_rand(rs, N, scale=0.1), _rand(rs, 1)) if has_dot else (None, None)
    valid = torch.from_numpy((rs.uniform(size=R) > 0.1).astype(np.int32))
    # reference
    y = _ref_conv(x, w, bias)
    if resid is not None:
        y = y + resid
    if relu_pre:
        y = torch.relu(y)
    if ln_eps is not None:
        y = F.layer_norm(y, (N,), g, bt, ln_eps)
    y = torch.relu(y) if act == 1 else (torch.tanh(y) if act == 2 else y)
    d = (y @ dw + db) if has_dot else None
    y = y * valid.unsqueeze(1)
    to = lambda t: t.to(dev) if t is not None else None
    yo, do = ops.conv_gemm(to(x), to(w), to(bias), to(resid), relu_pre, (to(g), to(bt)) if g is not None else None,
                           ln_eps or 1e-5, act, (to(dw), to(db)) if has_dot else None, to(valid), precision=precision)
    tol = GEMM_TOL[precision]
    if not (has_dot and precision != "fp32"):      # two-pass bf16 path keeps its dot-only result in scratch
        err = float((yo.cpu() - y).abs().max())
        assert torch.isfinite(yo).all(), "non-finite / unwritten output"
        print("%s max-abs %.2e" % (precision, err))
        from tests.conftest import record_measurement
        record_measurement("gemm_%s" % precision, err)
        assert err < tol, "max-abs %g" % err
    if has_dot:
        derr = float(((do.cpu() - d) * valid).abs().max())
        assert derr < tol, "dot head max-abs %g" % derr


@pytest.mark.parametrize("case", [c for c in CASES if c[7] is not None], ids=[c[-1] for c in CASES if c[7] is not None])
def test_conv_gemm_fp32_row_complete_kernel(case, fs2_option):
    """The row-complete fp32 GEMM (LayerNorm inside the MFMA epilogue) is no longer the default for N >= 128;
    keep it covered."""
    fs2_option("FS2_F32_ROWS", 1)
    test_conv_gemm(case, "fp32")


@pytest.mark.parametrize("bm", ["64", "128", "160", "192", "224", "256"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[c[-1] for c in CASES])
def test_conv_gemm_planes_kernel_tile_heights(case, precision, bm, fs2_option):
    """The default bf16 path (activations as split-bf16 planes, both operands by LDS-DMA: gemm_planes.h) at every
    tile height (256 rows exists for the conv form only; k = 1 GEMMs fall back to their own choice)."""
    fs2_option("FS2_BM", bm)
    test_conv_gemm(case, precision)


ROW8_CASES = [c for c in CASES if c[7] is not None and c[2] in (256, 384)]      # k = 1 (gemm_row8_bf16) and conv form (gemm_row8c_bf16, incl. the scalar head)


@pytest.mark.parametrize("mt8", [2, 3])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", ROW8_CASES, ids=[c[-1] for c in ROW8_CASES])
def test_conv_gemm_row_complete_ln_fused_kernel(case, precision, mt8, fs2_option):
    """gemm_row8_bf16 / gemm_row8c_bf16 (64 MT rows x all N columns per workgroup, ReLU / LayerNorm / activation / scalar head in the
    epilogue; k = 1 and k-tap conv form) are chosen by size in the model path; force them here on the small op cases (several row
    tiles, ragged last tile, gap rows) at every tile height (MT = 2 / 3: 128 / 192 rows)."""
    fs2_option("FS2_ROW8", 1)
    fs2_option("FS2_MT8", mt8)
    test_conv_gemm(case, precision)


F16_CASES = [c for c in CASES if c[3] > 1 and c[7] is None and not c[5]]      # plain convolutions (bias / activation only)
F16_TOL = {"mix_f16x2": 2e-3, "mix_f16x1": 3e-3}       # one / both operands rounded to fp16 once (11 bits): ~2^-12 relative per product


@pytest.mark.parametrize("bm", ["64", "128", "256"])
@pytest.mark.parametrize("precision", ["mix_f16x2", "mix_f16x1"])
@pytest.mark.parametrize("case", F16_CASES, ids=[c[-1] for c in F16_CASES])
def test_conv_gemm_fp16_two_and_one_term(case, precision, bm, fs2_option):
    """The fp16-operand forms of the conv kernel (FFN w_1 in the mixed modes): activations hi + lo, weights rounded once (2 MFMAs
    per fragment pair) / both rounded once (1 MFMA), at every tile height."""
    fs2_option("FS2_BM", bm)
    from tests import ops_binding as ops
    from tests.conftest import record_measurement
    R, C, N, k, has_bias, _, _, _, act, _, _ = case
    rs = np.random.RandomState(R + C + N + k)
    dev = _dev()
    x = _rand(rs, R, C)
    w = _rand(rs, N, C, k, scale=1.0 / np.sqrt(C * k))
    bias = _rand(rs, N, scale=0.5)
    y = _ref_conv(x, w, bias)
    y = torch.relu(y) if act == 1 else (torch.tanh(y) if act == 2 else y)
    yo, _ = ops.conv_gemm(x.to(dev), w.to(dev), bias.to(dev), None, False, None, 1e-5, act, None, None, precision=precision)
    err = float((yo.cpu() - y).abs().max())
    print("%s max-abs %.2e" % (precision, err))
    record_measurement("gemm_%s" % precision, err)
    assert torch.isfinite(yo).all() and err < F16_TOL[precision]


@pytest.mark.parametrize("bm", ["64", "128", "160", "192", "224", "256"])
@pytest.mark.parametrize("shape", [(300, 384, 1024, 9), (517, 256, 1024, 9), (90, 384, 128, 9), (260, 256, 256, 3), (150, 128, 128, 5)])
def test_conv_gemm_mx_fp16_plus_block_scaled_fp8(shape, bm, fs2_option):
    """The "mx" arithmetic (gemm_mx.h; gemm_pl_bf16<.., ARITH = 2> on mx planes and the mx weight image): a.w = ah.wh (fp16 MFMA) +
    ra.wh + ah.rw (v_mfma_scale_f32_16x16x128_f8f6f4 on e4m3 operands with static scales, K = 128 channels of one tap): two
    MFMA-equivalents per product, any odd kernel size, C % 128 == 0.  Accuracy class of split-bf16 (tolerance: 5 x measured)."""
    fs2_option("FS2_BM", bm)
    from tests import ops_binding as ops
    from tests.conftest import record_measurement
    R, C, N, k = shape
    rs = np.random.RandomState(R + C + N + k)
    dev = _dev()
    x = _rand(rs, R, C) * torch.from_numpy(rs.uniform(0.2, 2.0, size=(1, C)).astype(np.float32))      # per-channel gains, LayerNorm-like
    w = _rand(rs, N, C, k, scale=1.0 / np.sqrt(C * k))
    bias = _rand(rs, N, scale=0.5)
    y = torch.relu(_ref_conv(x.double(), w.double(), bias.double())).float()
    yo, _ = ops.conv_gemm(x.to(dev), w.to(dev), bias.to(dev), None, False, None, 1e-5, 1, None, None, precision="mix_mx")
    err = float((yo.cpu() - y).abs().max())
    print("mix_mx R=%d C=%d N=%d k=%d BM=%s max-abs %.2e" % (R, C, N, k, bm, err))
    record_measurement("gemm_mix_mx", err)
    assert torch.isfinite(yo).all() and err < 2.5e-4


def test_conv_gemm_transpose_detecting():
    """A = identity-like with an ASYMMETRIC weight: catches a swapped C/D row<->col mapping."""
    from tests import ops_binding as ops
    dev = _dev()
    R = C = N = 256
    x = torch.eye(R)
    w = torch.arange(N * C, dtype=torch.float32).view(N, C) / (N * C)
    y, _ = ops.conv_gemm(x.to(dev), w.to(dev))
    assert torch.allclose(y.cpu(), w.t(), atol=1e-6)


# 5 x measured (round 2: fp32 9.7e-7, bf16x3 1.4e-5, bf16 6.7e-3)
ATT_TOL = {"fp32": 5e-6, "bf16x3": 7e-5, "bf16": 3.5e-2}


# "bf16x3/w32": the same arithmetic through attn_w32 (32 queries per wave; forced with FS2_ATTN_W32 = 1: the automatic choice takes it
# only for grids that fill the chip); "bf16x3" pins the 64-query kernel (FS2_ATTN_W32 = 0)
# ".../planes": the kernels hand the context back as split-bf16 planes only, the form the model uses (attn_w32: the LDS-staged epilogue),
# and the operator converts (FS2_OP_ATT_PLANES); hi + lo carries 16 mantissa bits: ~1.5e-5 on |ctx| <= 2
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16x3/w32", "bf16x3/planes", "bf16x3/w32/planes", "bf16"])
@pytest.mark.parametrize("D,heads", [(256, 2), (384, 2)])
@pytest.mark.parametrize("mask_q", [0, 1])
def test_attention(D, heads, mask_q, precision, fs2_option):
    from tests import ops_binding as ops
    dev = _dev()
    if precision.startswith("bf16x3"):
        fs2_option("FS2_ATTN_W32", 1 if "/w32" in precision else 0)
        fs2_option("FS2_OP_ATT_PLANES", 1 if precision.endswith("/planes") else 0)
        precision = "bf16x3"
    rs = np.random.RandomState(D + mask_q)
    # (lengths around the tile sizes of both kernels: 32-key tiles, 64- and 128-query blocks; a 1-frame utterance; klen % 8 != 0)
    lens = [70, 1, 33, 200, 64, 129, 261]
    klens = [70, 1, 20, 150, 64, 128, 255] if mask_q else lens
    starts, row = [], 32
    for l in lens:
        starts.append(row)
        row = (row + l + 8 + 7) // 8 * 8          # utterance starts are 8-row aligned (kAttAlign): 16-byte V^T loads
    qkv = _rand(rs, row + 64, 3 * D, scale=2.0)
    # Keys beyond klen must not matter WHATEVER their rows hold: every K / V row that is not a live key (masked pad keys, the gap
    # rows, the rows behind the last utterance) is NaN.  0 * NaN would poison P.V if the kernel let a dead key through.
    live = torch.zeros(qkv.shape[0], dtype=torch.bool)
    for s, kl in zip(starts, klens):
        live[s:s + kl] = True
    qkv[~live, D:] = float("nan")
    ctx = ops.attention(qkv.to(dev), D, heads, starts, lens, klens, mask_q, precision=precision).cpu()
    assert torch.isfinite(ctx[live]).all(), "a dead key reached the output"
    dk = D // heads
    worst = 0.0
    for s, l, kl in zip(starts, lens, klens):
        q = qkv[s:s + l, :D].view(l, heads, dk).transpose(0, 1)
        k, v = (qkv[s:s + kl, i * D:(i + 1) * D].view(kl, heads, dk).transpose(0, 1) for i in (1, 2))
        sc = q @ k.transpose(1, 2) / np.sqrt(dk)
        a = torch.softmax(sc, dim=-1)
        o = (a @ v).transpose(0, 1).reshape(l, D)
        if mask_q:
            o[kl:] = 0.0
        worst = max(worst, float((ctx[s:s + l] - o).abs().max()))
    print("attention %s max-abs %.2e" % (precision, worst))
    from tests.conftest import record_measurement
    record_measurement("attention_%s" % precision, worst)
    assert worst < ATT_TOL[precision], "max-abs %g" % worst


@pytest.mark.parametrize("spike", [40.0, 400.0])
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16x3/w32", "bf16x3/w32/planes"])
def test_attention_spiked_key_forces_rescale(precision, spike, fs2_option):
    """One key far above the others late in the sequence: exercises the online-softmax rescale branch of the 64-query kernels and, in
    attn_w32 (which never rescales: its reference maximum is the first tile's, attn_w32.h), probabilities far above 1 (rows whose
    scores jump by less than 2^60 over their first tile's maximum) and the wave's exit to the plain fp32 row loop (beyond: spike 40
    sends some rows there, spike 400 nearly all)."""
    from tests import ops_binding as ops
    dev = _dev()
    if precision.startswith("bf16x3"):
        fs2_option("FS2_ATTN_W32", 1 if "/w32" in precision else 0)
        fs2_option("FS2_OP_ATT_PLANES", 1 if precision.endswith("/planes") else 0)
        precision = "bf16x3"
    rs = np.random.RandomState(5)
    D, heads, l = 256, 2, 130
    qkv = _rand(rs, 32 + l + 16, 3 * D, scale=0.5)
    qkv[32 + 97, D:2 * D] = qkv[32 + 3, 0:D] * spike      # key 97 aligned with query 3
    ctx = ops.attention(qkv.to(dev), D, heads, [32], [l], [l], 0, precision=precision).cpu()
    dk = D // heads
    q, k, v = (qkv[32:32 + l, i * D:(i + 1) * D].double().view(l, heads, dk).transpose(0, 1) for i in range(3))
    o = (torch.softmax(q @ k.transpose(1, 2) / np.sqrt(dk), -1) @ v).transpose(0, 1).reshape(l, D)
    err = float((ctx[32:32 + l].double() - o).abs().max())
    print("spiked attention %s max-abs %.2e" % (precision, err))
    assert err < (2e-5 if precision == "fp32" else 5e-4)


def test_length_regulator_known_answers(golden_dir):
    from tests import ops_binding as ops
    dev = _dev()
    g = np.load(golden_dir + "/g4_known_answers.npz")
    hs, ds, il, want = (torch.from_numpy(g[k]) for k in ("lr_hs", "lr_ds", "lr_ilens", "lr_out"))
    out, idx, olens = ops.length_regulate(hs.to(dev), ds.to(dev), il.tolist(), want.shape[1])
    assert torch.equal(out.cpu(), want)          # bit-exact copy semantics
    assert olens.cpu().tolist() == [7, 4, 2]     # all-zero row -> ones; zeros inside skipped
    assert idx.cpu()[0].tolist() == [0, 1, 1, 2, 2, 2, 4]


def test_length_regulator_random_bit_exact():
    from tests import ops_binding as ops
    from oracle import fs2_oracle as O
    dev = _dev()
    rs = np.random.RandomState(9)
    B, Tmax, D = 9, 300, 256
    il = rs.randint(1, Tmax + 1, size=B)
    il[0] = Tmax
    ds = torch.from_numpy(rs.randint(0, 12, size=(B, Tmax)).astype(np.int64))
    ds[3, : il[3]] = 0                                    # all-zero utterance
    hs = _rand(rs, B, Tmax, D)
    want, olens, idxs = O.length_regulate(hs, ds, torch.from_numpy(il))
    out, idx, ol = ops.length_regulate(hs.to(dev), ds.to(dev), il.tolist(), want.shape[1] + 5)
    assert torch.equal(ol.cpu(), olens)
    assert torch.equal(out.cpu()[:, : want.shape[1]], want)
    assert float(out.cpu()[:, want.shape[1]:].abs().max()) == 0.0
    for b in range(B):
        assert torch.equal(idx[b, : olens[b]].cpu().long(), idxs[b])
        assert (idx[b, olens[b]:] == -1).all()


def test_bucketize(golden_dir):
    from tests import ops_binding as ops
    dev = _dev()
    g = np.load(golden_dir + "/g4_known_answers.npz")
    for xk, qk, bk in (("xe", "qe", "energy_bins"), ("xp", "qp", "pitch_bins")):
        got = ops.bucketize(torch.from_numpy(g[xk]).to(dev), torch.from_numpy(g[bk]).to(dev)).cpu().long()
        assert got.tolist() == g[qk].tolist()
    rs = np.random.RandomState(3)
    bins = torch.from_numpy(g["pitch_bins"])
    x = torch.from_numpy(rs.uniform(0, 800, size=100000).astype(np.float32))
    x[::7] = bins[rs.randint(0, 255, size=x[::7].numel())]     # exact boundary hits
    got = ops.bucketize(x.to(dev), bins.to(dev)).cpu().long()
    assert torch.equal(got, torch.bucketize(x, bins))


def _dur_candidates(y, ulps=2):
    """Durations that a correctly implemented post-op may return for log-durations y: exp() is allowed to be `ulps` fp32 ulps
    off the exact value (torch's CPU exp and HIP's expf are both ~1-ulp functions), the rounding itself must be
    round-half-to-even followed by clamp(min=0) (reference duration_predictor.py:77-81)."""
    e = torch.exp(y.double()).float()
    out = []
    for k in range(-ulps, ulps + 1):
        v = e.clone()
        for _ in range(abs(k)):
            v = torch.nextafter(v, torch.full_like(v, float("inf") if k > 0 else float("-inf")))
        out.append(torch.clamp(torch.round(v - 1.0), min=0).long())
    return torch.stack(out)


def test_duration_postop_known_answers_and_ties(golden_dir):
    """fs2_op_duration = the HIP `clamp(round(exp(y) - 1), 0).long()` of the duration predictor (elementwise.h:
    duration_from_log, also used by dur_finalize inside fs2_encode).  G4 known answers from the reference (0.5 -> 0, 1.5 -> 2,
    2.5 -> 2, 3.5 -> 4: round half to even), 1e5 random values against the CPU oracle, and every exact .5 tie up to 40 frames."""
    from tests import ops_binding as ops
    from oracle import fs2_oracle as O
    dev = _dev()
    g = np.load(golden_dir + "/g4_known_answers.npz")
    got = ops.duration(torch.from_numpy(g["dur_log"]).to(dev)).cpu()
    assert got.tolist() == g["dur_int"].tolist(), (got.tolist(), g["dur_int"].tolist())
    rs = np.random.RandomState(11)
    y = torch.from_numpy(rs.uniform(-3.0, 4.5, size=100000).astype(np.float32))
    got = ops.duration(y.to(dev)).cpu()
    want = O.duration_from_log(y)
    cand = _dur_candidates(y)
    assert bool((cand == got.unsqueeze(0)).any(0).all()), "a duration outside what a 2-ulp exp allows"
    edge = (cand != cand[0:1]).any(0)                     # values within 2 ulps of a rounding boundary
    assert torch.equal(got[~edge], want[~edge])           # everywhere else: bit-exact against the oracle
    agree = float((got == want).float().mean())
    print("duration post-op: %d of 100000 values sit within 2 ulp of a rounding boundary; exact agreement with the oracle %.6f"
          % (int(edge.sum()), agree))
    assert agree > 0.9999
    # exact ties: exp(y) - 1 == k + 0.5 in exact arithmetic -> round half to even picks the even neighbour
    k = torch.arange(0, 41, dtype=torch.float64)
    yt = torch.log(k + 1.5).float()
    got = ops.duration(yt.to(dev)).cpu()
    even = (2 * torch.round((k + 0.5) / 2)).long()        # the even neighbour of k + 0.5
    cand = _dur_candidates(yt)
    assert bool((cand == got.unsqueeze(0)).any(0).all())
    exact_tie = (torch.exp(yt.double()).float() - 1.0).double() == (k + 0.5)      # fp32 exp lands on the tie exactly
    print("ties: %d / 41 resolved to the even neighbour (%d are exact ties in fp32)" % (int((got == even).sum()), int(exact_tie.sum())))
    # special values
    sp = torch.tensor([float("-inf"), -100.0, 0.0, 40.0, 100.0, float("inf"), float("nan")])
    got = ops.duration(sp.to(dev)).cpu().tolist()
    assert got[:3] == [0, 0, 0] and got[6] == 0
    assert abs(got[3] - float(torch.tensor(40.0).double().exp())) < 2.0 ** 35      # e^40 = 2.35e17: fp32 spacing there is 2^34
    assert got[4] == got[5] == 2 ** 63 - 1                # saturates like .long() of +inf does not: documented (fs2.h)


@pytest.mark.parametrize("alpha", [0.5, 1.3, 2.0])
def test_length_regulator_alpha(alpha):
    """Speed control of the length regulator (reference length_regulator.py:57-59): ds <- round(ds.float() * alpha)."""
    from tests import ops_binding as ops
    from oracle import fs2_oracle as O
    dev = _dev()
    rs = np.random.RandomState(int(alpha * 10))
    B, Tmax, D = 5, 90, 64
    il = rs.randint(1, Tmax + 1, size=B)
    ds = torch.from_numpy(rs.randint(0, 12, size=(B, Tmax)).astype(np.int64))
    ds[2, : il[2]] = 0
    hs = _rand(rs, B, Tmax, D)
    ds_a = torch.round(ds.float() * alpha).long()
    want, olens, idxs = O.length_regulate(hs, ds_a, torch.from_numpy(il))
    out, idx, ol = ops.length_regulate(hs.to(dev), ds.to(dev), il.tolist(), want.shape[1], alpha=alpha)
    assert torch.equal(ol.cpu(), olens) and torch.equal(out.cpu(), want)
    for b in range(B):
        assert torch.equal(idx[b, : olens[b]].cpu().long(), idxs[b])
