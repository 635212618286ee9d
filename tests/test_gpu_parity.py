"""End-to-end parity of the HIP path (through FeedForwardTransformer -> C ABI) against
 (1) the golden vectors captured from the real reference (tests/golden, oracle/gen_golden.py),
 (2) the CPU oracle on the same seeded inputs, and
 (3) size-independent properties at BASELINE.json's full sizes.
Bar (BASELINE.json north_star): mel max-abs <= 1e-3 vs the reference CPU path; length-regulator indices
and (teacher-forced) bucket indices bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
MEL_TOL = 1e-3


@pytest.fixture(scope="module")
def env():
    from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
    from fastspeech2_amd.synthetic import portable_state_dict
    from oracle import fs2_oracle as O
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    hp = default_hparams()
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
    sd = portable_state_dict(model.state_dict(), seed=0)
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, hp.audio.num_mels)
    return model, sd, cfg, O


def _t(a, dev="cuda:0"):
    return torch.from_numpy(np.asarray(a)).to(dev)


def _maxabs(a, b):
    return float((a.detach().cpu().float() - torch.as_tensor(b).float()).abs().max())


# the reference's own fixtures in every mode that ships (until round 5 they ran in the module default, fp32, only; the mode
# bench.py runs reached the reference through the pinned oracle alone)
SHIPPING_MODES = ("fp32", "bf16x3", "mix_mx")


@pytest.fixture
def in_mode(env):
    model = env[0]

    def set_(precision):
        model.precision = precision
    yield set_
    model.precision = "fp32"


@pytest.mark.parametrize("precision", SHIPPING_MODES)
def test_g1_teacher_forced_single(env, golden_dir, in_mode, precision):
    model, sd, cfg, O = env
    g = np.load(golden_dir + "/g1_teacher_b1.npz")
    in_mode(precision)
    with torch.no_grad():
        r = model._run(_t(g["xs"]), g["ilens"], g["olens"], _t(g["ds"]), _t(g["es"]), _t(g["ps"]), is_inference=False,
                       want=("before", "after", "e_outs", "p_outs", "qe", "qp", "lr_index", "encoder_out", "decoder_out"))
    L = int(g["olens"][0])
    assert r["lr_index"][0, :L].cpu().tolist() == g["lr_index"].tolist()          # bit-exact
    assert r["qe"][0].cpu().tolist() == g["qe"][0].tolist() and r["qp"][0].cpu().tolist() == g["qp"][0].tolist()
    d = {k: _maxabs(r[k], g[k]) for k in ("before", "after", "e_outs", "p_outs", "encoder_out")}
    d["d_outs"] = _maxabs(r["d_log"], g["d_outs"])
    d["decoder_rows"] = _maxabs(r["decoder_out"][0][torch.as_tensor(g["decoder_rows"])], g["decoder_out_rows"])
    print("G1 [%s] max-abs:" % precision, {k: "%.2e" % v for k, v in d.items()})
    assert max(d.values()) <= MEL_TOL, d


def test_g2_padded_compat_and_losses(env, golden_dir):
    model, sd, cfg, O = env
    g = np.load(golden_dir + "/g2_teacher_padded_b3.npz")
    model.batch_semantics = "padded_compat"
    try:
        with torch.no_grad():
            out = model._forward(_t(g["xs"]), _t(g["ilens"]), _t(g["olens"]), _t(g["ds"]), _t(g["es"]), _t(g["ps"]))
            loss, rep = model(_t(g["xs"]), _t(g["ilens"]), _t(g["ys"]), _t(g["olens"]), _t(g["ds"]), _t(g["es"]), _t(g["ps"]))
    finally:
        model.batch_semantics = "per_utterance"
    d = {k: _maxabs(o, g[k]) for k, o in zip(("before", "after", "d_outs", "e_outs", "p_outs"), out)}
    print("G2 max-abs:", {k: "%.2e" % v for k, v in d.items()})
    assert max(d.values()) <= MEL_TOL, d
    assert [list(x.keys())[0] for x in rep] == g["report_names"].tolist()
    got = np.array([list(x.values())[0] for x in rep])
    assert np.allclose(got, g["report_values"], rtol=1e-4), (got, g["report_values"])
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))


def test_g9_weighted_masking_losses_on_device(golden_dir):
    """forward() with hp.model.use_weighted_masking = True, use_masking = False against the REAL reference's loss and report values
    (fixture G9, fastspeech.py:308-333); with both flags the reference itself raises IndexError (ys is 1-D after masked_select) and
    so does this module."""
    from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
    from fastspeech2_amd.synthetic import portable_state_dict
    g = np.load(golden_dir + "/g9_weighted_masking_b3.npz")
    hp = default_hparams()
    hp.model.use_masking, hp.model.use_weighted_masking = False, True
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, 80, hp).eval()
    model.load_state_dict(portable_state_dict(model.state_dict(), seed=0))
    model = model.to("cuda:0")
    with torch.no_grad():
        loss, rep = model(_t(g["xs"]), _t(g["ilens"]), _t(g["ys"]), _t(g["olens"]), _t(g["ds"]), _t(g["es"]), _t(g["ps"]))
    assert [list(x.keys())[0] for x in rep] == g["report_names"].tolist()
    got = np.array([list(x.values())[0] for x in rep])
    assert np.allclose(got, g["report_values"], rtol=1e-4), (got, g["report_values"])
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    model.use_masking = True
    with pytest.raises(IndexError):
        with torch.no_grad():
            model(_t(g["xs"]), _t(g["ilens"]), _t(g["ys"]), _t(g["olens"]), _t(g["ds"]), _t(g["es"]), _t(g["ps"]))


@pytest.mark.parametrize("precision", SHIPPING_MODES)
def test_g6_per_utterance_semantics_in_a_batch(env, golden_dir, in_mode, precision):
    """Default semantics: every utterance of a padded batch comes out as if it had been run alone."""
    model, sd, cfg, O = env
    g2 = np.load(golden_dir + "/g2_teacher_padded_b3.npz")
    g6 = np.load(golden_dir + "/g6_teacher_per_utt_b3.npz")
    in_mode(precision)
    with torch.no_grad():
        before, after, *_ = model._forward(_t(g2["xs"]), _t(g2["ilens"]), _t(g2["olens"]), _t(g2["ds"]), _t(g2["es"]), _t(g2["ps"]))
    for b in range(3):
        L = int(g2["olens"][b])
        assert _maxabs(after[b, :L], g6["after_%d" % b]) <= MEL_TOL
        assert _maxabs(before[b, :L], g6["before_%d" % b]) <= MEL_TOL
        assert float(after[b, L:].abs().max()) == 0.0 if L < after.shape[1] else True


@pytest.mark.parametrize("precision", SHIPPING_MODES)
def test_g3_free_running_inference(env, golden_dir, in_mode, precision):
    """The reference's `inference()` (fastspeech.py:339-357) on fixture G3, free-running in durations, pitch and energy: every integer decision
    (24 durations; 102 + 102 bucket indices) equals the reference's in every shipping mode, the mel within the tolerance."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import bias_durations
    g = np.load(golden_dir + "/g3_inference_t24.npz")
    in_mode(precision)
    model.load_state_dict(bias_durations(sd, 4.0))
    try:
        with torch.no_grad():
            mel = model.inference(_t(g["x"]))
            before, after, d_outs, oe, op = model._forward(_t(g["x"]).unsqueeze(0), torch.tensor([24]), is_inference=True)
    finally:
        model.load_state_dict(sd)
    assert d_outs.dtype == torch.int64 and d_outs[0].cpu().tolist() == g["d_outs"][0].tolist()
    assert mel.shape == tuple(g["mel"].shape)
    assert oe.shape == (1, mel.shape[0], 256) and float(oe.sum()) == mel.shape[0]
    agree = float((oe.argmax(-1)[0].cpu() == torch.as_tensor(g["qe"][0])).float().mean())
    print("G3 [%s] mel max-abs %.2e, energy-code agreement %.3f" % (precision, _maxabs(mel, g["mel"]), agree))
    assert _maxabs(mel, g["mel"]) <= MEL_TOL and _maxabs(after[0], g["mel"]) <= MEL_TOL


# Factor on the decoder's Q and K projections (weights and biases): the attention logits grow 49-fold.  Measured on MI355X, c2 teacher-forced
# (slow-path waves of one forward / mel max-abs vs the oracle in mix_mx, bf16x3, fp32): x5: 0 / 6.9e-5, 6.3e-5, 1.3e-5; x6: 49 / 2.5e-4, 1.5e-4, 1.6e-5;
# x7: 640 / 5.1e-4, 3.8e-4, 4.2e-5; x8: 1,584 / 9.2e-4, 9.3e-4, 9.9e-5; x16: 3,127 / 4.1e-2, 3.5e-2, 4.9e-3 -- beyond x8 the softmax is an argmax between
# near-equal logits and even fp32 on the GPU leaves the fp32 CPU result.  The growth in the split-bf16 modes is the operands' (Q and K at 16-17
# significant bits under logits of several hundred), not the slow path's: the 64-query kernel, which has none, lands on the same numbers.
PEAK_SCALE = 7.0


def test_peaked_decoder_attention_takes_the_slow_path_and_keeps_parity(env, fs2_option):
    """attn_w32 runs the softmax relative to the FIRST key tile's maximum and sends a wave whose probabilities then sum beyond 2^60 to a plain fp32
    two-pass loop (attn_w32_rows_slow).  Random-init weights give flat attention, so no benchmark or model-level test ever took that exit (VERDICT r05
    item 6) while a trained model's peaked rows would.  Here the decoder's Q / K projections are scaled so that its attention IS peaked: the whole
    forward (c2 batch, teacher-forced, mix_mx: decoder attention on attn_w32) must match the oracle on the same weights, the handle's counter
    `attn_slow_path_waves` must have moved, and the per-call status word of a sync-free call must report the same exit (reference
    core/attention.py:55-62: one softmax formula for every row)."""
    from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
    from fastspeech2_amd.synthetic import make_batch
    from tests.conftest import record_measurement
    _, sd, cfg, O = env
    sd2 = {k: v.clone() for k, v in sd.items()}
    for k in sd2:
        if k.startswith("decoder.encoders_.") and (".self_attn.linear_q." in k or ".self_attn.linear_k." in k):
            sd2[k] = sd2[k] * PEAK_SCALE
    hp = default_hparams()
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
    model.load_state_dict(sd2)
    model = model.to("cuda:0")
    model.precision = "mix_mx"
    b = make_batch("c2")
    with torch.no_grad():
        r = model._run(b["xs"].cuda(), b["ilens"], b["olens"], b["ds"].cuda(), b["es"].cuda(), b["ps"].cuda(), is_inference=False, want=("after",))
        slow = model.counter("attn_slow_path_waves", reset=True)
        # the same forward on the 64-query kernel (running maximum, no slow path): what separates the exit's arithmetic from the operands' precision
        fs2_option("FS2_ATTN_W32", 0)
        r64 = model._run(b["xs"].cuda(), b["ilens"], b["olens"], b["ds"].cuda(), b["es"].cuda(), b["ps"].cuda(), is_inference=False, want=("after",))
        assert model.counter("attn_slow_path_waves", reset=True) == 0
        fs2_option("FS2_ATTN_W32", -1)
        # the flat model of the other tests on the same batch: no wave leaves the fast path
        flat = env[0]
        flat.precision = "mix_mx"
        try:
            flat._run(b["xs"].cuda(), b["ilens"], b["olens"], b["ds"].cuda(), b["es"].cuda(), b["ps"].cuda(), is_inference=False, want=("after",))
            slow_flat = flat.counter("attn_slow_path_waves", reset=True)
        finally:
            flat.precision = "fp32"
        # sync-free call: the same count in this call's own status word
        model.inference_batch(b["xs"].cuda(), b["ilens"], d_override=b["ds"].cuda())
        model.counter("attn_slow_path_waves", reset=True)
        res = model.inference_batch(b["xs"].cuda(), b["ilens"], d_override=b["ds"].cuda(), sync=False)
        st = res.status.cpu()
        assert res.ok()
        slow_call = model.counter("attn_slow_path_waves")
    o = O.per_utterance_forward(sd2, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])
    worst, worst64, between = 0.0, 0.0, 0.0
    for i in range(b["xs"].shape[0]):
        L = int(b["olens"][i])
        worst = max(worst, _maxabs(r["after"][i, :L], o["after"][i, :L]))
        worst64 = max(worst64, _maxabs(r64["after"][i, :L], o["after"][i, :L]))
        between = max(between, _maxabs(r["after"][i, :L], r64["after"][i, :L].cpu()))
    print("peaked decoder attention (Q, K x %.0f): %d wave(s) took attn_w32's slow path in one teacher-forced c2 forward (flat model: %d), %d in one sync-free "
          "call (its status word: %d); mel max-abs vs the oracle %.2e (the 64-query kernel, which has no such exit: %.2e; the two kernels apart by %.2e)"
          % (PEAK_SCALE, slow, slow_flat, slow_call, int(st[5]), worst, worst64, between))
    record_measurement("c2_peaked_attention_mel_maxabs_mix_mx", worst)
    record_measurement("c2_peaked_attention_mel_maxabs_mix_mx_attn_bf16", worst64)
    record_measurement("c2_peaked_attention_slow_path_waves", slow)
    assert worst64 <= MEL_TOL and between <= MEL_TOL, (worst64, between)
    assert slow > 0, "the scaled model's attention did not leave the fast path: raise PEAK_SCALE"
    assert slow_flat == 0
    assert int(st[5]) == slow_call > 0
    assert worst <= MEL_TOL, worst


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
def test_c2_batch_vs_oracle(env, precision):
    """BASELINE config c2 (B=16, T in 64..128), teacher-forced, per-utterance semantics, in every arithmetic
    mode.  fp32 and bf16x3 are parity modes (<= 1e-3); plain bf16 is measured and reported only (it cannot
    meet the tolerance: BASELINE.md section 2)."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c2")
    model.precision = precision
    try:
        _c2_body(model, sd, cfg, O, b, precision)
    finally:
        model.precision = "fp32"


@pytest.mark.parametrize("precision", ["mix_f16x2", "mix_f16x1", "mix_mx"])
def test_c2_mixed_modes_vs_oracle(env, precision):
    """The mixed arithmetic modes (bf16x3 everywhere except the FFN convolution w_1, which runs on fp16 operands with 2 / 1 MFMAs
    per fragment pair): c2 against the oracle within the 1e-3 mel tolerance; integer decisions still exact.  Measured errors
    and speed-ups: BASELINE.md section 4 (they are NOT the default: they use up a third to a half of the tolerance)."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    from tests.conftest import record_measurement
    model.precision = precision
    try:
        b = make_batch("c2")
        with torch.no_grad():
            r = model._run(b["xs"].cuda(), b["ilens"], b["olens"], b["ds"].cuda(), b["es"].cuda(), b["ps"].cuda(),
                           is_inference=False, want=("before", "after", "lr_index"))
        o = O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])
        for i in range(b["xs"].shape[0]):
            L = int(o["olens"][i])
            assert torch.equal(r["lr_index"][i, :L].cpu().long(), o["lr_index"][i])
        d = max(_maxabs(r["before"], o["before"]), _maxabs(r["after"], o["after"]))
        print("c2 [%s] mel max-abs vs oracle %.2e" % (precision, d))
        record_measurement("c2_mel_maxabs_" + precision, d)
        assert d <= (3e-5 if precision == "mix_mx" else MEL_TOL)      # mix_mx is a parity-grade mode: split-bf16 class (2.3e-5), measured 2.5e-5
    finally:
        model.precision = "fp32"


@pytest.mark.parametrize("row8", ["0", "1"])
def test_c2_row_complete_kernel_choice(env, row8, fs2_option):
    """The LN-terminated k = 1 GEMMs have two bf16 implementations (128-column tiles + row kernel / row-complete tile with
    the LayerNorm fused, chosen by size): c2 in the parity mode with each one forced."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    fs2_option("FS2_ROW8", row8)
    model.precision = "bf16x3"
    try:
        _c2_body(model, sd, cfg, O, make_batch("c2"), "bf16x3")
    finally:
        model.precision = "fp32"


def test_c2_row_complete_tile_heights_are_bit_identical(env, fs2_option):
    """The 8-wave row-complete kernels (LayerNorm-fused k = 1 GEMMs, fused QKV, predictor convolutions) take 128 or 192 rows per
    workgroup, and the fused QKV projection runs its three passes in one workgroup or in three (FS2_QKV_SPLIT), whichever avoids a
    nearly empty last round: the choices must not change a single bit."""
    model = env[0]
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c2")
    fs2_option("FS2_ROW8", 1)
    fs2_option("FS2_QKV8", 1)
    model.precision = "bf16x3"
    try:
        outs = []
        for mt, split in ((2, 0), (3, 0), (2, 1), (3, 1)):
            fs2_option("FS2_MT8", mt)
            fs2_option("FS2_QKV_SPLIT", split)
            with torch.no_grad():
                outs.append(model.inference_batch(b["xs"].cuda(), b["ilens"], d_override=b["ds"].cuda())[0])
        assert all(torch.equal(outs[0], o) for o in outs[1:])
    finally:
        model.precision = "fp32"


def test_c2_one_wave_per_simd_row_kernel_against_the_8_wave_one(env, fs2_option):
    """gemm_row4_bf16 (one wave per SIMD, accumulators in literal AGPRs; the decoder's out-proj + LN1, FFN2 + LN2 and input layer) keeps
    gemm_row8_bf16's operands, layouts and per-accumulator MFMA order, and since round 6 runs the PLANES-ONLY form of these launches: no fp32 rows
    out, the residual read from the producing launch's planes (16-17 significant bits of the fp32 row, ~15 from mx planes).  With the row-complete
    kernels forced at c2: the forward at both tile heights (FS2_MT4 = 4 | 5: 128 / 160 rows) must not differ in a single bit, and must stay within
    1e-4 of the forward with FS2_ROW4 = 0 (gemm_row8_bf16: fp32 residual) -- in split-bf16 and, with FFN2 kept on split-bf16 (FS2_FFN2_MX = 0), in
    mix_mx (out-proj epilogue writes mx planes: EPI 1; FFN2's residual comes out of them: RES 2).  Bit-identity of the kernel itself against
    gemm_row8_bf16 on the same (reconstructed) residual is tools/probes/row_probe.hip's job (profiles/r06_row_probe.txt: 0 words differ in all
    eight forms).  With FFN2 + LN2 in the mx arithmetic (gemm_row4.h ARITH = 2, the default of mix_mx where that kernel runs) the result is another
    rounding of the same sums: within 5e-5 of the split-bf16 FFN2, the same at both tile heights bit for bit, and within the mode's tolerance of
    the oracle.  (The variance adaptor runs in front of the decoder: the bucket decisions of the runs compared here are the same.)"""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c2")
    fs2_option("FS2_ROW8", 1)
    fs2_option("FS2_QKV8", 1)
    run = lambda: model.inference_batch(b["xs"].cuda(), b["ilens"], d_override=b["ds"].cuda())[0]
    try:
        for precision in ("bf16x3", "mix_mx"):
            model.precision = precision
            fs2_option("FS2_FFN2_MX", 0)
            outs = {}
            for row4, mt in ((0, -1), (1, 4), (1, 5)):
                fs2_option("FS2_ROW4", row4)
                fs2_option("FS2_MT4", mt)
                with torch.no_grad():
                    outs[(row4, mt)] = run()
            assert torch.equal(outs[(1, 4)], outs[(1, 5)]), precision
            d8 = float((outs[(0, -1)] - outs[(1, 5)]).abs().max())
            print("c2 [%s]: planes-only residual stream (gemm_row4_bf16) vs fp32 residual rows (gemm_row8_bf16): mel max-abs %.2e" % (precision, d8))
            from tests.conftest import record_measurement
            record_measurement("c2_planes_only_vs_fp32_residual_mel_maxabs_" + precision, d8)
            assert 0.0 < d8 <= 1e-4, (precision, d8)
        fs2_option("FS2_FFN2_MX", 1)                                   # (mix_mx still selected)
        mx = {}
        for mt in (4, 5):
            fs2_option("FS2_MT4", mt)
            with torch.no_grad():
                mx[mt] = run()
        assert torch.equal(mx[4], mx[5])
        d = float((mx[5] - outs[(1, 5)]).abs().max())
        assert 0.0 < d <= 5e-5, d                                      # another rounding of the same sums (measured 2.1e-5), and really another kernel: d > 0
        _c2_body(model, sd, cfg, O, b, "mix_mx")
    finally:
        model.precision = "fp32"


@pytest.mark.parametrize("row8", [0, 1])
def test_c2_fused_pitch_and_energy_predictors(env, row8, fs2_option):
    """The two variance predictors read the same input (reference fastspeech.py:194-196,214-217): in the bf16 modes they run as one launch per
    layer -- layer 0 stacked along N over one A tile, layer 1 a grouped convolution with both scalar heads.  Both kernel families (64-row
    tiles + row pass / 8-wave row-complete) against the oracle, and against the separate launches: predictor outputs within 1e-5 (LayerNorm sums associate differently), bucket
    indices identical."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c2")
    fs2_option("FS2_ROW8", row8)
    model.precision = "bf16x3"
    try:
        outs = []
        for fuse in (1, 0):
            fs2_option("FS2_FUSE_VAR", fuse)
            with torch.no_grad():
                outs.append(model._run(b["xs"].cuda(), b["ilens"], is_inference=True, d_override=b["ds"].cuda(), want=("after", "e_outs", "p_outs", "qe", "qp")))
        fs2_option("FS2_FUSE_VAR", 1)
        _c2_body(model, sd, cfg, O, b, "bf16x3")
    finally:
        model.precision = "fp32"
    f, u = outs
    assert _maxabs(f["e_outs"], u["e_outs"].cpu()) <= 1e-5 and _maxabs(f["p_outs"], u["p_outs"].cpu()) <= 1e-5
    assert torch.equal(f["qe"], u["qe"]) and torch.equal(f["qp"], u["qp"])
    assert _maxabs(f["after"], u["after"].cpu()) <= 1e-5


@pytest.mark.parametrize("qkv8", ["0", "1"])
def test_c2_qkv_kernel_choice(env, qkv8, fs2_option):
    """The fused QKV projection has two bf16 implementations (64 x 128 tiles / three 128 x D passes of an 8-wave workgroup,
    chosen by size): c2 in the parity mode with each one forced."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    fs2_option("FS2_QKV8", qkv8)
    model.precision = "bf16x3"
    try:
        _c2_body(model, sd, cfg, O, make_batch("c2"), "bf16x3")
    finally:
        model.precision = "fp32"


def _c2_body(model, sd, cfg, O, b, precision):
    with torch.no_grad():
        r = model._run(b["xs"].cuda(), b["ilens"], b["olens"], b["ds"].cuda(), b["es"].cuda(), b["ps"].cuda(),
                       is_inference=False, want=("before", "after", "e_outs", "p_outs", "lr_index", "qe", "qp"))
    o = O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])
    assert torch.equal(r["olens"], o["olens"])
    for i in range(b["xs"].shape[0]):
        L = int(o["olens"][i])
        assert torch.equal(r["lr_index"][i, :L].cpu().long(), o["lr_index"][i])
        assert torch.equal(r["qe"][i, :L].cpu().long(), o["qe"][i, :L]) and torch.equal(r["qp"][i, :L].cpu().long(), o["qp"][i, :L])
    d = {k: _maxabs(r[k], o[k]) for k in ("before", "after", "e_outs", "p_outs")}
    d["d_outs"] = _maxabs(r["d_log"], o["d_outs"])
    print("c2 [%s] max-abs vs oracle:" % precision, {k: "%.2e" % v for k, v in d.items()})
    if precision == "bf16":
        assert max(d["before"], d["after"]) < 0.5     # sanity only
    else:
        assert max(d.values()) <= MEL_TOL, d


def test_batch_invariance_and_order(env):
    """Utterances never interact: permuting the batch permutes the outputs bit-for-bit (this is what makes
    sharding across GPUs exact)."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c3", B=12)
    perm = torch.from_numpy(np.random.RandomState(0).permutation(12))
    with torch.no_grad():
        a1, ol1 = model.inference_batch(b["xs"].cuda(), b["ilens"], d_override=b["ds"].cuda())
        a2, ol2 = model.inference_batch(b["xs"][perm].cuda(), b["ilens"][perm], d_override=b["ds"][perm].cuda())
        solo, ol3 = model.inference_batch(b["xs"][5:6, : int(b["ilens"][5])].cuda(), b["ilens"][5:6],
                                          d_override=b["ds"][5:6, : int(b["ilens"][5])].cuda())
    assert torch.equal(ol1[perm], ol2)
    for j, i in enumerate(perm.tolist()):
        L = int(ol1[i])
        assert torch.equal(a1[i, :L], a2[j, :L])
    assert torch.equal(solo[0], a1[5, : int(ol1[5])])


# mix_mx must stay in split-bf16's accuracy class (measured 2.8e-5 until round 5, 4.3e-5 with the planes-only residual stream of round 6; bf16x3 2.1e-5);
# mix_mx4 spends part of the 1e-3 budget on purpose (fp4 cross terms in the decoder's FFN conv: simulated +9.7e-5, tools/arith_sim_ffn_pareto.py)
C3_TOL = {"mix_mx": 1e-4, "mix_mx4": 3e-4}


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "mix_mx", "mix_mx4"])
def test_full_size_c3_properties(env, precision):
    model = env[0]
    model.precision = precision
    try:
        _c3_body(env, precision)
    finally:
        model.precision = "fp32"


_C3_ORACLE = {}      # utterance index -> the oracle's mel: computed once, shared by the four arithmetic modes (as at c4)


def _c3_body(env, precision):
    """BASELINE config c3 (B=64 LJSpeech-shape), free-running with forced durations: frame counts equal the
    duration sums, pads are exactly zero, outputs finite, length-regulator indices exact, and EVERY utterance matches the
    oracle within the mel tolerance."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    from tests.conftest import record_measurement
    b = make_batch("c3")
    with torch.no_grad():
        r = model._run(b["xs"].cuda(), b["ilens"], is_inference=True, d_override=b["ds"].cuda(),
                       want=("after", "before", "lr_index"))
    assert torch.equal(r["olens"], b["olens"])
    after = r["after"]
    assert torch.isfinite(after).all()
    after_h = after.cpu()
    worst = 0.0
    for i in range(after.shape[0]):
        L, T = int(b["olens"][i]), int(b["ilens"][i])
        assert float(after[i, L:].abs().max() if L < after.shape[1] else 0.0) == 0.0
        idx = r["lr_index"][i, :L].cpu().long()
        assert torch.equal(idx, torch.repeat_interleave(torch.arange(T), b["ds"][i, :T])), i          # bit-exact
        if i not in _C3_ORACLE:
            _C3_ORACLE[i] = O.padded_forward(sd, cfg, b["xs"][i:i + 1, :T], b["ilens"][i:i + 1], is_inference=True, d_override=b["ds"][i:i + 1, :T])["after"][0]
        d = float((after_h[i, :L] - _C3_ORACLE[i]).abs().max())
        assert d <= C3_TOL.get(precision, MEL_TOL), (i, d)
        worst = max(worst, d)
    print("c3 [%s] all %d utterances (%d frames): worst mel max-abs vs oracle %.2e" % (precision, after.shape[0], int(b["olens"].sum()), worst))
    record_measurement("c3_all_utterances_mel_maxabs_" + precision, worst)


def test_device_driven_layout_matches_host_driven(env):
    """fs2_decode's device-driven mode (no frame-count read-back between encode and decode; layout, attention work list
    and bounds built by a kernel inside capacities) gives bit-identical mels to the host-driven mode, in every
    arithmetic mode; an insufficient capacity is reported through the status flags, never by corrupting memory."""
    model = env[0]
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c3", B=12)
    xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()      # forced LJSpeech-like durations (~6 k frames)
    for precision in ("fp32", "bf16x3"):
        model.precision = precision
        try:
            with torch.no_grad():
                ref, ol = model.inference_batch(xs, il, d_override=ds)                       # synchronous, host-driven layout
                got, ol_dev = model.inference_batch(xs, il, d_override=ds, sync=False)       # capacities learnt from the call above
                assert model.async_ok()
                assert torch.equal(ol_dev.cpu(), ol)
                Lmax = int(ol.max())
                assert got.shape[1] >= Lmax and float(got[:, Lmax:].abs().max() if got.shape[1] > Lmax else 0.0) == 0.0
                assert torch.equal(got[:, :Lmax], ref), "device-driven layout changed the result (%s)" % precision
                r = model._run(xs, il, is_inference=True, want=("after",), d_override=ds, capacity=(int(ol.sum()) // 2, Lmax + 32))
                st = r["status"].cpu()
                assert int(st[2]) & 1, "row overflow not flagged: %s" % st.tolist()
                r = model._run(xs, il, is_inference=True, want=("after",), d_override=ds, capacity=(int(ol.sum()) + 64, Lmax - 8))
                assert int(r["status"].cpu()[2]) & 2, "Lmax overflow not flagged"
                pk, ol_dev = model.inference_batch(xs, il, d_override=ds, sync=False, packed=True)      # packed output, device offsets
                assert model.async_ok()
                want_pk, _ = model.inference_batch(xs, il, d_override=ds, packed=True)
                assert torch.equal(pk[: want_pk.shape[0]], want_pk)
                again, _ = model.inference_batch(xs, il, d_override=ds)                      # the handle is still healthy
                assert torch.equal(again, ref)
        finally:
            model.precision = "fp32"


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_device_driven_layout_random_batches(env, seed):
    """Host-driven vs device-driven layout on ragged random batches: single-phoneme utterances, all-zero duration rows (the
    length regulator's all-ones rule), zero durations inside, one utterance much longer than the rest, B = 1."""
    model = env[0]
    rs = np.random.RandomState(100 + seed)
    B = [1, 5, 17, 33][seed]
    il = torch.from_numpy(rs.randint(1, 60, size=B)).long()
    il[rs.randint(B)] = 1
    if B > 1:
        il[rs.randint(B)] = 150
    T = int(il.max())
    xs = torch.zeros(B, T, dtype=torch.long)
    ds = torch.zeros(B, T, dtype=torch.long)
    for b in range(B):
        t = int(il[b])
        xs[b, :t] = torch.from_numpy(rs.randint(1, 68, size=t))
        ds[b, :t] = torch.from_numpy(rs.randint(0, 12, size=t))
    ds[0, : int(il[0])] = 0                                  # all-zero row -> every phoneme gets one frame
    xs, ds = xs.cuda(), ds.cuda()
    model.precision = "bf16x3"
    try:
        with torch.no_grad():
            ref, ol = model.inference_batch(xs, il, d_override=ds)
            assert int(ol[0]) == int(il[0])
            got, ol_dev = model.inference_batch(xs, il, d_override=ds, sync=False)
            assert model.async_ok() and torch.equal(ol_dev.cpu(), ol)
            Lmax = int(ol.max())
            assert torch.equal(got[:, :Lmax], ref) and float(got[:, Lmax:].abs().sum()) == 0.0
    finally:
        model.precision = "fp32"


def test_inference_second_call_uses_device_layout(env):
    """`inference(x)` (reference fastspeech.py:339-357): the first call learns the frames-per-phoneme ratio with the host-driven
    layout, later calls run sync-free inside capacities and read the frame count once at the end; same mel either way, and an
    utterance that overflows the learnt capacity falls back transparently."""
    model = env[0]
    rs = np.random.RandomState(7)
    x = torch.from_numpy(rs.randint(1, 68, size=40)).long().cuda()
    with torch.no_grad():
        model._frames_per_token = None
        first = model.inference(x)
        assert model._frames_per_token is not None
        second = model.inference(x)
        assert torch.equal(first, second)
        model._frames_per_token = (0.01, 0.01)          # absurdly small capacities: must fall back, not fail
        third = model.inference(x)
        assert torch.equal(first, third)


def test_hip_graph_replay_of_the_forward(env):
    """The sync-free forward (device-driven layout) captured once as a HIP graph and replayed: same mels as the eager
    call, for the captured ids and for new ids of the same lengths."""
    model = env[0]
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c3", B=4)
    xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()
    model.precision = "bf16x3"
    try:
        with torch.no_grad():
            run = model.capture_graph(xs, il, d_override=ds)
            ref, ol = model.inference_batch(xs, il, d_override=ds)
            mel, ol_dev, status = run(xs)
            assert int(status.cpu()[2]) == 0 and torch.equal(ol_dev.cpu(), ol)
            assert torch.equal(mel[:, : ref.shape[1]], ref)
            xs2 = xs.clone()
            for i in range(xs.shape[0]):
                T = int(il[i])
                xs2[i, :T] = xs[i, :T].flip(0)                 # other phonemes, same lengths
            ref2, _ = model.inference_batch(xs2, il, d_override=ds)
            mel2, _, status = run(xs2)
            assert int(status.cpu()[2]) == 0
            assert torch.equal(mel2[:, : ref2.shape[1]], ref2) and not torch.equal(ref2, ref)
    finally:
        model.precision = "fp32"


def test_planes_only_regime_through_the_other_entry_points(env, fs2_option):
    """Round 6's big-regime forms -- the planes-only residual stream (gemm_row4_bf16 RES 1 / 2 / 3) and, in mix_mx4, the fp4 FFN conv with its row
    scales -- reach every entry point, not only the synchronous per-utterance call the full-size tests make.  On a small batch with the row-complete kernels
    forced (the regime a c3 / c4 batch is in): the device-driven layout is bit-identical to the host-driven one, a HIP-graph replay to the eager call, an
    insufficient capacity NaN-fills the mels, `decoder_out` (rebuilt from the planes: no fp32 row exists any more) and the padded_compat teacher-forced pass
    (`_forward` as the loss path runs it: convolutions and attention see the pad rows) match the oracle."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c3", B=6)
    xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()
    fs2_option("FS2_ROW8", 1)
    fs2_option("FS2_QKV8", 1)
    for precision in ("mix_mx4", "bf16x3"):
        model.precision = precision
        try:
            with torch.no_grad():
                ref, ol = model.inference_batch(xs, il, d_override=ds)                       # host-driven layout
                res = model.inference_batch(xs, il, d_override=ds, sync=False)               # device-driven layout
                assert res.ok() and torch.equal(res[1].cpu(), ol) and torch.equal(res[0][:, : ref.shape[1]], ref), precision
                small = model.inference_batch(xs, il, d_override=ds, sync=False, capacity=(int(ol.sum()) // 2, int(ol.max()) + 32))
                assert not small.ok() and torch.isnan(small[0]).all()
                assert not model.async_ok() and model.async_ok()
                run = model.capture_graph(xs, il, d_override=ds)
                mel, ol_dev, status = run(xs)
                assert int(status.cpu()[2]) == 0 and torch.equal(mel[:, : ref.shape[1]], ref), precision
                r = model._run(xs, il, b["olens"], ds, b["es"].cuda(), b["ps"].cuda(), is_inference=False, want=("after", "decoder_out"))
                rc = model._run(xs, il, b["olens"], ds, b["es"].cuda(), b["ps"].cuda(), is_inference=False, compat=True, want=("after", "before"))
        finally:
            model.precision = "fp32"
        o = O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])
        oc = O.padded_forward(sd, cfg, b["xs"], b["ilens"], b["olens"], b["ds"], b["es"], b["ps"])
        tol = 3e-4 if precision == "mix_mx4" else 1e-4
        for i in range(xs.shape[0]):
            L = int(b["olens"][i])
            assert _maxabs(r["after"][i, :L], o["after"][i, :L]) <= tol, (precision, i)
            assert _maxabs(r["decoder_out"][i, :L], o["decoder_out"][i, :L]) <= 1e-3, (precision, i)
        assert _maxabs(rc["after"], oc["after"]) <= tol and _maxabs(rc["before"], oc["before"]) <= tol, precision


def test_graph_refuses_to_run_after_a_weight_reload_and_async_ring_wraps(env):
    """(1) A captured graph holds pointers into the library's weight copies: after the weights were re-uploaded `run` must raise instead
    of replaying onto released memory.  (2) More asynchronous calls in flight than pinned slots (16): the ring waits for and folds the
    oldest; every result stays valid and bit-identical."""
    model, sd, cfg, O = env
    from fastspeech2_amd import _lib
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c3", B=3)
    xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()
    with torch.no_grad():
        run = model.capture_graph(xs, il, d_override=ds)
        mel, _, status = run(xs)
        assert int(status.cpu()[2]) == 0
        model.load_state_dict(sd)                       # same values, new upload: device copies are re-allocated
        model.inference_batch(xs, il, d_override=ds)    # (triggers the upload)
        with pytest.raises(RuntimeError, match="capture"):
            run(xs)
        ref, ol = model.inference_batch(xs, il, d_override=ds)
        outs = [model.inference_batch(xs, il, d_override=ds, sync=False) for _ in range(40)]
        assert model.async_ok()
        for o in outs[::7] + outs[-1:]:
            assert o.ok() and torch.equal(o[0][:, : ref.shape[1]], ref) and torch.equal(o[1].cpu(), ol)
    with pytest.raises(_lib.Fs2Error):
        _lib.set_option("FS2_NO_SUCH_SWITCH", 1)


def test_reference_smoke_shape(env):
    """Counterpart of the reference's only test (tests/test_fastspeech2.py:7-20): B=2, T=L=100, all ones,
    through forward(); here in eval mode, asserting what the reference merely runs."""
    model, sd, cfg, O = env
    x = torch.ones(2, 100, dtype=torch.int64, device="cuda:0")
    il = torch.tensor([100, 100])
    y = torch.ones(2, 100, 80, device="cuda:0")
    dur = torch.ones(2, 100, dtype=torch.int64, device="cuda:0")
    e = torch.ones(2, 100, device="cuda:0")
    p = torch.ones(2, 100, device="cuda:0")
    with torch.no_grad():
        loss, rep = model(x, il, y, il.clone(), dur, e, p)
    assert torch.isfinite(loss)
    assert [list(d.keys())[0] for d in rep] == ["l1_loss", "before_loss", "after_loss", "duration_loss", "energy_loss", "pitch_loss", "loss"]
    assert all(np.isfinite(list(d.values())[0]) for d in rep)


def test_errors(env):
    model, sd, cfg, O = env
    from fastspeech2_amd import _lib
    with pytest.raises(RuntimeError):
        model.inference(torch.ones(5, dtype=torch.int64))                  # CPU tensor: no fallback
    with pytest.raises(_lib.Fs2Error):
        model._run(torch.ones(1, 5, dtype=torch.int64, device="cuda:0"), torch.tensor([9]), is_inference=True)
    with pytest.raises(ValueError):                                        # olens inconsistent with ds
        model._forward(torch.ones(1, 4, dtype=torch.int64, device="cuda:0"), torch.tensor([4]), torch.tensor([3]),
                       torch.ones(1, 4, dtype=torch.int64), torch.ones(1, 4), torch.ones(1, 4))


@pytest.mark.parametrize("bad_id", ["minus_one", "idim"])
def test_out_of_range_phoneme_ids_raise(env, bad_id):
    """A phoneme id outside [0, idim): the reference's torch.nn.Embedding raises IndexError (fastspeech.py:65-67, core/encoder.py:196).  Here
    fs2_encode marks the utterance (frame count -1, include/fs2.h); the synchronous entry points raise `Fs2IndexError` (an IndexError) before
    any mel is returned, an asynchronous call reports FS2_OVF_BAD_ID: NaN-filled mels, `check()` raises, `async_ok()` is False.  An id outside
    the range in the PADDING behind an utterance's `ilens` raises as well (round 6; round-5 advisor finding): the reference's nn.Embedding indexes
    the whole padded `xs`, so a -1 padding convention fails there too, and `dur_scan` now walks all Tmax positions of a row."""
    from fastspeech2_amd.fastspeech import Fs2IndexError, FS2_OVF_BAD_ID
    from fastspeech2_amd.synthetic import make_batch
    model = env[0]
    b = make_batch("c3", B=6)
    il = b["ilens"]
    val = -1 if bad_id == "minus_one" else model.idim
    j = int(torch.argmin(il))                                   # an utterance with padding behind it
    assert int(il[j]) < b["xs"].shape[1]
    bad = b["xs"].clone()
    bad[j, 3] = val
    padded_only = b["xs"].clone()
    padded_only[j, int(il[j]):] = val
    with torch.no_grad():
        ref, ol = model.inference_batch(b["xs"].cuda(), il)
        assert model.async_ok()
        with pytest.raises(Fs2IndexError):
            model.inference_batch(padded_only.cuda(), il)                        # out-of-range values in the padding: the reference's embedding raises for them too
        with pytest.raises(Fs2IndexError):
            model.inference_batch(bad.cuda(), il)
        with pytest.raises(IndexError):
            model.inference(bad[j, : int(il[j])].cuda())                        # (device-driven attempt first, then the synchronous path raises)
        with pytest.raises(IndexError):
            model._forward(bad.cuda(), il, b["olens"], b["ds"].cuda(), b["es"].cuda(), b["ps"].cuda())
        r = model.inference_batch(bad.cuda(), il, sync=False)
        good = model.inference_batch(b["xs"].cuda(), il, sync=False)
        assert not r.ok() and good.ok()
        assert int(r.status.cpu()[2]) & FS2_OVF_BAD_ID
        assert torch.isnan(r[0]).all()
        with pytest.raises(Fs2IndexError):
            r.check()
        assert not model.async_ok() and model.async_ok()
        assert torch.equal(good[0][:, : ref.shape[1]], ref)


_C4_ORACLE = {}      # utterance index -> oracle (mel, energy codes, pitch codes, predictor outputs): computed once, shared by the three arithmetic modes
EDGE_TOL = 5e-5      # a free-running bucket decision may differ from the oracle's only where the oracle's predictor output is this close to a bin edge


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "mix_mx", "mix_mx4"])
def test_full_size_c4_length_regulator_stress(env, precision):
    """BASELINE config c4 (B=256, 32..512 phonemes, ~0.5 M frames, Lmax > 4000, with Postnet): frame counts,
    zero pads, exact length-regulator indices and the mel of EVERY one of the 256 utterances against the oracle (the oracle's
    results are computed once and shared by the three modes), in fp32, bf16x3 and the bench default mix_mx."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    from tests.conftest import record_measurement
    b = make_batch("c4")
    model.precision = precision
    try:
        with torch.no_grad():
            r = model._run(b["xs"].cuda(), b["ilens"], is_inference=True, d_override=b["ds"].cuda(), want=("after", "lr_index", "qe", "qp"))
            # the same launch sequence again, three times: bit-identical (a data race in a kernel shows up here as a handful of frames that
            # differ from run to run -- attn_w32's staged epilogue had one, ~17 frames of one utterance per few runs at this size)
            for _ in range(3):
                r2 = model._run(b["xs"].cuda(), b["ilens"], is_inference=True, d_override=b["ds"].cuda(), want=("after",))
                assert torch.equal(r2["after"], r["after"])
    finally:
        model.precision = "fp32"
    assert torch.equal(r["olens"], b["olens"])
    after, lri = r["after"], r["lr_index"].cpu().long()
    assert torch.isfinite(after).all()
    ar = torch.arange(after.shape[1], device=after.device).unsqueeze(0)
    pad = ar >= b["olens"].to(after.device).unsqueeze(1)
    assert float((after.abs().amax(-1) * pad).max()) == 0.0                       # pads exactly zero
    for i in range(after.shape[0]):
        L, T = int(b["olens"][i]), int(b["ilens"][i])
        idx = lri[i, :L]
        assert torch.equal(idx, torch.repeat_interleave(torch.arange(T), b["ds"][i, :T])), i   # bit-exact
        assert (lri[i, L:] == -1).all()
    order = torch.argsort(b["olens"]).tolist()
    pick = list(range(after.shape[0]))
    after_h, qe_h, qp_h = after.cpu(), r["qe"].cpu().long(), r["qp"].cpu().long()
    worst, flipped_utts, flipped_frames = 0.0, [], 0
    if not _C4_ORACLE:
        i0 = order[len(order) // 2]
        _tune_oracle_threads(O, sd, cfg, b["xs"][i0:i0 + 1, :int(b["ilens"][i0])], b["ilens"][i0:i0 + 1], b["ds"][i0:i0 + 1, :int(b["ilens"][i0])])
    for i in pick:
        T, L = int(b["ilens"][i]), int(b["olens"][i])
        if i not in _C4_ORACLE:
            o = O.padded_forward(sd, cfg, b["xs"][i:i + 1, :T], b["ilens"][i:i + 1], is_inference=True, d_override=b["ds"][i:i + 1, :T])
            _C4_ORACLE[i] = (o["after"][0], o["qe"][0, :L].long(), o["qp"][0, :L].long(), o["e_outs"][0, :L].float(), o["p_outs"][0, :L].float())
        o_after, o_qe, o_qp, o_e, o_p = _C4_ORACLE[i]
        d = _maxabs(after_h[i, :L], o_after)
        if d > MEL_TOL:
            # The pass is free-running in pitch and energy: ~1 M bucket decisions per mode, each a predictor output within ~1e-5 of the
            # CPU's, so a few land on the other side of a bin edge in the reduced-precision modes (SURVEY.md hard part 2; fp32 has none).
            # Such an utterance must (a) differ from the oracle only in bucket indices that moved to the NEIGHBOURING bucket at frames where
            # the oracle's own predictor output lies within EDGE_TOL (5 x the 1e-5 predictor error of these modes) of the edge between
            # the two, and (b) match the oracle within the tolerance once pitch and energy are teacher-forced (the arithmetic, without
            # the decision).
            flips = 0
            for q_dev, q_orc, x_orc, bins in ((qe_h[i, :L], o_qe, o_e, sd["energy_predictor.energy_bins"]), (qp_h[i, :L], o_qp, o_p, sd["pitch_predictor.pitch_bins"])):
                t = torch.nonzero(q_dev != q_orc).flatten()
                if len(t):
                    assert ((q_dev[t] - q_orc[t]).abs() == 1).all(), (i, "a bucket index moved by more than one")
                    edge = bins.float()[torch.minimum(q_dev[t], q_orc[t])]
                    assert float((x_orc[t] - edge).abs().max()) <= EDGE_TOL, (i, L, float((x_orc[t] - edge).abs().max()))
                    flips += len(t)
            assert precision != "fp32" and flips > 0, (i, L, d, flips)
            sub = {k: b[k][i:i + 1] for k in ("xs", "ilens", "ds", "olens", "es", "ps")}
            sub["xs"], sub["ds"], sub["es"], sub["ps"] = sub["xs"][:, :T], sub["ds"][:, :T], sub["es"][:, :L], sub["ps"][:, :L]
            model.precision = precision
            try:
                with torch.no_grad():
                    rt = model._run(sub["xs"].cuda(), sub["ilens"], sub["olens"], sub["ds"].cuda(), sub["es"].cuda(), sub["ps"].cuda(), is_inference=False, want=("after",))
            finally:
                model.precision = "fp32"
            ot = O.per_utterance_forward(sd, cfg, sub["xs"], sub["ilens"], sub["ds"], sub["es"], sub["ps"])
            d = _maxabs(rt["after"], ot["after"])
            flipped_utts.append(i)
            flipped_frames += flips
        assert d <= MEL_TOL, (i, L, d)
        worst = max(worst, d)
    print("c4 [%s]: %d frames, Lmax %d; %d / %d utterances (L %d..%d) vs the oracle: worst mel max-abs %.2e; %d bucket decision(s) in %d utterance(s) "
          "fell on the other side of a bin edge (those utterances verified teacher-forced)"
          % (precision, int(b["olens"].sum()), after.shape[1], len(pick), after.shape[0], int(b["olens"][order[0]]), int(b["olens"][order[-1]]), worst,
             flipped_frames, len(flipped_utts)))
    record_measurement("c4_all256_mel_maxabs_" + precision, worst)
    record_measurement("c4_all256_flipped_bucket_decisions_" + precision, flipped_frames)


def _tune_oracle_threads(O, sd, cfg, xs, il, ds):
    """The CPU oracle's B = 1 calls are small GEMM / conv calls: all logical cores of a two-socket host oversubscribe badly (bench.py's
    cpu_baseline tunes the same way).  Picks the fastest of a few thread counts on one utterance; returns the previous setting."""
    import time
    prev = torch.get_num_threads()
    ncpu = os.cpu_count() or 1
    best, best_dt = prev, float("inf")
    for nt in sorted({t for t in (8, 16, 32, ncpu // 2, ncpu) if 1 <= t <= ncpu}):
        torch.set_num_threads(nt)
        O.padded_forward(sd, cfg, xs, il, is_inference=True, d_override=ds)
        t0 = time.perf_counter()
        O.padded_forward(sd, cfg, xs, il, is_inference=True, d_override=ds)
        dt = time.perf_counter() - t0
        if dt < best_dt:
            best, best_dt = nt, dt
    torch.set_num_threads(best)
    return prev


def _explain_by_bucket_flips(model, precision, b, i, oracle, sd, cfg, O, decisions=None):
    """An utterance of a pass that is free-running in pitch and energy differs from the oracle by more than the tolerance: legitimate only if
    (a) its bucket indices differ from the oracle's solely by moves to the NEIGHBOURING bucket at frames where the oracle's own predictor
    output lies within EDGE_TOL of the edge between the two, and (b) it matches the oracle within the tolerance once pitch and energy are
    teacher-forced.  ``decisions`` = (qe, qp) the run under test actually embedded for this utterance ([L] each); without them the utterance is run
    once more on its own to get some (another kernel-variant regime: its decisions are its own).  Returns (teacher-forced max-abs, number of
    flipped decisions)."""
    T, L = int(b["ilens"][i]), int(b["olens"][i])
    o_after, o_qe, o_qp, o_e, o_p = oracle
    if decisions is None:
        model.precision = precision
        try:
            with torch.no_grad():
                r1 = model._run(b["xs"][i:i + 1, :T].cuda(), b["ilens"][i:i + 1], is_inference=True, d_override=b["ds"][i:i + 1, :T].cuda(), want=("after", "qe", "qp"))
        finally:
            model.precision = "fp32"
        decisions = (r1["qe"][0, :L].cpu().long(), r1["qp"][0, :L].cpu().long())
    flips = 0
    for q_dev, q_orc, x_orc, bins in ((decisions[0], o_qe, o_e, sd["energy_predictor.energy_bins"]),
                                      (decisions[1], o_qp, o_p, sd["pitch_predictor.pitch_bins"])):
        t = torch.nonzero(q_dev != q_orc).flatten()
        if len(t):
            assert ((q_dev[t] - q_orc[t]).abs() == 1).all(), (i, "a bucket index moved by more than one")
            edge = bins.float()[torch.minimum(q_dev[t], q_orc[t])]
            assert float((x_orc[t] - edge).abs().max()) <= EDGE_TOL, (i, L, float((x_orc[t] - edge).abs().max()))
            flips += len(t)
    assert precision != "fp32" and flips > 0, (i, L, flips)
    sub = {k: b[k][i:i + 1] for k in ("xs", "ilens", "ds", "olens", "es", "ps")}
    sub["xs"], sub["ds"], sub["es"], sub["ps"] = sub["xs"][:, :T], sub["ds"][:, :T], sub["es"][:, :L], sub["ps"][:, :L]
    model.precision = precision
    try:
        with torch.no_grad():
            rt = model._run(sub["xs"].cuda(), sub["ilens"], sub["olens"], sub["ds"].cuda(), sub["es"].cuda(), sub["ps"].cuda(), is_inference=False, want=("after",))
    finally:
        model.precision = "fp32"
    ot = O.per_utterance_forward(sd, cfg, sub["xs"], sub["ilens"], sub["ds"], sub["es"], sub["ps"])
    return _maxabs(rt["after"], ot["after"]), flips


def test_c5_all_1024_utterances_through_the_eight_shards(env, monkeypatch):
    """BASELINE config c5 (batch = 1024 sharded over 8 MI355X), ALL of it: the eight LPT shards (`shard_indices`) run one after another on this
    box's one GPU through `ShardedSynthesizer` -- the sync-free single-GPU path into the very send buffers of the collective -- with
    `torch.distributed` replaced by a stand-in that plays the eight ranks in turn and hands the last one the eight send buffers as its all-gather
    result (tests/fake_dist.py); `gather_shards` then does what it does on a node (frame counts out of the tail rows, device-side offsets,
    `fs2_op_unpack_rows_dev`).  Checked, in mix_mx (what bench.py runs):
      * SURVEY.md section 8e's criterion: the assembled result is BIT-IDENTICAL to the one-GPU call of all 1,024 utterances -- every rank names the
        whole batch as the basis of its kernel-variant choice (`regime`, include/fs2.h: fs2_batch.regime_*; round 6), so a 9.6 k-token shard runs
        the very kernels the 77 k-token batch runs (round 5: variants by the shard's own size, ~2e-5 apart, one utterance 0.044 apart through a bucket
        decision at a bin edge);
      * so is every shard run alone through the plain entry point with the same `regime` (host-driven layout): mels AND the 1.2 M bucket decisions;
      * every one of the 1,024 utterances (613 k frames) matches the oracle -- within the tolerance, or, the pass being free-running in pitch and
        energy, with its differences explained by decisions that fell on the other side of a bin edge the oracle's own predictor output touches
        (then re-verified teacher-forced), as at c4;
      * without the override (`global_regime=False`) the shards agree with the one-GPU call to 5e-5 wherever the decisions agree (recorded).
    The real 8-rank collective is covered over gloo (tests/test_parallel_gloo.py) and over nccl in tests/test_gpu_zmulti.py."""
    model, sd, cfg, O = env
    import fastspeech2_amd.parallel as P
    from fastspeech2_amd.synthetic import make_batch
    from tests import fake_dist
    from tests.conftest import record_measurement
    b = make_batch("c5")
    B, W = 1024, 8
    parts = P.shard_indices(b["ilens"].tolist(), W)
    assert sorted(sum(parts, [])) == list(range(B)) and all(100 <= len(p_) <= 160 for p_ in parts)
    xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()
    whole = (int(il.sum()), B)
    model.precision = "mix_mx"
    try:
        with torch.no_grad():
            un = model._run(xs, il, is_inference=True, d_override=ds, want=("after", "qe", "qp"))          # the one-GPU call (host-driven layout)
            assert torch.equal(un["olens"], b["olens"]) and torch.isfinite(un["after"]).all()
            un_after, un_qe, un_qp = un["after"].cpu(), un["qe"].cpu().long(), un["qp"].cpu().long()
            del un
            model.inference_batch(xs[:64], il[:64], d_override=ds[:64])                                      # (teaches the capacity predictor a ratio)
            ratio = model._frames_per_token
            mels, ol_dev = fake_dist.run_all_ranks(P, monkeypatch, model, W, xs, il, (float(ratio[0]) * 1.05, float(ratio[1]) * 1.2), d_override=ds)
            assert model.async_ok()
            assert torch.equal(ol_dev.cpu(), b["olens"])
            mels_h = mels.cpu()
            del mels
            Lm = un_after.shape[1]
            assert torch.equal(mels_h[:, :Lm], un_after) and float(mels_h[:, Lm:].abs().sum()) == 0.0, "sharded (8 ranks) != the one-GPU call of the whole batch"
            # every shard alone, naming the whole batch: bit-identical too, bucket decisions included; and once without the override
            worst_local, apart_local = 0.0, 0
            for r, p_ in enumerate(parts):
                sel = torch.as_tensor(p_)
                il_s = il[sel]
                Tm = int(il_s.max())
                rs = model._run(xs[sel.cuda()][:, :Tm], il_s, is_inference=True, d_override=ds[sel.cuda()][:, :Tm], want=("after", "qe", "qp"), regime=whole)
                assert torch.equal(rs["olens"], b["olens"][sel])
                a_h, qe_h, qp_h = rs["after"].cpu(), rs["qe"].cpu().long(), rs["qp"].cpu().long()
                rl = model._run(xs[sel.cuda()][:, :Tm], il_s, is_inference=True, d_override=ds[sel.cuda()][:, :Tm], want=("after", "qe", "qp"))
                l_h, lqe_h, lqp_h = rl["after"].cpu(), rl["qe"].cpu().long(), rl["qp"].cpu().long()
                for j, g in enumerate(p_):
                    L = int(rs["olens"][j])
                    assert torch.equal(un_after[g, :L], a_h[j, :L]), (r, g)
                    assert torch.equal(un_qe[g, :L], qe_h[j, :L]) and torch.equal(un_qp[g, :L], qp_h[j, :L]), (r, g)
                    if torch.equal(un_qe[g, :L], lqe_h[j, :L]) and torch.equal(un_qp[g, :L], lqp_h[j, :L]):
                        worst_local = max(worst_local, _maxabs(un_after[g, :L], l_h[j, :L]))
                    else:
                        apart_local += 1
    finally:
        model.precision = "fp32"
    assert worst_local <= 5e-5, worst_local
    # every utterance against the oracle
    i0 = int(torch.argmax(il))
    prev_threads = _tune_oracle_threads(O, sd, cfg, b["xs"][i0:i0 + 1, :int(il[i0])], il[i0:i0 + 1], b["ds"][i0:i0 + 1, :int(il[i0])])
    worst, flipped_utts, flipped_frames = 0.0, 0, 0
    try:
        for i in range(B):
            T, L = int(il[i]), int(b["olens"][i])
            o = O.padded_forward(sd, cfg, b["xs"][i:i + 1, :T], il[i:i + 1], is_inference=True, d_override=b["ds"][i:i + 1, :T])
            orc = (o["after"][0], o["qe"][0, :L].long(), o["qp"][0, :L].long(), o["e_outs"][0, :L].float(), o["p_outs"][0, :L].float())
            d = _maxabs(mels_h[i, :L], orc[0])
            if d > MEL_TOL:
                d, flips = _explain_by_bucket_flips(model, "mix_mx", b, i, orc, sd, cfg, O, decisions=(un_qe[i, :L], un_qp[i, :L]))
                flipped_utts += 1
                flipped_frames += flips
            assert d <= MEL_TOL, (i, L, d)
            worst = max(worst, d)
    finally:
        torch.set_num_threads(prev_threads)
    print("c5, all %d utterances / %d frames through the 8 LPT shards (%s utterances each): assembled == the one-GPU call of the whole batch == each shard run "
          "alone with regime = the whole batch, bit for bit (mels and bucket decisions); worst mel max-abs vs the oracle %.2e; %d bucket decision(s) in %d "
          "utterance(s) on the other side of a bin edge (verified teacher-forced, mel error of those utterances beside an agreed decision); shards with their OWN "
          "regime vs the whole batch: within %.1e on the utterances with the same decisions, %d utterance(s) apart through a decision"
          % (B, int(b["olens"].sum()), [len(p_) for p_ in parts], worst, flipped_frames, flipped_utts, worst_local, apart_local))
    record_measurement("c5_all1024_mel_maxabs_mix_mx", worst)
    record_measurement("c5_all1024_flipped_bucket_decisions_mix_mx", flipped_frames)
    record_measurement("c5_all1024_own_regime_vs_whole_batch_same_decisions_maxabs_mix_mx", worst_local)
    record_measurement("c5_all1024_own_regime_vs_whole_batch_utterances_apart_by_a_decision", apart_local)


@pytest.mark.parametrize("B,world", [(19, 2), (67, 8)])
def test_shards_named_after_the_whole_batch_equal_the_one_gpu_call(env, monkeypatch, B, world):
    """tests/test_gpu_zmulti.py's exact shapes (its batches of 8 W + 3 utterances, bf16x3, teacher-forced durations) on ONE GPU: B = 19 over 2 ranks
    puts the whole batch (regime 11.8 k rows: attn_w32) and its shards (6 k: attn_bf16) on different sides of the attention-kernel threshold,
    B = 67 over 8 ranks on different sides of every threshold (row-complete LayerNorm-fused kernels, split-K, attn_w32).  With every rank naming
    the whole batch (`regime`), both the sync-free sharded path (device-driven layout, packs -> `gather_shards`) and each shard's synchronous call
    are bit-identical to the one-GPU call -- mels and bucket decisions -- which is what that file asserts over RCCL on a multi-GPU node."""
    model = env[0]
    import fastspeech2_amd.parallel as P
    from fastspeech2_amd.synthetic import make_batch
    from tests import fake_dist
    b = make_batch("c5", B=B)
    xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()
    parts = P.shard_indices(il.tolist(), world)
    whole = (int(il.sum()), B)
    model.precision = "bf16x3"
    try:
        with torch.no_grad():
            un = model._run(xs, il, is_inference=True, d_override=ds, want=("after", "qe", "qp"))
            ref, ol = model.inference_batch(xs, il, d_override=ds)
            assert torch.equal(ref, un["after"])
            ratio = model._frames_per_token
            mels, ol_dev = fake_dist.run_all_ranks(P, monkeypatch, model, world, xs, il, (float(ratio[0]) * 1.05, float(ratio[1]) * 1.2), d_override=ds)
            assert model.async_ok() and torch.equal(ol_dev.cpu(), ol)
            L = ref.shape[1]
            assert torch.equal(mels[:, :L], ref) and float(mels[:, L:].abs().sum()) == 0.0, "sync-free sharded != the one-GPU call"
            differs = 0
            for p_ in parts:
                sel = torch.as_tensor(p_)
                Tm = int(il[sel].max())
                rs = model._run(xs[sel.cuda()][:, :Tm], il[sel], is_inference=True, d_override=ds[sel.cuda()][:, :Tm], want=("after", "qe", "qp"), regime=whole)
                rl = model._run(xs[sel.cuda()][:, :Tm], il[sel], is_inference=True, d_override=ds[sel.cuda()][:, :Tm], want=("after",))
                for j, g in enumerate(p_):
                    n = int(ol[g])
                    assert torch.equal(rs["after"][j, :n], ref[g, :n]), g
                    assert torch.equal(rs["qe"][j, :n], un["qe"][g, :n]) and torch.equal(rs["qp"][j, :n], un["qp"][g, :n]), g
                    differs += int(not torch.equal(rl["after"][j, :n], ref[g, :n]))
                    assert _maxabs(rl["after"][j, :n], ref[g, :n].cpu()) <= 1e-3
            # (the premise of the test: without the override these shapes do pick other kernels -- if this ever reads 0 the thresholds have moved
            #  and the shapes no longer straddle them)
            assert differs > 0, "the shards' own regimes picked the whole batch's kernels: the test no longer straddles a threshold"
    finally:
        model.precision = "fp32"


def test_sharded_synthesizer_over_nccl_world_size_1(env):
    """`ShardedSynthesizer(model)` with an initialised RCCL process group (world size 1: the one GPU of this box): LPT
    partition, sync-free capacity packs, the ONE all_gather_into_tensor over the nccl backend, device-side unpack ==
    the unsharded batched call, bit for bit; an insufficient capacity is reported by ok() and NaN-fills the mels."""
    import socket
    import torch.distributed as dist
    from fastspeech2_amd.parallel import ShardedSynthesizer
    from fastspeech2_amd.synthetic import make_batch
    model = env[0]
    b = make_batch("c3", B=24)
    xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda:0"))
    model.precision = "bf16x3"
    try:
        with torch.no_grad():
            ref, ol = model.inference_batch(xs, il, d_override=ds)
            synth = ShardedSynthesizer(model)
            m1, o1 = synth(xs, il, d_override=ds)                  # synchronous first call (host-driven gather)
            m2, o2 = synth(xs, il, d_override=ds)                  # sync-free: device layout + one collective
            assert synth.ok()
            L = ref.shape[1]
            assert torch.equal(o1.cpu(), ol) and torch.equal(o2.cpu(), ol)
            assert torch.equal(m1[:, :L], ref) and torch.equal(m2[:, :L], ref) and float(m2[:, L:].abs().sum()) == 0.0
            over = ShardedSynthesizer(model, overlap=True)         # throughput mode: gather + unpack on a side stream
            over._ratio = synth._ratio
            outs = [over(xs, il, d_override=ds) for _ in range(3)] # three batches in flight, none waited for
            over.wait()
            assert over.ok()
            for m, o in outs:
                assert torch.equal(o.cpu(), ol) and torch.equal(m[:, :L], ref)
            torch.cuda.synchronize()
            model.overlap_encoder = True                           # ... and each call's encoder on its own side stream as well (bench.py's N > 1 mode)
            try:
                outs = [over(xs, il, d_override=ds) for _ in range(4)]
                over.wait()
                assert over.ok()
                for m, o in outs:
                    assert torch.equal(o.cpu(), ol) and torch.equal(m[:, :L], ref)
            finally:
                model.overlap_encoder = False
            synth._ratio = (0.05, 0.05)                            # absurd capacities -> overflow must be visible, not silent
            m3, _ = synth(xs, il, d_override=ds)
            assert not synth.ok()
            assert torch.isnan(m3).any()                           # the overflowed rank's pack is NaN-filled
            model.async_ok()                                       # (drains the model's own bookkeeping)
            synth._ratio = over._ratio
            bad = xs.clone()
            bad[1, 0] = model.idim                                 # a phoneme id outside [0, idim): the reference's nn.Embedding raises
            m4, _ = synth(bad, il, d_override=ds)
            assert not synth.ok() and torch.isnan(m4).any()
            m5, _ = synth(xs, il, d_override=ds)
            assert synth.ok() and torch.equal(m5[:, :L], ref)
    finally:
        model.precision = "fp32"
        dist.destroy_process_group()


def test_async_overflow_is_never_silent(env):
    """`inference_batch(sync=False)`: each call carries its own validity record (AsyncMels.ok / .check / .status); several
    calls may be in flight; an overflowed call returns NaN-filled mels (padded and packed) and is reported by async_ok()
    even when later calls were fine."""
    from fastspeech2_amd.fastspeech import Fs2CapacityError
    from fastspeech2_amd.synthetic import make_batch
    model = env[0]
    b = make_batch("c3", B=6)
    xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()
    with torch.no_grad():
        ref, ol = model.inference_batch(xs, il, d_override=ds)
        assert model.async_ok()
        good1 = model.inference_batch(xs, il, d_override=ds, sync=False)
        bad = model.inference_batch(xs, il, d_override=ds, sync=False, capacity=(int(ol.sum()) // 3, int(ol.max()) + 32))
        bad_pk = model.inference_batch(xs, il, d_override=ds, sync=False, packed=True, capacity=(int(ol.sum()) // 3, int(ol.max()) + 32))
        good2 = model.inference_batch(xs, il, d_override=ds, sync=False)
        assert good1.ok() and good2.ok() and not bad.ok() and not bad_pk.ok()
        assert int(bad.status.cpu()[2]) & 1
        assert torch.isnan(bad[0]).all() and torch.isnan(bad_pk[0]).all()
        with pytest.raises(Fs2CapacityError):
            bad.check()
        assert not model.async_ok()                  # the overflow of an EARLIER call is still reported
        assert model.async_ok()                      # ... once
        mel, ol_dev = good2
        assert torch.equal(mel[:, : ref.shape[1]], ref) and torch.equal(ol_dev.cpu(), ol)
        empty, eo = model.inference_batch(xs[:0], il[:0])
        assert empty.shape[0] == 0 and eo.numel() == 0


def test_overlap_encoder_mode_changes_nothing_but_the_schedule(env):
    """`model.overlap_encoder` (throughput mode of the sync-free entry points): each call's token-level half runs on a side stream, so that it executes
    while the previous call's frame-level kernels still run.  Two different batches alternate, eight calls in flight, none waited for: every result
    is bit-identical to the synchronous call of its batch, frame counts included, and the validity records stay per call."""
    from fastspeech2_amd.synthetic import make_batch
    model = env[0]
    b1, b2 = make_batch("c3", B=24), make_batch("c2", B=12)
    ins = [(b["xs"].cuda(), b["ilens"], b["ds"].cuda()) for b in (b1, b2)]
    model.precision = "mix_mx"
    try:
        with torch.no_grad():
            free = [model.inference_batch(x, il) for x, il, d in ins]                    # free-running durations: the frame counts are the device's own
            refs = [model.inference_batch(x, il, d_override=d) for x, il, d in ins]      # (last: these calls leave the capacity predictor the ratios the forced durations need)
            assert model.async_ok()
            torch.cuda.synchronize()
            model.overlap_encoder = True
            outs = [model.inference_batch(*ins[i & 1][:2], d_override=ins[i & 1][2], sync=False) for i in range(8)]
            outs_free = [model.inference_batch(*ins[i & 1][:2], sync=False, packed=bool(i & 2)) for i in range(8)]
            assert len(outs) == 8
            # a shard cut out of a batch by rows AND columns is not contiguous: the copy that makes it so must run on the encoder's stream too
            sel = torch.argsort(ins[0][1])[:8]
            m = int(ins[0][1][sel].max())
            assert m < ins[0][0].shape[1]
            model.overlap_encoder = False
            sub_ref = model.inference_batch(ins[0][0][sel.cuda()][:, :m], ins[0][1][sel], d_override=ins[0][2][sel.cuda()][:, :m])
            torch.cuda.synchronize()
            model.overlap_encoder = True
            subs = []
            for _ in range(4):
                with torch.cuda.stream(model.input_stream(ins[0][0].device)):
                    xs_s, ds_s = ins[0][0][sel.cuda()][:, :m], ins[0][2][sel.cuda()][:, :m]
                # (exact capacities from here on: the synchronous calls on free-running durations and on the eight shortest utterances have lowered the
                #  predictor's frames-per-phoneme ratios, and an under-predicted capacity -- reported by ok(), correctly -- is not what this test is about)
                subs.append(model.inference_batch(xs_s, ins[0][1][sel], d_override=ds_s, sync=False,
                                                  capacity=(int(sub_ref[1].sum()) + 64 * len(sel), int(sub_ref[1].max()) + 32)))
                for k in (1, 0):                                                                            # (two whole batches keep the caller's stream busy in between)
                    cap_k = (int(refs[k][1].sum()) + 64 * len(refs[k][1]), int(refs[k][1].max()) + 32)
                    outs.append(model.inference_batch(*ins[k][:2], d_override=ins[k][2], sync=False, capacity=cap_k))
            flags = [int(o.status.cpu()[2]) for o in outs + outs_free + subs]
            assert not any(flags), {i: f for i, f in enumerate(flags) if f}
            assert all(o.ok() for o in outs + outs_free + subs) and model.async_ok()
            for mel, ol in subs:
                assert torch.equal(ol.cpu(), sub_ref[1]) and torch.equal(mel[:, : sub_ref[0].shape[1]], sub_ref[0])
            for i, (mel, ol) in enumerate(outs):
                ref, rol = refs[i & 1] if i < 8 else refs[1 - (i & 1)]          # (the eight alternating calls, then the b2 / b1 pairs queued between the shard calls)
                assert torch.equal(ol.cpu(), rol) and torch.equal(mel[:, : ref.shape[1]], ref) and float(mel[:, ref.shape[1]:].abs().sum()) == 0.0
            for i, (mel, ol) in enumerate(outs_free):
                ref, rol = free[i & 1]
                assert torch.equal(ol.cpu(), rol)
                if i & 2:
                    st = (torch.cumsum(rol, 0) - rol).tolist()
                    for j in range(len(rol)):
                        assert torch.equal(mel[st[j]:st[j] + int(rol[j])], ref[j, : int(rol[j])])
                else:
                    assert torch.equal(mel[:, : ref.shape[1]], ref)
    finally:
        model.overlap_encoder = False
        model.precision = "fp32"


@pytest.mark.parametrize("overlap_encoder", [False, True])
def test_two_steps_in_flight_on_two_streams_change_nothing_but_the_schedule(env, overlap_encoder):
    """bench.py's one-GPU schedule: consecutive sync-free calls are issued on alternating streams, so two whole forwards are in flight on the chip
    (with `overlap_encoder` each stream has its own side stream for the token-level half).  Two different batches, twelve calls on two streams (so
    both batches run on both streams and against each other), padded and packed results: bit-identical to the synchronous calls, frame counts and
    validity records included."""
    from fastspeech2_amd.synthetic import make_batch
    model = env[0]
    b1, b2 = make_batch("c3", B=24), make_batch("c2", B=12)
    ins = [(b["xs"].cuda(), b["ilens"]) for b in (b1, b2)]
    model.precision = "mix_mx"
    try:
        with torch.no_grad():
            refs = [model.inference_batch(x, il) for x, il in ins]
            caps = [(int(r[1].sum()) + 64 * len(r[1]), int(r[1].max()) + 32) for r in refs]
            torch.cuda.synchronize()
            model.overlap_encoder = overlap_encoder
            from fastspeech2_amd import StepStreams
            rot = StepStreams(2)
            order = [0, 1, 1, 0, 0, 0, 1, 1, 1, 0, 0, 1]         # batch of call i; call i runs on stream i & 1
            outs = []
            for i, k in enumerate(order):
                with rot.next():
                    outs.append(model.inference_batch(*ins[k], sync=False, packed=(i % 3 == 2), capacity=caps[k]))
            rot.join()                                           # (no host synchronisation: the comparisons below run on the current stream)
            assert all(o.ok() for o in outs) and model.async_ok()
            for i, k in enumerate(order):
                mel, ol = outs[i]
                ref, rol = refs[k]
                assert torch.equal(ol.cpu(), rol)
                if i % 3 == 2:
                    st = (torch.cumsum(rol, 0) - rol).tolist()
                    for j in range(len(rol)):
                        assert torch.equal(mel[st[j]:st[j] + int(rol[j])], ref[j, : int(rol[j])])
                else:
                    assert torch.equal(mel[:, : ref.shape[1]], ref) and float(mel[:, ref.shape[1]:].abs().sum()) == 0.0
    finally:
        model.overlap_encoder = False
        model.precision = "fp32"


@pytest.mark.parametrize("alpha", [0.7, 1.5])
def test_duration_alpha_speed_control(env, alpha):
    """Length-regulator speed control (reference length_regulator.py:57-59) through the batched entry point: the mels equal
    the oracle's for durations round(d * alpha)."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c2", B=3, tlens=[31, 12, 20])
    ds_a = torch.round(b["ds"].float() * alpha).long()
    with torch.no_grad():
        mel, ol = model.inference_batch(b["xs"].cuda(), b["ilens"], d_override=b["ds"].cuda(), alpha=alpha)
    for i in range(3):
        T = int(b["ilens"][i])
        o = O.padded_forward(sd, cfg, b["xs"][i:i + 1, :T], b["ilens"][i:i + 1], is_inference=True, d_override=ds_a[i:i + 1, :T])
        L = int(o["olens"][0])
        assert int(ol[i]) == L and _maxabs(mel[i, :L], o["after"][0]) <= MEL_TOL


@pytest.mark.parametrize("arch", ["default", "ddim256_arch"])
def test_checkpoint_to_device_inference(arch, tmp_path):
    """SURVEY section 8 row f2 on the device, the reference's synthesis flow (inference.py:149-166, train_fastspeech.py:235-244):
    torch.save({"model", "optim", "step", "hp_str", "githash"}) -> torch.load -> hparams from the embedded hp_str ->
    FeedForwardTransformer(idim, odim, hp) -> load_state_dict -> .to(device) -> inference(ids), checked against the oracle;
    also a bare, DataParallel-prefixed `--old_model` checkpoint that lacks the unused keys (strict=False)."""
    import yaml
    from fastspeech2_amd import FeedForwardTransformer, default_hparams, load_checkpoint, hparams_from_str, N_PHONEME_SYMBOLS
    from fastspeech2_amd.synthetic import portable_state_dict, bias_durations
    from oracle import fs2_oracle as O
    hp = default_hparams()
    if arch != "default":
        for k, v in VARIANTS[arch].items():
            hp.model[k] = v
    plain = lambda d: {k: (plain(v) if isinstance(v, dict) else v) for k, v in d.items()}
    hp_str = yaml.safe_dump(plain(hp))
    src = FeedForwardTransformer(N_PHONEME_SYMBOLS, 80, hp)
    sd = bias_durations(portable_state_dict(src.state_dict(), seed=11), 3.0)
    path = str(tmp_path / "checkpoint_58000.pyt")
    torch.save({"model": sd, "optim": {"state": {}, "param_groups": []}, "step": 58000, "hp_str": hp_str, "githash": "deadbeef"}, path)
    # --- what inference.py does
    extras_hp = hparams_from_str(torch.load(path, map_location="cpu", weights_only=True)["hp_str"])
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, extras_hp.audio.num_mels, extras_hp).eval()
    extras = load_checkpoint(model, path)
    assert extras["step"] == 58000 and extras["githash"] == "deadbeef"
    model = model.to("cuda:0")
    cfg = O.config_from_hp(extras_hp, N_PHONEME_SYMBOLS, 80)
    x = torch.from_numpy(np.random.RandomState(3).randint(1, 68, size=37)).long()
    for precision in ("fp32", "bf16x3"):
        model.precision = precision
        with torch.no_grad():
            mel = model.inference(x.cuda())
        o = O.padded_forward(sd, cfg, x.unsqueeze(0), torch.tensor([37]), is_inference=True)
        assert mel.shape == tuple(o["after"][0].shape), (mel.shape, o["after"][0].shape)
        d = _maxabs(mel, o["after"][0])
        print("checkpoint -> device inference [%s, %s]: L=%d, mel max-abs %.2e" % (arch, precision, mel.shape[0], d))
        assert d <= MEL_TOL
    # --- --old_model: bare state dict, DataParallel prefix, unused keys absent -> strict=False
    old = {("module." + k): v for k, v in sd.items() if "concat_linear" not in k and "after_norm" not in k}
    old_path = str(tmp_path / "old.pyt")
    torch.save(old, old_path)
    model2 = FeedForwardTransformer(N_PHONEME_SYMBOLS, 80, extras_hp).eval()
    ex = load_checkpoint(model2, old_path, old_model=True)
    assert ex["missing_keys"] and not ex["unexpected_keys"]
    model2 = model2.to("cuda:0")
    with torch.no_grad():
        mel2 = model2.inference(x.cuda())
    model.precision = "fp32"
    with torch.no_grad():
        assert torch.equal(mel2, model.inference(x.cuda()))       # the unused keys do not enter the path


VARIANTS = {
    "linear_ffn": dict(positionwise_layer_type="linear", positionwise_conv_kernel_size=1),          # modules.py:186-201
    "plain_posenc": dict(use_scaled_pos_enc=False),                                                 # embedding.py:68-80
    "no_batchnorm": dict(use_batch_norm=False),
    "no_postnet": dict(postnet_layers=0),
    "ddim256_arch": dict(ddim=256, dunits=1024, elayers=2, dlayers=3, postnet_layers=3),            # assets/model.txt era
    "conv_k3_ffn": dict(positionwise_conv_kernel_size=3, duration_predictor_layers=3),
    # any adim % aheads == 0 the reference accepts (core/attention.py:18-20): head dims that are not a kernel size run zero-padded
    "aheads4": dict(aheads=4),                                  # d_k 64 (encoder) / 96 -> padded to 128 (decoder)
    "aheads8": dict(aheads=8),                                  # d_k 32 -> 64 / 48 -> 64
    "aheads1_ddim256": dict(aheads=1, ddim=256),                # d_k 256 in both stacks (the 256-wide attention kernels)
    "aheads4_pre_ln_concat": dict(aheads=4, decoder_normalize_before=True, decoder_concat_after=True),
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_config_variants_vs_oracle(name, precision):
    """Non-default hp.model settings the reference supports: same kernels, different
    shapes / epilogues, each checked against the oracle on a small teacher-forced batch in both parity modes."""
    from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
    from fastspeech2_amd.synthetic import portable_state_dict, make_batch
    from oracle import fs2_oracle as O
    hp = default_hparams()
    for k, v in VARIANTS[name].items():
        hp.model[k] = v
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, 80, hp).eval()
    sd = portable_state_dict(model.state_dict(), seed=3)
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    model.precision = precision
    cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, 80)
    b = make_batch("c2", B=3, tlens=[40, 23, 57])
    with torch.no_grad():
        r = model._run(b["xs"].cuda(), b["ilens"], b["olens"], b["ds"].cuda(), b["es"].cuda(), b["ps"].cuda(),
                       is_inference=False, want=("before", "after", "e_outs", "p_outs"))
        x = b["xs"][0, :40].cuda()
        mel = model.inference(x)                       # free-running single utterance (durations ~0 with random weights)
    o = O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])
    d = {k: _maxabs(r[k], o[k]) for k in ("before", "after", "e_outs", "p_outs")}
    d["d_outs"] = _maxabs(r["d_log"], o["d_outs"])
    print("variant %s [%s]:" % (name, precision), {k: "%.1e" % v for k, v in d.items()})
    assert max(d.values()) <= MEL_TOL, d
    oi = O.padded_forward(sd, cfg, b["xs"][:1, :40], b["ilens"][:1], is_inference=True)
    if precision == "fp32":
        assert mel.shape == tuple(oi["after"][0].shape)
        assert _maxabs(mel, oi["after"][0]) <= MEL_TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("name", ["pre_ln", "concat_after", "pre_ln_concat", "enc_pre_ln_dec_concat"])
def test_g7_block_variants_on_device(name, precision, golden_dir):
    """The non-default FFT-block variants of the reference (hp.model.{encoder,decoder}_{normalize_before,concat_after},
    core/encoder.py:53-71,201-202) on the HIP path against the REAL reference's outputs (fixture G7), both parity modes; plus a
    free-running single-utterance call against the oracle."""
    from tests.test_oracle_golden import variant_setup
    from oracle import fs2_oracle as O
    g = np.load(golden_dir + "/g7_block_variants_b2.npz")
    hp, model, sd, cfg = variant_setup(name)
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    model.precision = precision
    with torch.no_grad():
        before, after, d_outs, _, _ = model._forward(_t(g["xs"]), _t(g["ilens"]), _t(g["olens"]), _t(g["ds"]), _t(g["es"]), _t(g["ps"]))
        x = _t(g["xs"])[0, : int(g["ilens"][0])]
        mel = model.inference(x)
    worst = 0.0
    for i in range(2):
        L, T = int(g["olens"][i]), int(g["ilens"][i])
        worst = max(worst, _maxabs(after[i, :L], g["%s_after_%d" % (name, i)]), _maxabs(before[i, :L], g["%s_before_%d" % (name, i)]),
                    _maxabs(d_outs[i, :T], g["%s_d_outs_%d" % (name, i)]))
    print("G7 %s [%s] max-abs vs the reference %.2e" % (name, precision, worst))
    assert worst <= MEL_TOL
    oi = O.padded_forward(sd, cfg, torch.from_numpy(g["xs"][:1, : int(g["ilens"][0])]), torch.from_numpy(g["ilens"][:1]), is_inference=True)
    if precision == "fp32":
        assert mel.shape == tuple(oi["after"][0].shape) and _maxabs(mel, oi["after"][0]) <= MEL_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
def test_g8_reduction_factor_on_device(precision, golden_dir):
    """reduction_factor = 2 (reference fastspeech.py:153,228-230): the HIP path against the real reference's outputs (fixture G8),
    teacher-forced (_forward) and free-running (inference, inference_batch in both layout modes, padded and packed, and through
    ShardedSynthesizer); mel frames = 2 x decoder frames."""
    from tests.test_oracle_golden import reduction_setup
    g = np.load(golden_dir + "/g8_reduction_factor2_b2.npz")
    hp, model, sd, cfg = reduction_setup()
    model.load_state_dict(sd)
    model = model.cuda()
    model.precision = precision
    tol = 1e-3
    xs, il = torch.from_numpy(g["xs"]), torch.from_numpy(g["ilens"])
    with torch.no_grad():
        for i in range(2):
            T, L = int(il[i]), int(g["olens"][i])
            r = model._forward(xs[i:i + 1, :T].cuda(), il[i:i + 1], torch.from_numpy(g["olens"][i:i + 1]), torch.from_numpy(g["ds"][i:i + 1, :T]).cuda(),
                               torch.from_numpy(g["es"][i:i + 1, :L]).cuda(), torch.from_numpy(g["ps"][i:i + 1, :L]).cuda(), is_inference=False)
            assert r[1].shape[1] == 2 * L
            assert float((r[1][0].cpu() - torch.from_numpy(g["tf_after_%d" % i])).abs().max()) < tol
            assert float((r[0][0].cpu() - torch.from_numpy(g["tf_before_%d" % i])).abs().max()) < tol
            y = model.inference(xs[i, :T].cuda()).cpu()
            want = torch.from_numpy(g["free_after_%d" % i])
            assert y.shape == want.shape
            assert float((y - want).abs().max()) < tol
        mels, ol = model.inference_batch(xs.cuda(), il)                       # host-driven layout
        am = model.inference_batch(xs.cuda(), il, sync=False)                  # device-driven layout
        mels2, ol2 = am
        assert am.ok()
        for i in range(2):
            want = torch.from_numpy(g["free_after_%d" % i])
            assert int(ol[i]) == want.shape[0] and int(ol2[i]) == want.shape[0]
            assert float((mels[i, : want.shape[0]].cpu() - want).abs().max()) < tol
            assert torch.equal(mels[i, : want.shape[0]], mels2[i, : want.shape[0]])
            assert float(mels[i, want.shape[0]:].abs().max()) == 0.0 if mels.shape[1] > want.shape[0] else True
        # the packed form (what the multi-GPU gather ships) holds MEL frames: 2 per decoder frame, utterances back to back -- in both
        # layout modes -- and ShardedSynthesizer (the sharded path minus the collective on this one-GPU box) returns the padded result
        packed, olp = model.inference_batch(xs.cuda(), il, packed=True)
        assert torch.equal(olp, ol) and packed.shape == (int(ol.sum()), mels.shape[2])
        ap = model.inference_batch(xs.cuda(), il, packed=True, sync=False)
        packed2, olp2 = ap
        assert ap.ok() and torch.equal(olp2.cpu(), ol) and packed2.shape[0] >= int(ol.sum())
        s0 = 0
        for i in range(2):
            n = int(ol[i])
            assert torch.equal(packed[s0:s0 + n], mels[i, :n]) and torch.equal(packed2[s0:s0 + n], mels[i, :n])
            s0 += n
        from fastspeech2_amd.parallel import ShardedSynthesizer
        synth = ShardedSynthesizer(model)
        m1, o1 = synth(xs.cuda(), il)              # synchronous first call
        m2, o2 = synth(xs.cuda(), il)              # sync-free: device layout, capacities in decoder frames, result in mel frames
        assert synth.ok() and torch.equal(o1.cpu(), ol) and torch.equal(o2.cpu(), ol)
        Lm = mels.shape[1]
        assert torch.equal(m1[:, :Lm], mels) and torch.equal(m2[:, :Lm], mels) and float(m2[:, Lm:].abs().sum()) == 0.0


def test_packed_output_and_unpack_kernel(env):
    """`after_packed` (valid frames back to back, what the multi-GPU all-gather ships) and its inverse
    fs2_op_unpack_rows reproduce the padded output bit-for-bit."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch
    from fastspeech2_amd.parallel import unpack_rows
    b = make_batch("c3", B=9)
    with torch.no_grad():
        r = model._run(b["xs"].cuda(), b["ilens"], is_inference=True, d_override=b["ds"].cuda(), want=("after", "after_packed"))
    ol = r["olens"].tolist()
    assert r["after_packed"].shape == (sum(ol), 80)
    starts = [sum(ol[:i]) for i in range(len(ol))]
    for i, (s0, L) in enumerate(zip(starts, ol)):
        assert torch.equal(r["after_packed"][s0:s0 + L], r["after"][i, :L])
    back = unpack_rows(r["after_packed"], starts, ol, r["after"].shape[1])
    assert torch.equal(back, r["after"])


def test_g5_script_twin_eager_scripted_and_reloaded(golden_dir, tmp_path):
    """The TorchScript twin (reference utils/fastspeech2_script.py, export_torchscript.py:46-58) on the HIP path:
    eager forward, scripted forward, traced forward (with the reference's CPU example input) and a save/load round trip all
    reproduce the reference's mel -- and the saved archive runs in a process that never imports this package: the op
    fs2::twin_inference lives in libfs2_torch.so (C++)."""
    import subprocess
    import sys
    from fastspeech2_amd import default_hparams, N_PHONEME_SYMBOLS
    from fastspeech2_amd.fastspeech2_script import FeedForwardTransformer as Twin, OP_LIBRARY
    from fastspeech2_amd.synthetic import portable_state_dict, bias_durations
    g = np.load(golden_dir + "/g5_script_twin_t30.npz")
    twin = Twin(N_PHONEME_SYMBOLS, 80, default_hparams()).eval()
    twin.load_state_dict(bias_durations(portable_state_dict(twin.state_dict(), seed=5), 4.0))
    twin = twin.to("cuda:0")
    x = _t(g["x"])
    with torch.no_grad():
        eager = twin(x)
        direct = twin.inference(x)
        scripted = torch.jit.script(twin)
        s_out = scripted(x)
        traced = torch.jit.trace(twin, torch.ones(50, dtype=torch.int64))      # export_torchscript.py:53-56: a CPU example input
        t_out = traced(x)
        path = str(tmp_path / "twin.pt")
        scripted.save(path)
        r_out = torch.jit.load(path)(x)
        # a second twin alive in the same process: the op's handle cache (an LRU) serves both without rebuilding either
        twin2 = Twin(N_PHONEME_SYMBOLS, 80, default_hparams()).eval()
        twin2.load_state_dict(portable_state_dict(twin2.state_dict(), seed=6))
        twin2 = twin2.to("cuda:0")
        other = twin2(x)
        again = twin(x)
    d = _maxabs(eager, g["mel"])
    print("G5 twin mel max-abs %.2e (L=%d)" % (d, eager.shape[0]))
    assert eager.shape == tuple(g["mel"].shape) and d <= MEL_TOL
    for o in (direct, s_out, t_out, r_out, again):
        assert torch.equal(o, eager)
    assert other.shape[1] == 80 and not torch.equal(other[: min(len(other), len(eager))], eager[: min(len(other), len(eager))])
    # the archive in a fresh interpreter that loads ONLY torch + the op library
    out_path = str(tmp_path / "out.pt")
    code = ("import sys, torch\n"
            "assert not any(m.startswith('fastspeech2_amd') for m in sys.modules)\n"
            "torch.ops.load_library(%r)\n"
            "m = torch.jit.load(%r, map_location='cuda:0')\n"
            "x = torch.tensor(%r, dtype=torch.int64)\n"
            "y = m(x.cuda())\n"
            "assert not any(m.startswith('fastspeech2_amd') for m in sys.modules)\n"
            "torch.save(y.cpu(), %r)\n") % (OP_LIBRARY, path, g["x"].tolist(), out_path)
    env_ = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    res = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), env=env_, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert torch.equal(torch.load(out_path), eager.cpu())


def test_vocoder_hand_off(env):
    """inference.py:173-178 on the device: packed mels -> [1, 80, sum L]."""
    model, sd, cfg, O = env
    from fastspeech2_amd import vocoder_input
    from fastspeech2_amd.synthetic import make_batch
    b = make_batch("c3", B=5)
    with torch.no_grad():
        packed, ol = model.inference_batch(b["xs"].cuda(), b["ilens"], d_override=b["ds"].cuda(), packed=True)
    v = vocoder_input(packed)
    assert v.shape == (1, 80, int(ol.sum())) and torch.equal(v[0], packed.t())


def test_integration_md_usage_runs_end_to_end(env):
    """INTEGRATION.md section 2 (the generated struct mirrors + the documented usage) executed verbatim against the built library:
    one free-running utterance through fs2_create / fs2_load_weights / fs2_encode / fs2_decode equals the module's inference()
    bit for bit, and the oracle within the tolerance."""
    import os
    import re
    model, sd, cfg, O = env
    from fastspeech2_amd import _lib
    from fastspeech2_amd.synthetic import bias_durations
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"<!-- BEGIN (?:generated[^>]*|usage) -->\n```python\n(.*?)```", doc, re.S)
    assert len(blocks) == 2
    ns = {}
    for b in blocks:
        exec(compile(b, "INTEGRATION.md", "exec"), ns)
    sdb = bias_durations(sd, 4.0)
    ids = torch.from_numpy(np.random.RandomState(11).randint(1, 68, size=31)).to("cuda:0")
    dev_sd = {k: v.to("cuda:0") for k, v in sdb.items()}
    mel = ns["fs2_synthesize"](_lib.LIB_PATH, dev_sd, ids, precision=_lib.FS2_PREC_FP32)
    model.load_state_dict(sdb)
    try:
        with torch.no_grad():
            ref = model.inference(ids)
    finally:
        model.load_state_dict(sd)
    assert mel.shape == ref.shape and torch.equal(mel, ref)
    want = O.padded_forward(sdb, cfg, ids.cpu().unsqueeze(0), [ids.numel()], is_inference=True)["after"][0]
    assert _maxabs(mel, want) <= MEL_TOL


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "mix_mx"])
def test_c1_single_utterance_inference_vs_oracle(env, precision):
    """BASELINE config c1 -- the reference's production call (inference.py:111-130): ONE utterance of 80 phonemes through
    `inference(x)`, free-running, with LJSpeech-like durations.  First call: host-driven layout (frame count read back between encode
    and decode); second call: device-driven layout inside the capacities learnt from the first.  Both equal the CPU oracle's
    free-running result: durations and frame count exactly, mel within the tolerance; the two layouts are bit-identical."""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch, ljspeech_durations
    from tests.conftest import record_measurement
    b = make_batch("c1")
    x = b["xs"][0]
    sdl = ljspeech_durations(sd)
    o = O.padded_forward(sdl, cfg, x.unsqueeze(0), torch.tensor([80]), is_inference=True)
    L = int(o["olens"][0])
    assert 400 <= L <= 900, L                       # ~8 frames per phoneme
    model.load_state_dict(sdl)
    model.precision = precision
    model._frames_per_token = None                  # the first call of a fresh model is synchronous (host-driven layout)
    try:
        with torch.no_grad():
            m1 = model.inference(x.cuda())
            assert model._frames_per_token is not None
            m2 = model.inference(x.cuda())          # device-driven layout
            r = model._run(x.cuda().unsqueeze(0), torch.tensor([80]), is_inference=True, want=("after",))
    finally:
        model.precision = "fp32"
        model.load_state_dict(sd)
        model._frames_per_token = None
    assert m1.shape == (L, 80) and torch.equal(m1, m2)
    assert torch.equal(r["d_int"][0].cpu(), o["d_outs"][0])
    d = _maxabs(m1, o["after"][0])
    print("c1 [%s] %d frames, mel max-abs vs oracle %.2e" % (precision, L, d))
    record_measurement("c1_mel_maxabs_" + precision, d)
    assert d <= C3_TOL.get(precision, MEL_TOL)


@pytest.mark.parametrize("precision", ["bf16x3", "mix_mx"])
def test_c3_free_running_decisions_equal_the_oracles(env, precision):
    """BASELINE config c3 with NOTHING forced (what bench.py times): every data-dependent integer decision of the GPU path --
    durations clamp(round(exp(x) - 1), 0), frame counts, pitch / energy bucket indices -- equals the CPU oracle's own free-running
    decision on all 64 utterances, and the mels agree within the tolerance.  (duration_predictor.py:77-84, variance_predictor.py:154-159)"""
    model, sd, cfg, O = env
    from fastspeech2_amd.synthetic import make_batch, ljspeech_durations
    b = make_batch("c3")
    xs, il = b["xs"], b["ilens"]
    sdl = ljspeech_durations(sd)
    model.load_state_dict(sdl)
    model.precision = precision
    try:
        with torch.no_grad():
            r = model._run(xs.cuda(), il, is_inference=True, want=("after", "qe", "qp"))
    finally:
        model.precision = "fp32"
        model.load_state_dict(sd)
    after, qe, qp, dint = r["after"].cpu(), r["qe"].cpu().long(), r["qp"].cpu().long(), r["d_int"].cpu()
    tok = frames = 0
    worst = 0.0
    for i in range(xs.shape[0]):
        T = int(il[i])
        o = O.padded_forward(sdl, cfg, xs[i:i + 1, :T], il[i:i + 1], is_inference=True)
        L = int(o["olens"][0])
        assert torch.equal(dint[i, :T], o["d_outs"][0]), i
        assert int(r["olens"][i]) == L
        assert torch.equal(qe[i, :L], o["qe"][0]) and torch.equal(qp[i, :L], o["qp"][0]), i
        worst = max(worst, float((after[i, :L] - o["after"][0]).abs().max()))
        tok += T
        frames += L
    print("c3 free-running [%s]: %d phonemes -> %d frames (%.2f per phoneme), every decision equal, mel max-abs %.2e" % (precision, tok, frames, frames / tok, worst))
    assert 7.6 <= frames / tok <= 8.1               # LJSpeech-like (7.87)
    assert worst <= C3_TOL.get(precision, MEL_TOL)


# ---- static-scale stress (VERDICT r03 "what's weak" 1): every parity number above comes from ONE weight distribution (uniform +-1/sqrt(fan_in),
# LayerNorm affine 1 +- 0.1), while the bench default mix_mx keeps the cross terms of the FFN conv in e4m3 with STATIC per-tensor scales
# (kw from max |w|, ka from sqrt(D) max|gamma| + max|beta|; gemm_mx.h): one outlier weight or a large gamma costs every other element of
# that tensor exponent range.  Real checkpoints are unreachable here, so the heavy tails are synthesised.
def _stressed_state_dict(sd, kind, seed=123):
    rs = np.random.RandomState(seed)
    out = {}
    for k, v in sd.items():
        v = v.clone()
        is_mat = v.dim() >= 2 and v.is_floating_point() and "embed" not in k and not k.endswith("pe")
        if kind == "student_t" and is_mat:
            t = torch.from_numpy(rs.standard_t(3.0, size=tuple(v.shape))).float()
            v = t * (v.pow(2).mean().sqrt() / t.pow(2).mean().sqrt())              # same rms, Student-t(3) tails (max / rms ~ 30 .. 100)
        elif kind == "ffn_outliers" and k.endswith("w_1.weight"):
            m = torch.from_numpy(rs.uniform(size=tuple(v.shape)) < 1e-3)
            v = torch.where(m, v * 30.0, v)                                        # 0.1 % of the FFN conv weights x 30
        elif kind == "ln_affine" and v.dim() == 1 and ("norm" in k) and k.endswith("weight"):
            v = torch.from_numpy(rs.uniform(0.1, 8.0, size=tuple(v.shape))).float()
        elif kind == "ln_affine" and v.dim() == 1 and ("norm" in k) and k.endswith("bias"):
            v = torch.from_numpy(rs.uniform(-2.0, 2.0, size=tuple(v.shape))).float()
        out[k] = v
    return out


def test_c3_mix_mx4_really_runs_the_fp4_conv_and_stays_close_to_mix_mx(env):
    """mix_mx4 = mix_mx except that, where the decoder's activations travel as planes only (c3: yes), the FFN conv's cross terms run on e2m1 operands
    with one scale per frame and per output channel (gemm_pl_bf16<.., ARITH = 3>; the producing LayerNorm epilogue writes the row scales:
    gemm_row4_bf16 EPI 4).  The two modes must differ (the kernel really runs) by no more than the simulated cost of the fp4 cross terms, agree bit
    for bit in the encoder-side outputs (durations, bucket decisions: the variance adaptor is upstream of the decoder), and below the planes-only
    regime (c2) the mode IS mix_mx, bit for bit."""
    model = env[0]
    from fastspeech2_amd.synthetic import make_batch
    from tests.conftest import record_measurement
    b3, b2 = make_batch("c3"), make_batch("c2")
    out = {}
    try:
        for prec in ("mix_mx", "mix_mx4"):
            model.precision = prec
            with torch.no_grad():
                r3 = model._run(b3["xs"].cuda(), b3["ilens"], is_inference=True, d_override=b3["ds"].cuda(), want=("after", "qe", "qp"))
                r2 = model._run(b2["xs"].cuda(), b2["ilens"], is_inference=True, d_override=b2["ds"].cuda(), want=("after",))
            out[prec] = (r3["after"], r3["qe"], r3["qp"], r2["after"])
    finally:
        model.precision = "fp32"
    assert torch.equal(out["mix_mx"][1], out["mix_mx4"][1]) and torch.equal(out["mix_mx"][2], out["mix_mx4"][2])
    assert torch.equal(out["mix_mx"][3], out["mix_mx4"][3]), "below the planes-only regime mix_mx4 must be mix_mx"
    d = float((out["mix_mx"][0] - out["mix_mx4"][0]).abs().max())
    print("c3: mix_mx4 vs mix_mx mel max-abs %.2e" % d)
    record_measurement("c3_mix_mx4_vs_mix_mx_mel_maxabs", d)
    assert 1e-6 < d <= 3e-4, d


def test_postnet_in_the_mx_arithmetic_stays_within_1e5_of_split_bf16(env, fs2_option):
    """Round 6: in the mixed modes the Postnet's 512 -> 512 convolutions (3 of its 5 layers) run on the mx conv kernel -- fp16 main term + e4m3 cross terms with
    STATIC scales: their operands are tanh outputs (|x| <= 1, an exact a-priori bound) and BatchNorm-folded weights (reference core/modules.py:285-358); simulated at
    +6e-6 on the mel (tools/arith_sim_postnet.py).  FS2_POST_MX = 0 restores split-bf16: the two must differ (the kernel runs) by no more than 2e-5."""
    model = env[0]
    from fastspeech2_amd.synthetic import make_batch
    from tests.conftest import record_measurement
    b = make_batch("c2")
    model.precision = "mix_mx"
    try:
        outs = []
        for v in (1, 0):
            fs2_option("FS2_POST_MX", v)
            with torch.no_grad():
                outs.append(model._run(b["xs"].cuda(), b["ilens"], b["olens"], b["ds"].cuda(), b["es"].cuda(), b["ps"].cuda(), is_inference=False, want=("before", "after")))
    finally:
        model.precision = "fp32"
    assert torch.equal(outs[0]["before"], outs[1]["before"])          # everything in front of the Postnet is untouched
    d = float((outs[0]["after"] - outs[1]["after"]).abs().max())
    record_measurement("c2_postnet_mx_vs_split_bf16_mel_maxabs", d)
    assert 0.0 < d <= 2e-5, d


@pytest.mark.parametrize("kind", ["student_t", "ffn_outliers", "ln_affine"])
def test_static_scale_arithmetic_under_heavy_tailed_weights(env, kind, fs2_option):
    """mix_mx (and bf16x3 as the control) against the oracle with heavy-tailed weights / outlier FFN weights / wide LayerNorm affines, on
    the c2 batch and on the two longest utterances of c4.  The mel's own scale grows with these weights, so the bar is relative to it:
    2.5e-4 x max(1, max|mel| / 4) (4 = the mel range of the plain synthetic model); the numbers are recorded."""
    from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
    from fastspeech2_amd.synthetic import make_batch
    from tests.conftest import record_measurement
    _, sd0, cfg, O = env
    sd = _stressed_state_dict(sd0, kind)
    hp = default_hparams()
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
    model.load_state_dict(sd)
    model = model.to("cuda:0")
    b2 = make_batch("c2")
    b4 = make_batch("c4")
    long2 = torch.argsort(b4["olens"])[-2:]
    T4, L4 = int(b4["ilens"][long2].max()), int(b4["olens"][long2].max())
    sub4 = dict(xs=b4["xs"][long2][:, :T4], ilens=b4["ilens"][long2], ds=b4["ds"][long2][:, :T4], olens=b4["olens"][long2],
                es=b4["es"][long2][:, :L4], ps=b4["ps"][long2][:, :L4])
    for name, b in (("c2", b2), ("c4_longest2", sub4)):
        # teacher-forced (durations, energy, pitch given): the comparison measures arithmetic, not flipped bucket decisions
        o = O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])
        scale = float(o["after"].abs().max())
        errs = {}
        for prec in ("bf16x3", "mix_mx", "mix_mx4"):
            model.precision = prec
            if prec == "mix_mx4":      # (the fp4 conv runs in the planes-only regime only: force the row-complete kernels on these small batches)
                fs2_option("FS2_ROW8", 1)
                fs2_option("FS2_QKV8", 1)
            with torch.no_grad():
                r = model._run(b["xs"].cuda(), b["ilens"], b["olens"], b["ds"].cuda(), b["es"].cuda(), b["ps"].cuda(), is_inference=False, want=("after",))
            if prec == "mix_mx4":
                fs2_option("FS2_ROW8", -1)
                fs2_option("FS2_QKV8", -1)
            assert torch.isfinite(r["after"]).all(), (kind, name, prec)
            errs[prec] = _maxabs(r["after"], o["after"])
            record_measurement("stress_%s_%s_%s_mel_maxabs" % (kind, name, prec), errs[prec])
        record_measurement("stress_%s_%s_mel_scale" % (kind, name), scale)
        bar = 2.5e-4 * max(1.0, scale / 4.0)
        print("static-scale stress [%s, %s]: max|mel| %.2f; mel max-abs vs oracle: bf16x3 %.2e, mix_mx %.2e (bar %.2e)" % (kind, name, scale, errs["bf16x3"], errs["mix_mx"], bar))
        assert errs["mix_mx"] <= bar, (kind, name, errs, scale)
        # mix_mx4 spends part of the budget by design (simulated: +3.7e-4 / +4.7e-4 under the LayerNorm / Student-t sets): its bar is the north star's 1e-3 at the plain model's mel scale
        assert errs["mix_mx4"] <= 1e-3 * max(1.0, scale / 4.0), (kind, name, errs, scale)
