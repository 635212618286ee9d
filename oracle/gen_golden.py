"""Golden-vector generator: runs the REAL reference (imported from /root/reference, build container
only) on portable synthetic weights/inputs and writes small fixtures to tests/golden/*.npz.

TEST INFRASTRUCTURE ONLY.  The reference's Python cannot travel to the GPU box, so what travels are
these vectors (inputs + expected outputs); the weights are NOT stored, both sides regenerate them
with fastspeech2_amd.synthetic.portable_state_dict(seed).  The script also checks the oracle
restatement (oracle/fs2_oracle.py) against the reference while it is at it and prints the diffs.

Usage (in the build container):  python oracle/gen_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def import_reference():
    """Stub the five third-party modules the reference imports at module top but the path never uses
    (SURVEY.md Appendix C), then import its model."""
    def stub(n, **a):
        m = types.ModuleType(n)
        m.__dict__.update(a)
        sys.modules[n] = m
    stub("typeguard", check_argument_types=lambda *a, **k: True)
    stub("librosa")
    stub("nltk")
    stub("g2p_en", G2p=type("G2p", (), {"__call__": lambda s, t: []}))
    stub("unidecode", unidecode=lambda s: s)
    stub("inflect", engine=lambda: None)
    sys.path.insert(0, REF)
    from utils.hparams import HParam
    from dataset.texts import valid_symbols
    from fastspeech import FeedForwardTransformer
    return HParam(os.path.join(REF, "configs/default.yaml")), len(valid_symbols), FeedForwardTransformer


def small_batch(seed, tlens, dmax):
    rs = np.random.RandomState(seed)
    B, Tmax = len(tlens), max(tlens)
    xs = np.zeros((B, Tmax), np.int64)
    ds = np.zeros((B, Tmax), np.int64)
    for b, T in enumerate(tlens):
        xs[b, :T] = rs.randint(1, 68, size=T)
        ds[b, :T] = rs.randint(0, dmax + 1, size=T)      # zeros inside exercise the skip rule
        ds[b, 0] = max(ds[b, 0], 1)
    olens = ds.sum(1)
    Lmax = int(olens.max())
    es = np.zeros((B, Lmax), np.float32)
    ps = np.zeros((B, Lmax), np.float32)
    for b in range(B):
        L = int(olens[b])
        es[b, :L] = rs.uniform(0.0, 130.5, size=L)
        p = rs.uniform(71.0, 676.0, size=L)
        p[rs.uniform(size=L) < 0.3] = 0.0
        ps[b, :L] = p
    t = lambda a: torch.from_numpy(a)
    return dict(xs=t(xs), ilens=torch.tensor(tlens), ds=t(ds), olens=t(olens), es=t(es), ps=t(ps))


def main():
    from fastspeech2_amd.synthetic import portable_state_dict, bias_durations
    from oracle import fs2_oracle as O
    hp, idim, Ref = import_reference()
    odim = hp.audio.num_mels
    torch.manual_seed(0)
    ref = Ref(idim, odim, hp).eval()
    sd = portable_state_dict(ref.state_dict(), seed=0)
    ref.load_state_dict(sd)
    cfg = O.config_from_hp(hp, idim, odim)
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    np_ = lambda t: t.detach().cpu().numpy()
    worst = 0.0

    def cmp(name, a, b):
        nonlocal worst
        d = float((a.float() - b.float()).abs().max()) if a.numel() else 0.0
        worst = max(worst, d)
        print("  oracle vs reference %-14s max-abs %.3e" % (name, d))

    with torch.no_grad():
        # ---- G1: teacher-forced single utterance, with intermediates captured by forward hooks ----
        b1 = small_batch(11, [24], 12)
        inter = {}
        hk = [ref.encoder.register_forward_hook(lambda m, i, o: inter.__setitem__("encoder_out", o[0])),
              ref.decoder.register_forward_hook(lambda m, i, o: inter.__setitem__("decoder_out", o[0])),
              ref.length_regulator.register_forward_hook(lambda m, i, o: inter.__setitem__("lr_out", o))]
        r = ref._forward(b1["xs"], b1["ilens"], b1["olens"], b1["ds"], b1["es"], b1["ps"], is_inference=False)
        for h in hk:
            h.remove()
        qe = torch.bucketize(b1["es"], ref.energy_predictor.energy_bins)
        qp = torch.bucketize(b1["ps"], ref.pitch_predictor.pitch_bins)
        lr_idx = torch.repeat_interleave(torch.arange(24), b1["ds"][0])
        assert torch.equal(inter["lr_out"][0], inter["encoder_out"][0][lr_idx])
        L = int(b1["olens"][0])
        rows = np.array([0, L // 2, L - 1])
        np.savez_compressed(os.path.join(out_dir, "g1_teacher_b1.npz"), xs=np_(b1["xs"]), ilens=np_(b1["ilens"]),
                            olens=np_(b1["olens"]), ds=np_(b1["ds"]), es=np_(b1["es"]), ps=np_(b1["ps"]),
                            before=np_(r[0]), after=np_(r[1]), d_outs=np_(r[2]), e_outs=np_(r[3]), p_outs=np_(r[4]),
                            encoder_out=np_(inter["encoder_out"]), lr_index=np_(lr_idx), qe=np_(qe), qp=np_(qp),
                            decoder_rows=rows, decoder_out_rows=np_(inter["decoder_out"][0][rows]))
        o = O.padded_forward(sd, cfg, b1["xs"], b1["ilens"], b1["olens"], b1["ds"], b1["es"], b1["ps"])
        print("G1 (L=%d)" % L)
        for k, i in (("before", 0), ("after", 1), ("d_outs", 2), ("e_outs", 3), ("p_outs", 4)):
            cmp(k, o[k], r[i])
        cmp("encoder_out", o["encoder_out"], inter["encoder_out"])
        cmp("decoder_out", o["decoder_out"], inter["decoder_out"])

        # ---- G2: padded batch of 3, teacher-forced (padded-compat semantics) + forward() losses ----
        b2 = small_batch(12, [16, 11, 7], 6)
        r = ref._forward(b2["xs"], b2["ilens"], b2["olens"], b2["ds"].clone(), b2["es"], b2["ps"], is_inference=False)
        ys = torch.from_numpy(np.random.RandomState(13).uniform(-2, 2, size=tuple(r[0].shape)).astype(np.float32))
        loss, rep = ref(b2["xs"], b2["ilens"], ys, b2["olens"], b2["ds"].clone(), b2["es"], b2["ps"])
        rep_names = [list(d.keys())[0] for d in rep]
        rep_vals = np.array([list(d.values())[0] for d in rep], np.float64)
        np.savez_compressed(os.path.join(out_dir, "g2_teacher_padded_b3.npz"), xs=np_(b2["xs"]), ilens=np_(b2["ilens"]),
                            olens=np_(b2["olens"]), ds=np_(b2["ds"]), es=np_(b2["es"]), ps=np_(b2["ps"]), ys=np_(ys),
                            before=np_(r[0]), after=np_(r[1]), d_outs=np_(r[2]), e_outs=np_(r[3]), p_outs=np_(r[4]),
                            loss=np.float64(loss.item()), report_names=np.array(rep_names), report_values=rep_vals)
        o = O.padded_forward(sd, cfg, b2["xs"], b2["ilens"], b2["olens"], b2["ds"], b2["es"], b2["ps"])
        print("G2 (olens=%s)" % b2["olens"].tolist())
        for k, i in (("before", 0), ("after", 1), ("d_outs", 2), ("e_outs", 3), ("p_outs", 4)):
            cmp(k, o[k], r[i])
        ol, orep = O.loss_report(o, ys, b2["ilens"], b2["olens"], b2["ds"], b2["es"], b2["ps"])
        print("  loss reference %.6f oracle %.6f" % (loss.item(), ol.item()))

        # ---- G6: the same three utterances one at a time (per-utterance semantics) ----
        after_pu, before_pu = [], []
        for b in range(3):
            T, Lb = int(b2["ilens"][b]), int(b2["olens"][b])
            rb = ref._forward(b2["xs"][b:b + 1, :T], b2["ilens"][b:b + 1], b2["olens"][b:b + 1], b2["ds"][b:b + 1, :T].clone(),
                              b2["es"][b:b + 1, :Lb], b2["ps"][b:b + 1, :Lb], is_inference=False)
            before_pu.append(np_(rb[0][0]))
            after_pu.append(np_(rb[1][0]))
        np.savez_compressed(os.path.join(out_dir, "g6_teacher_per_utt_b3.npz"),
                            **{"after_%d" % b: a for b, a in enumerate(after_pu)},
                            **{"before_%d" % b: a for b, a in enumerate(before_pu)})
        o = O.per_utterance_forward(sd, cfg, b2["xs"], b2["ilens"], b2["ds"], b2["es"], b2["ps"])
        for b in range(3):
            cmp("per-utt after %d" % b, o["after"][b, : after_pu[b].shape[0]], torch.from_numpy(after_pu[b]))

        # ---- G3: free-running inference(x), durations made non-trivial through the bias ----
        sd3 = bias_durations(sd, 4.0)
        ref.load_state_dict(sd3)
        x3 = torch.from_numpy(np.random.RandomState(14).randint(1, 68, size=24).astype(np.int64))
        r3 = ref._forward(x3.unsqueeze(0), torch.tensor([24]), is_inference=True)
        mel = ref.inference(x3)
        assert torch.equal(mel, r3[1][0])
        np.savez_compressed(os.path.join(out_dir, "g3_inference_t24.npz"), x=np_(x3), d_outs=np_(r3[2]),
                            olens=np.array([mel.shape[0]]), mel=np_(mel), before=np_(r3[0][0]),
                            qe=np_(r3[3].argmax(-1)), qp=np_(r3[4].argmax(-1)))
        o = O.padded_forward(sd3, cfg, x3.unsqueeze(0), torch.tensor([24]), is_inference=True)
        print("G3 (L=%d, durations %s)" % (mel.shape[0], r3[2][0].tolist()))
        assert torch.equal(o["d_outs"], r3[2]), "duration mismatch"
        cmp("mel", o["after"][0], mel)
        ref.load_state_dict(sd)

        # ---- G4: known answers of the primitives the reference relies on ----
        sys.path.insert(0, REF)
        from utils.util import make_pad_mask, make_non_pad_mask
        from core.duration_modeling.length_regulator import LengthRegulator
        ebins, pbins = ref.energy_predictor.energy_bins, ref.pitch_predictor.pitch_bins
        xe = torch.tensor([-1.0, float(ebins[0]), float(ebins[0]) + 1e-3, float(ebins[100]), float(ebins[254]),
                           float(ebins[254]) + 1.0, float("nan"), 0.0, 65.0])
        xp = torch.tensor([0.0, 71.0, float(pbins[1]), float(pbins[200]), 700.0, float("nan"), 300.0])
        yr = torch.tensor([-0.2, 0.0, math_log(1.5), math_log(2.5), math_log(3.5), math_log(4.5), 1.7, 3.3, -5.0])
        dr = torch.clamp(torch.round(yr.exp() - 1.0), min=0).long()
        lr = LengthRegulator()
        hs = torch.arange(1, 3 * 5 * 4 + 1, dtype=torch.float32).view(3, 5, 4)
        dsl = torch.tensor([[1, 2, 3, 0, 1], [0, 0, 0, 0, 0], [2, 0, 0, 9, 9]])
        ill = torch.tensor([5, 4, 3])
        lro = lr(hs, dsl.clone(), ill)
        att = ref.encoder.encoders_[0].self_attn
        xa = torch.from_numpy(np.random.RandomState(15).uniform(-1, 1, size=(2, 6, 256)).astype(np.float32))
        valid = make_non_pad_mask([6, 3])
        am = valid.unsqueeze(-2) & valid.unsqueeze(-1)
        ao = att(xa, xa, xa, am)
        pe256, pe384 = ref.encoder.embed[1].pe[0], ref.decoder.embed[4].pe[0]
        np.savez_compressed(os.path.join(out_dir, "g4_known_answers.npz"),
                            energy_bins=np_(ebins), pitch_bins=np_(pbins), xe=np_(xe), qe=np_(torch.bucketize(xe, ebins)),
                            xp=np_(xp), qp=np_(torch.bucketize(xp, pbins)), dur_log=np_(yr), dur_int=np_(dr),
                            lr_hs=np_(hs), lr_ds=np_(dsl), lr_ilens=np_(ill), lr_out=np_(lro),
                            pad_mask_5_3_2=np_(make_pad_mask([5, 3, 2])), attn_x=np_(xa), attn_out=np_(ao),
                            pe_rows=np.array([0, 1, 4999]), pe256=np_(pe256[[0, 1, 4999]]), pe384=np_(pe384[[0, 1, 4999]]))
        # ---- G5: the TorchScript twin (utils/fastspeech2_script.py): different architecture, forward(x) ----
        from utils import fastspeech2_script as ref_script
        twin = ref_script.FeedForwardTransformer(idim, odim, hp).eval()
        sd5 = bias_durations(portable_state_dict(twin.state_dict(), seed=5), 4.0)
        twin.load_state_dict(sd5)
        x5 = torch.from_numpy(np.random.RandomState(16).randint(1, 68, size=30).astype(np.int64))
        mel5 = twin(x5)
        keys5 = sorted(sd5.keys())
        np.savez_compressed(os.path.join(out_dir, "g5_script_twin_t30.npz"), x=np_(x5), mel=np_(mel5),
                            keys=np.array(keys5), shapes=np.array([str(tuple(sd5[k].shape)) for k in keys5]))
        cfg5 = O.config_from_hp(hp, idim, odim, script_twin=True)
        o5 = O.padded_forward(sd5, cfg5, x5.unsqueeze(0), torch.tensor([30]), is_inference=True)
        print("G5 twin (L=%d, %d state-dict keys, %d params)" % (mel5.shape[0], len(keys5), sum(p.numel() for p in twin.parameters())))
        cmp("twin mel", o5["after"][0], mel5)

    print("worst oracle-vs-reference max-abs: %.3e" % worst)
    assert worst < 2e-5, "oracle restatement drifted from the reference"
    print("fixtures written to", out_dir)
    for f in sorted(os.listdir(out_dir)):
        print("  %-32s %7.1f KB" % (f, os.path.getsize(os.path.join(out_dir, f)) / 1024))


BLOCK_VARIANTS = {
    "pre_ln": dict(encoder_normalize_before=True, decoder_normalize_before=True),
    "concat_after": dict(encoder_concat_after=True, decoder_concat_after=True),
    "pre_ln_concat": dict(encoder_normalize_before=True, decoder_normalize_before=True, encoder_concat_after=True, decoder_concat_after=True),
    "enc_pre_ln_dec_concat": dict(encoder_normalize_before=True, decoder_concat_after=True),
}


def block_variants():
    """G7: the non-default FFT-block variants of the reference (core/encoder.py:53-71,201-202: normalize_before, concat_after), real
    reference run on a small teacher-forced batch per variant; also checks the oracle against each."""
    from fastspeech2_amd.synthetic import portable_state_dict
    from oracle import fs2_oracle as O
    hp, idim, Ref = import_reference()
    odim = hp.audio.num_mels
    out_dir = os.path.join(ROOT, "tests", "golden")
    b = small_batch(17, [19, 9], 7)
    arrays = dict(xs=b["xs"].numpy(), ilens=b["ilens"].numpy(), olens=b["olens"].numpy(), ds=b["ds"].numpy(), es=b["es"].numpy(), ps=b["ps"].numpy(),
                  names=np.array(sorted(BLOCK_VARIANTS)))
    worst = 0.0
    for name in sorted(BLOCK_VARIANTS):
        for k, v in BLOCK_VARIANTS[name].items():
            setattr(hp.model, k, v)
        torch.manual_seed(0)
        ref = Ref(idim, odim, hp).eval()
        sd = portable_state_dict(ref.state_dict(), seed=21)
        ref.load_state_dict(sd)
        cfg = O.config_from_hp(hp, idim, odim)
        with torch.no_grad():
            outs = []
            for i in range(2):          # one utterance at a time: per-utterance semantics (what the HIP path computes by default)
                T, L = int(b["ilens"][i]), int(b["olens"][i])
                r = ref._forward(b["xs"][i:i + 1, :T], b["ilens"][i:i + 1], b["olens"][i:i + 1], b["ds"][i:i + 1, :T].clone(),
                                 b["es"][i:i + 1, :L], b["ps"][i:i + 1, :L], is_inference=False)
                outs.append(r)
                arrays["%s_after_%d" % (name, i)] = r[1][0].numpy()
                arrays["%s_before_%d" % (name, i)] = r[0][0].numpy()
                arrays["%s_d_outs_%d" % (name, i)] = r[2][0].numpy()
        o = O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])
        for i in range(2):
            L = int(b["olens"][i])
            d = float((o["after"][i, :L] - outs[i][1][0]).abs().max())
            worst = max(worst, d)
            print("G7 %-22s utterance %d (L=%d): oracle vs reference mel max-abs %.3e" % (name, i, L, d))
        for k in BLOCK_VARIANTS[name]:
            setattr(hp.model, k, False)
    np.savez_compressed(os.path.join(out_dir, "g7_block_variants_b2.npz"), **arrays)
    assert worst < 2e-5, "oracle restatement of the block variants drifted from the reference"
    print("G7 written; worst %.3e" % worst)


def reduction_factor_fixture():
    """G8: reduction_factor = 2 (reference fastspeech.py:153,228-230: feat_out emits r mel frames per decoder frame, the Postnet runs
    on L*r frames).  Real reference, one utterance at a time, teacher-forced and free-running; also checks the oracle."""
    from fastspeech2_amd.synthetic import portable_state_dict, bias_durations
    from oracle import fs2_oracle as O
    hp, idim, Ref = import_reference()
    odim = hp.audio.num_mels
    out_dir = os.path.join(ROOT, "tests", "golden")
    hp.model.reduction_factor = 2
    torch.manual_seed(0)
    ref = Ref(idim, odim, hp).eval()
    sd = bias_durations(portable_state_dict(ref.state_dict(), seed=23), math_log(1 + 3.0))
    ref.load_state_dict(sd)
    cfg = O.config_from_hp(hp, idim, odim)
    b = small_batch(19, [19, 9], 7)
    arrays = dict(xs=b["xs"].numpy(), ilens=b["ilens"].numpy(), olens=b["olens"].numpy(), ds=b["ds"].numpy(), es=b["es"].numpy(), ps=b["ps"].numpy(),
                  reduction_factor=np.array(2))
    worst = 0.0
    with torch.no_grad():
        o = O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])
        f = O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], is_inference=True)
        for i in range(2):
            T, L = int(b["ilens"][i]), int(b["olens"][i])
            r = ref._forward(b["xs"][i:i + 1, :T], b["ilens"][i:i + 1], b["olens"][i:i + 1], b["ds"][i:i + 1, :T].clone(),
                             b["es"][i:i + 1, :L], b["ps"][i:i + 1, :L], is_inference=False)
            arrays["tf_after_%d" % i] = r[1][0].numpy()
            arrays["tf_before_%d" % i] = r[0][0].numpy()
            assert r[1].shape[1] == 2 * L
            d = float((o["after"][i, :2 * L] - r[1][0]).abs().max())
            worst = max(worst, d)
            print("G8 teacher-forced utterance %d (L=%d -> %d mel frames): oracle vs reference max-abs %.3e" % (i, L, 2 * L, d))
            y = ref.inference(b["xs"][i, :T])
            arrays["free_after_%d" % i] = y.numpy()
            Lf = int(f["olens"][i])
            assert y.shape[0] == 2 * Lf, (y.shape, Lf)
            d = float((f["after"][i, :2 * Lf] - y).abs().max())
            worst = max(worst, d)
            print("G8 free-running   utterance %d (%d decoder frames -> %d mel frames): oracle vs reference max-abs %.3e" % (i, Lf, 2 * Lf, d))
    hp.model.reduction_factor = 1
    np.savez_compressed(os.path.join(out_dir, "g8_reduction_factor2_b2.npz"), **arrays)
    assert worst < 2e-5, "oracle restatement of reduction_factor = 2 drifted from the reference"
    print("G8 written; worst %.3e" % worst)


def weighted_masking_fixture():
    """G9: the reference's forward() with hp.model.use_weighted_masking = True (fastspeech.py:308-333) -- which only runs with
    use_masking = False, because the weights are built from ys.size(2) and a masked-selected ys is 1-D.  The real reference's loss and
    its 7 report values on the G2 batch."""
    from fastspeech2_amd.synthetic import portable_state_dict
    from oracle import fs2_oracle as O
    hp, idim, Ref = import_reference()
    hp.model.use_masking = False
    hp.model.use_weighted_masking = True
    odim = hp.audio.num_mels
    ref = Ref(idim, odim, hp).eval()
    sd = portable_state_dict(ref.state_dict(), seed=0)
    ref.load_state_dict(sd)
    cfg = O.config_from_hp(hp, idim, odim)
    np_ = lambda t: t.detach().cpu().numpy()
    with torch.no_grad():
        b2 = small_batch(12, [16, 11, 7], 6)
        r = ref._forward(b2["xs"], b2["ilens"], b2["olens"], b2["ds"].clone(), b2["es"], b2["ps"], is_inference=False)
        ys = torch.from_numpy(np.random.RandomState(13).uniform(-2, 2, size=tuple(r[0].shape)).astype(np.float32))
        loss, rep = ref(b2["xs"], b2["ilens"], ys, b2["olens"], b2["ds"].clone(), b2["es"], b2["ps"])
        o = O.padded_forward(sd, cfg, b2["xs"], b2["ilens"], b2["olens"], b2["ds"], b2["es"], b2["ps"])
        ol, orep = O.loss_report(o, ys, b2["ilens"], b2["olens"], b2["ds"], b2["es"], b2["ps"], use_masking=False, use_weighted_masking=True)
    rep_names = [list(d.keys())[0] for d in rep]
    rep_vals = np.array([list(d.values())[0] for d in rep], np.float64)
    ovals = np.array([list(d.values())[0] for d in orep], np.float64)
    print("G9 weighted masking: loss reference %.6f oracle %.6f; report max rel diff %.2e" % (loss.item(), ol.item(), float(np.abs(ovals / rep_vals - 1).max())))
    assert np.allclose(ovals, rep_vals, rtol=1e-6)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g9_weighted_masking_b3.npz"), xs=np_(b2["xs"]), ilens=np_(b2["ilens"]),
                        olens=np_(b2["olens"]), ds=np_(b2["ds"]), es=np_(b2["es"]), ps=np_(b2["ps"]), ys=np_(ys),
                        loss=np.float64(loss.item()), report_names=np.array(rep_names), report_values=rep_vals)


def math_log(v):
    import math
    return math.log(v)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g7":
        block_variants()          # only the block-variant fixture (leaves G1-G6 untouched)
    elif len(sys.argv) > 1 and sys.argv[1] == "g8":
        reduction_factor_fixture()
    elif len(sys.argv) > 1 and sys.argv[1] == "g9":
        weighted_masking_fixture()
    else:
        main()
        block_variants()
        reduction_factor_fixture()
        weighted_masking_fixture()
