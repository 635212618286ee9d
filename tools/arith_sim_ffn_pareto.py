#!/usr/bin/env python3
"""Pareto table of the FFN convolution's arithmetic: MFMA-equivalents per product against mel error, by CPU simulation, BEFORE
any kernel is written.  The dominant kernel (dec.ffn1 = the 9-tap conv w_1, reference core/modules.py:237-248) computes, in mix_mx,

    a.w = ah.wh (fp16 MFMA: 1.0 equivalent)  +  ra.wh + ah.rw (block-scaled 8-bit MFMA, K = 128 in 32 cycles: 0.5 equivalents each)      = 2.0

with ah = fp16(a), ra = a - ah (likewise w) and measures 3.0e-5 on the mel against a 1e-3 bar.  Every cheaper point keeps the fp16 main term and
makes the cross terms cheaper: one of them dropped (0.5 saved), both in a 6- or 4-bit element format (v_mfma_scale_f32_16x16x128_f8f6f4 runs K = 128
in 16 cycles when BOTH operands are fp6 / fp4: 0.25 each), on every second tap only, or none (mix_f16x1).  fp8 operands keep today's STATIC
per-tensor scales (2^ka from the LayerNorm bound, 2^kw from max |w|, residuals x 2^11); fp6 / fp4 cannot (their 2-3 binades of range flush the
residuals of small activations), so they are simulated as real MX blocks: one E8M0 scale per 32 consecutive channels, chosen from the block's
maximum as the OCP MX spec does -- for the activations that would be a shuffle reduction in the producing LayerNorm epilogue.
The oracle's _ffn is replaced for BOTH stacks (encoder and decoder run the same kernel); w_2 stays exact, so the numbers are what the w_1
arithmetic ADDS to the fp32 oracle.  Products accumulate in fp64 (the kernel: fp32).  Test infrastructure: imports oracle/.

  python tools/arith_sim_ffn_pareto.py [--frames-from-utterances 4]  > profiles/r06_ffn_arith_pareto.txt
"""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS   # noqa: E402
from fastspeech2_amd.synthetic import portable_state_dict, make_batch                     # noqa: E402
from oracle import fs2_oracle as O                                                        # noqa: E402
from tools.arith_sim_residual import hostile_weights                                      # noqa: E402

FORMATS = {"e4m3": (4, 3, 448.0, 8), "e5m2": (5, 2, 57344.0, 15), "e3m2": (3, 2, 28.0, 4), "e2m3": (2, 3, 7.5, 2), "e2m1": (2, 1, 6.0, 2)}      # (E, M, max, emax)


def minifloat(x, fmt):
    """round-to-nearest-even onto the finite values of a small float format (saturating), subnormals included"""
    E, M, vmax, emax = FORMATS[fmt]
    emin = -14 if fmt == "e5m2" else emax - (2 ** E - 2)      # smallest normal exponent (e4m3fn: -6; e5m2, IEEE-like: -14; e3m2: -2; e2m3 / e2m1: 0)
    ax = x.abs().clamp(max=vmax)
    e = torch.floor(torch.log2(ax.clamp(min=2.0 ** (emin - M - 2)))).clamp(min=emin)
    step = torch.exp2(e - M)
    return torch.sign(x) * (torch.round(ax / step) * step).clamp(max=vmax)


def q_static(x, fmt, k):
    """x 2^k rounded to fmt, returned in x's own scale"""
    return minifloat(x * 2.0 ** k, fmt) * 2.0 ** -k


def q_block(x, fmt, dim):
    """MX block format: one power-of-two scale per 32 consecutive elements along `dim`, from the block maximum (OCP MX: 2^(floor(log2 max) - emax))"""
    emax = FORMATS[fmt][3]
    xs = x.movedim(dim, -1)
    shp = xs.shape
    xb = xs.reshape(*shp[:-1], shp[-1] // 32, 32)
    mx = xb.abs().amax(dim=-1, keepdim=True).clamp(min=2.0 ** -120)
    sc = torch.exp2(torch.floor(torch.log2(mx)) - emax)
    return (minifloat(xb / sc, fmt) * sc).reshape(shp).movedim(-1, dim)


def q_row(x, fmt, dims):
    """one power-of-two scale per slice: the maximum over `dims` (activations: all channels of a frame; weights: all channels and taps of an output
    channel) -- in the MFMA loop such a scale is a per-lane CONSTANT (a lane's row / column), not a per-block operand"""
    emax = FORMATS[fmt][3]
    mx = x.abs().amax(dim=dims, keepdim=True).clamp(min=2.0 ** -120)
    sc = torch.exp2(torch.floor(torch.log2(mx)) - emax)
    return minifloat(x / sc, fmt) * sc


def q_row2(x, fmt, dims, rule):
    """q_row with the scale rule spelled out: "ocp" = 2^(floor(log2 max) - emax) (max / scale in [2^emax, 2^(emax+1)): the top quarter saturates);
    "fit" = 2^(floor(log2(max / vmax)) + 1) (max / scale in [vmax / 2, vmax): nothing saturates, up to one binade of range unused)"""
    E, M, vmax, emax = FORMATS[fmt]
    mx = x.abs().amax(dim=dims, keepdim=True).clamp(min=2.0 ** -100)
    sc = torch.exp2(torch.floor(torch.log2(mx)) - emax) if rule == "ocp" else torch.exp2(torch.floor(torch.log2(mx / vmax)) + 1.0)
    return minifloat(x / sc, fmt) * sc


def q_shared(x_main, x_res, fmt, dims, rule="fit", pre=2048.0, clip=0.0):
    """what the mix_mx4 kernels do (round 6): ONE power-of-two scale per slice for the fp16 part AND its residual x 2^11 (so that both cross terms
    carry the same pair of scale bytes): scale = 2^(floor(log2(M / vmax)) + 1), M = max |x_main| over `dims` -- nothing saturates.  Returns the two
    quantised tensors in their own scales."""
    vmax, emax = FORMATS[fmt][2], FORMATS[fmt][3]
    mx = x_main.abs().amax(dim=dims, keepdim=True).clamp(min=2.0 ** -100)
    if clip > 0.0:      # a robust maximum: an outlier saturates instead of taking the resolution of its whole slice
        mx = torch.minimum(mx, clip * x_main.abs().mean(dim=dims, keepdim=True)).clamp(min=2.0 ** -100)
    sc = torch.exp2(torch.floor(torch.log2(mx)) - emax) if rule == "ocp" else torch.exp2(torch.floor(torch.log2(mx / vmax)) + 1.0)
    return minifloat(x_main / sc, fmt) * sc, minifloat(x_res * pre / sc, fmt) * sc / pre


def exp_for(bound):
    return int(math.floor(math.log2(448.0 / max(bound, 1e-30))))


# scheme = (MFMA-equivalents, label, cross term ra.wh, cross term ah.rw, taps that get cross terms); a cross term = (format, "static" | "block") or None
E4, E5 = ("e4m3", "static"), ("e5m2", "static")
F6, F6S, F4, F6W = ("e2m3", "block"), ("e2m3", "static"), ("e2m1", "block"), ("e3m2", "block")
F4S = ("e2m1", "shared")
F4SO, F4S12, F4SEP, F4SEPO = ("e2m1", "shared-ocp"), ("e2m1", "shared-4096"), ("e2m1", "sep-fit"), ("e2m1", "sep-ocp")
F4WB = ("e2m1", "shared-ocp-wblock16")
F4AB = ("e2m1", "shared-ocp-wblock16-ablock16")
F4C8, F4C16, F4CW = ("e2m1", "shared-ocp-clip8"), ("e2m1", "shared-ocp-clip16"), ("e2m1", "shared-ocp-clipw8")      # the scheme built as precision mode mix_mx4: per frame / per output channel, fp16 part and residual x 2^11 under ONE scale
F4R, F6R, F4H = ("e2m1", "row"), ("e2m3", "row"), ("e2m1", "hybrid")      # row: one scale per frame / per output channel; hybrid: activations per 32-block, weights per output channel
SCHEMES = [
    (2.00, "today: both cross terms e4m3, static scales", E4, E4, "all"),
    (2.00, "both cross terms e5m2, static scales", E5, E5, "all"),
    (1.75, "ra.wh e4m3 static + ah.rw fp6 e2m3 MX blocks", E4, F6, "all"),
    (1.75, "ra.wh fp6 e2m3 MX blocks + ah.rw e4m3 static", F6, E4, "all"),
    (1.50, "both cross terms fp6 e2m3, MX blocks of 32", F6, F6, "all"),
    (1.50, "both cross terms fp6 e3m2, MX blocks of 32", F6W, F6W, "all"),
    (1.50, "both cross terms fp6 e2m3, static scales", F6S, F6S, "all"),
    (1.50, "both cross terms fp4 e2m1, MX blocks of 32", F4, F4, "all"),
    (1.50, "both cross terms fp4 e2m1, one scale per frame / per output channel", F4R, F4R, "all"),
    (1.50, "first candidate: ONE scale per frame / output channel that FITS the maximum (no saturation)", F4S, F4S, "all"),
    (1.50, "second candidate (first build): one shared scale per frame / output channel, OCP rule (top quarter-binade saturates)", F4SO, F4SO, "all"),
    (1.50, "variant: one shared scale, residual x 2^12", F4S12, F4S12, "all"),
    (1.50, "second build: activations one scale per frame, weights one per 16-channel block of a tap (the B operand's native block), OCP rule", F4WB, F4WB, "all"),
    (1.50, "mix_mx4 AS BUILT: one scale per 16-channel block on BOTH sides (frame x 16 channels; weight row x tap x 16 channels), OCP rule", F4AB, F4AB, "all"),
    (1.50, "robust: as built, maxima clipped at 8 x the slice's mean |x| (weights and activations)", F4C8, F4C8, "all"),
    (1.50, "robust: as built, maxima clipped at 16 x mean", F4C16, F4C16, "all"),
    (1.50, "robust: as built, WEIGHT maxima clipped at 8 x mean only", F4CW, F4CW, "all"),
    (1.50, "variant: separate scales for the fp16 part and the residual, no saturation", F4SEP, F4SEP, "all"),
    (1.50, "variant: separate scales, OCP rule", F4SEPO, F4SEPO, "all"),
    (1.50, "both cross terms fp6 e2m3, one scale per frame / per output channel", F6R, F6R, "all"),
    (1.50, "both fp4 e2m1: activations MX blocks, weights per output channel", F4H, F4H, "all"),
    (1.50, "ra.wh only (weights rounded to fp16 once), e4m3", E4, None, "all"),
    (1.50, "ah.rw only (activations rounded to fp16 once), e4m3", None, E4, "all"),
    (1.50, "both cross terms e4m3 on every second tap", E4, E4, "even"),
    (1.25, "ra.wh only, fp6 e2m3 MX blocks", F6, None, "all"),
    (1.25, "both cross terms fp6 e2m3 MX on every second tap", F6, F6, "even"),
    (1.00, "no cross terms (mix_f16x1)", None, None, "all"),
]


def make_ffn(scheme, stats):
    _, _, c_ra, c_rw, taps = scheme

    def ffn(sd_, p, x, cfg_):
        w1, b1 = sd_[p + ".w_1.weight"], sd_[p + ".w_1.bias"]
        w2, b2 = sd_[p + ".w_2.weight"], sd_[p + ".w_2.bias"]
        k = w1.shape[-1]
        lnp = p.replace(".feed_forward", ".norm1")
        D = x.shape[-1]
        ka = exp_for(math.sqrt(D) * float(sd_[lnp + ".weight"].abs().max()) + float(sd_[lnp + ".bias"].abs().max()))
        kw = exp_for(float(w1.abs().max()))
        a = x.transpose(1, 2)                                  # [B, C, T]
        ah, wh = a.half().float(), w1.half().float()
        ra, rw = a - ah, w1 - wh
        conv = lambda u, v: F.conv1d(u.double(), v.double(), None, padding=(k - 1) // 2)
        y = conv(ah, wh)
        mask = torch.ones(k)
        if taps == "even":
            mask[1::2] = 0.0

        def quant(t, spec, kscale, dim):
            fmt, how = spec
            is_w = t.dim() == 3 and t.shape[0] == w1.shape[0] and t.shape[-1] == k
            if how == "static":
                return q_static(t, fmt, kscale)
            if how == "row" or (how == "hybrid" and is_w):
                return q_row(t, fmt, (1, 2) if is_w else (1,))
            return q_block(t, fmt, dim)
        if c_ra is not None and c_ra[1].startswith("shared"):
            rule = "ocp" if c_ra[1].startswith("shared-ocp") else "fit"
            pre = 4096.0 if c_ra[1] == "shared-4096" else 2048.0
            clip_w = 8.0 if c_ra[1].endswith(("clip8", "clipw8")) else (16.0 if c_ra[1].endswith("clip16") else 0.0)
            clip_a = 0.0 if c_ra[1].endswith("clipw8") else clip_w
            if c_ra[1].endswith("ablock16"):      # activations [B, C, T] -> blocks of 16 channels of one frame
                B_, C2_, T_ = ah.shape
                ab = lambda t: t.permute(0, 2, 1).reshape(B_, T_, C2_ // 16, 16)
                a4, r4a = q_shared(ab(ah), ab(ra), c_ra[0], (3,), rule, pre)
                ua = lambda t: t.reshape(B_, T_, C2_).permute(0, 2, 1)
                ah4, ra4 = ua(a4), ua(r4a)
            else:
                ah4, ra4 = q_shared(ah, ra, c_ra[0], (1,), rule, pre, clip_a)
            if "wblock16" in c_ra[1]:      # weights [N, C, k] -> blocks of 16 channels of one (n, tap)
                N_, C_, k_ = wh.shape
                wb = lambda t: t.permute(0, 2, 1).reshape(N_, k_, C_ // 16, 16)
                h4, r4 = q_shared(wb(wh), wb(rw), c_ra[0], (3,), rule, pre)
                un = lambda t: t.reshape(N_, k_, C_).permute(0, 2, 1)
                wh4, rw4 = un(h4), un(r4)
            else:
                wh4, rw4 = q_shared(wh, rw, c_ra[0], (1, 2), rule, pre, clip_w)
            y = y + conv(ra4, wh4 * mask) + conv(ah4, rw4 * mask)
        elif c_ra is not None and c_ra[1].startswith("sep"):
            rule = c_ra[1][4:]
            y = y + conv(q_row2(ra, c_ra[0], (1,), rule), q_row2(wh, c_ra[0], (1, 2), rule) * mask) + conv(q_row2(ah, c_ra[0], (1,), rule), q_row2(rw, c_ra[0], (1, 2), rule) * mask)
        else:
            if c_ra is not None:
                y = y + conv(quant(ra, c_ra, ka + 11, 1), quant(wh, c_ra, kw, 1) * mask)
            if c_rw is not None:
                y = y + conv(quant(ah, c_rw, ka, 1), quant(rw, c_rw, kw + 11, 1) * mask)
        exact = conv(a, w1)
        stats.append((float((y - exact).abs().max()), float((y - exact).pow(2).mean().sqrt()), float(exact.abs().max())))
        h = torch.relu(y.float() + b1.view(1, -1, 1))
        return F.conv1d(h, w2, b2).transpose(1, 2)
    return ffn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=4)
    ap.add_argument("--only", default="", help="substring of the scheme labels to run")
    ap.add_argument("--outliers", action="store_true", help="add a weight set with 0.1 %% of the FFN conv weights x 30")
    args = ap.parse_args()
    global SCHEMES
    if args.only:
        SCHEMES = [sc for sc in SCHEMES if args.only in sc[1]]
    torch.set_num_threads(8)
    torch.manual_seed(0)
    hp = default_hparams()
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
    sd0 = portable_state_dict(model.state_dict(), seed=0)
    cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, hp.audio.num_mels)
    b = make_batch("c2", B=args.utterances)
    orig = O._ffn
    sets = [("default synthetic weights", sd0), ("LayerNorm gamma in [0.1, 8], beta in +-2", hostile_weights(sd0, "ln_wide")),
            ("Student-t(3) weights, same rms", hostile_weights(sd0, "student_t"))]
    if args.outliers:
        g = torch.Generator().manual_seed(3)
        sdo = {k: v.clone() for k, v in sd0.items()}
        for k, v in sdo.items():
            if ".feed_forward.w_1.weight" in k:
                m = torch.rand(v.shape, generator=g) < 1e-3
                sdo[k] = torch.where(m, v * 30.0, v)
        sets.append(("0.1 % of the FFN w_1 weights x 30", sdo))
    out = {}
    for wname, sd in sets:
        run = lambda: O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])["after"]
        ref = run()
        out[wname] = (float(ref.abs().max()), [])
        for sch in SCHEMES:
            stats = []
            O._ffn = make_ffn(sch, stats)
            try:
                d = float((run() - ref).abs().max())
            finally:
                O._ffn = orig
            out[wname][1].append((d, max(s[0] for s in stats), max(s[1] for s in stats), max(s[2] for s in stats)))
            print("# %s | %.2f %s: mel +%.2e" % (wname, sch[0], sch[1], d), file=sys.stderr)
    print("FFN conv w_1 (encoder + decoder) in cheaper arithmetics: mel max-abs ADDED to the fp32 oracle; c2, %d utterances teacher-forced, %d frames" % (args.utterances, int(b["olens"].sum())))
    print("(MFMA-equivalents per product: fp16 main term 1.0; an 8-bit cross term 0.5; a cross term with both operands in fp6 / fp4 0.25)")
    names = [w for w, _ in sets]
    print("%-5s %-58s" % ("equiv", "scheme") + "".join(" | %-24s" % ("%s" % n[:24]) for n in names) + " | operator max-abs / rms (default)")
    print("%-5s %-58s" % ("", "") + "".join(" | max |mel| %-13.2f" % out[n][0] for n in names) + " |")
    for i, sch in enumerate(SCHEMES):
        row = "%-5.2f %-58s" % (sch[0], sch[1])
        for n in names:
            row += " | +%-23.2e" % out[n][1][i][0]
        row += " | %.2e / %.2e at max |y| %.1f" % (out[names[0]][1][i][1], out[names[0]][1][i][2], out[names[0]][1][i][3])
        print(row)
    bar = lambda n: 5e-4 * max(out[n][0] / out[names[0]][0], 1.0)
    print("criterion of the review: <= 1.5e-4 on the default weights and <= 5e-4 x (mel scale) under the hostile sets (%s)"
          % ", ".join("%.1e" % bar(n) for n in names[1:]))
    ok = [(sch[0], sch[1]) for i, sch in enumerate(SCHEMES) if out[names[0]][1][i][0] <= 1.5e-4 and all(out[n][1][i][0] <= bar(n) for n in names[1:])]
    print("qualifying points: " + "; ".join("%.2f %s" % q for q in ok))


if __name__ == "__main__":
    main()
