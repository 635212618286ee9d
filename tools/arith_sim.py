#!/usr/bin/env python3
"""CPU simulation of candidate reduced-MFMA arithmetic modes, run BEFORE any kernel is written.

A split scheme with fewer than three MFMAs per product keeps one operand at a single rounding (fp16: 11 significand bits,
bf16: 8).  Which operand is rounded does not change the error model, so the experiment rounds the WEIGHTS of the chosen GEMMs
(activations stay fp32 = "hi + lo" exact to ~22 bits) and pushes the teacher-forced c2 batch through the CPU oracle; the
mel max-abs difference against the unrounded oracle is what that mode would add on top of the bf16x3 floor (2e-5).
Test infrastructure: imports oracle/.

  python tools/arith_sim.py            # table on stdout
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS   # noqa: E402
from fastspeech2_amd.synthetic import portable_state_dict, make_batch                     # noqa: E402
from oracle import fs2_oracle as O                                                        # noqa: E402


def groups(sd):
    g = {"dec.ffn1": [], "enc.ffn1": [], "dec.ffn2": [], "enc.ffn2": [], "dec.qkv": [], "enc.qkv": [], "dec.out": [], "enc.out": [],
         "predictors": [], "postnet": [], "dec.in": [], "feat_out": []}
    for k in sd:
        if not k.endswith(".weight"):
            continue
        st = "dec" if k.startswith("decoder.") else ("enc" if k.startswith("encoder.") else None)
        if st and ".feed_forward.w_1." in k: g[st + ".ffn1"].append(k)
        elif st and ".feed_forward.w_2." in k: g[st + ".ffn2"].append(k)
        elif st and (".linear_q." in k or ".linear_k." in k or ".linear_v." in k): g[st + ".qkv"].append(k)
        elif st and ".linear_out." in k: g[st + ".out"].append(k)
        elif "predictor" in k and ".conv." in k and k.endswith(".0.weight"): g["predictors"].append(k)
        elif k.startswith("postnet.") and k.endswith(".0.weight"): g["postnet"].append(k)
        elif k == "decoder.embed.0.weight": g["dec.in"].append(k)
        elif k == "feat_out.weight": g["feat_out"].append(k)
    return g


def main():
    torch.set_num_threads(8)
    hp = default_hparams()
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
    sd = portable_state_dict(model.state_dict(), seed=0)
    cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, hp.audio.num_mels)
    b = make_batch("c2", B=8)
    run = lambda s: O.per_utterance_forward(s, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])["after"]
    ref = run(sd)
    g = groups(sd)
    every = sorted(sum(g.values(), []))
    rnd = {"fp16": lambda w: w.half().float(), "bf16": lambda w: w.bfloat16().float()}
    rows = [("all GEMMs", every), ("dec.ffn1 + enc.ffn1", g["dec.ffn1"] + g["enc.ffn1"]), ("dec.ffn1", g["dec.ffn1"]),
            ("enc.ffn1", g["enc.ffn1"]), ("dec.ffn1 + dec.ffn2", g["dec.ffn1"] + g["dec.ffn2"]),
            ("dec.qkv + dec.out", g["dec.qkv"] + g["dec.out"]), ("postnet", g["postnet"]), ("predictors", g["predictors"])]
    print("%-24s %12s %12s   (mel max-abs vs the unrounded oracle, c2 B=8 teacher-forced, %d frames)" % ("single-rounded operand in", "fp16", "bf16", int(b["olens"].sum())))
    for name, keys in rows:
        out = []
        for kind in ("fp16", "bf16"):
            s2 = dict(sd)
            for k in keys:
                s2[k] = rnd[kind](sd[k])
            out.append(float((run(s2) - ref).abs().max()))
        print("%-24s %12.2e %12.2e" % (name, out[0], out[1]))


if __name__ == "__main__":
    main()
