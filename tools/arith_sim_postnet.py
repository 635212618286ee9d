#!/usr/bin/env python3
"""CPU simulation of the mx arithmetic (fp16 main term + e4m3 cross terms, STATIC per-tensor scales: gemm_mx.h) on the Postnet's three middle convolutions
(512 -> 512, k = 5; reference core/modules.py:285-358), before anything is built.  Their inputs are tanh outputs: |x| <= 1 is an exact a-priori bound (scale 2^8),
the weights are BatchNorm-folded at load time (scale from their maximum).  The Postnet's output is ADDED to the mel, so its error reaches the result unattenuated.
Teacher-forced c2 batch through the CPU oracle with _postnet replaced; mel max-abs against the unmodified oracle.  Test infrastructure: imports oracle/.

  python tools/arith_sim_postnet.py"""
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


def e4m3(x):
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def make_postnet(mode):
    def postnet(sd, x, cfg):
        n = cfg["postnet_layers"]
        for l in range(n):
            w = sd["postnet.postnet.%d.0.weight" % l]
            b = None
            if cfg["use_batch_norm"]:      # fold the eval-mode BatchNorm into the conv, as the library does at load time
                p = "postnet.postnet.%d.1" % l
                g = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + 1e-5)
                w = w * g.view(-1, 1, 1)
                b = sd[p + ".bias"] - sd[p + ".running_mean"] * g
            k = w.shape[-1]
            conv = lambda u, v: F.conv1d(u.double(), v.double(), None, padding=(k - 1) // 2)
            if mode == "exact" or l == 0 or l == n - 1:
                y = conv(x, w)
            else:
                ka, kw = 8, int(math.floor(math.log2(448.0 / float(w.abs().max()))))
                xh, wh = x.half().float(), w.half().float()
                rx, rw = x - xh, w - wh
                y = conv(xh, wh)
                if mode == "mx":
                    y = y + conv(e4m3(rx * 2.0 ** (ka + 11)), e4m3(wh * 2.0 ** kw)) * 2.0 ** -(ka + kw + 11)
                    y = y + conv(e4m3(xh * 2.0 ** ka), e4m3(rw * 2.0 ** (kw + 11))) * 2.0 ** -(ka + kw + 11)
            x = y.float() + (b.view(1, -1, 1) if b is not None else 0.0)
            if l != n - 1:
                x = torch.tanh(x)
        return x
    return postnet


def main():
    torch.set_num_threads(8)
    torch.manual_seed(0)
    hp = default_hparams()
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
    sd0 = portable_state_dict(model.state_dict(), seed=0)
    cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, hp.audio.num_mels)
    b = make_batch("c2", B=8)
    orig = O._postnet
    print("Postnet layers 1-3 (512 -> 512, k = 5) in the mx arithmetic; mel max-abs ADDED to the fp32 oracle; c2 B=8 teacher-forced, %d frames" % int(b["olens"].sum()))
    for wname, sd in (("default synthetic weights", sd0), ("Student-t(3) weights, same rms", hostile_weights(sd0, "student_t"))):
        run = lambda: O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])["after"]
        O._postnet = make_postnet("exact")
        ref = run()
        fold = float((ref - (lambda: (setattr(O, "_postnet", orig), run())[1])()).abs().max())
        out = {}
        for mode in ("mx", "f16x1"):
            O._postnet = make_postnet(mode)
            try:
                out[mode] = float((run() - ref).abs().max())
            finally:
                O._postnet = orig
        print("  %-32s max |mel| %.2f: BatchNorm folding alone %.1e | mx (2.0 equivalents) +%.2e | fp16 once, no cross terms (1.0) +%.2e" % (wname, float(ref.abs().max()), fold, out["mx"], out["f16x1"]))


if __name__ == "__main__":
    main()
