#!/usr/bin/env python3
"""Diagnostic 2 (round 5): where does utterance 130 of a 512-utterance c5 call go wrong -- encoder or decoder, and under which switches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS, _lib   # noqa: E402
from fastspeech2_amd.synthetic import portable_state_dict, make_batch                         # noqa: E402

hp = default_hparams()
model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
sd = portable_state_dict(model.state_dict(), seed=0)
model.load_state_dict(sd)
model = model.to("cuda:0")
b = make_batch("c5")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
xs, il, ds = b["xs"][:N].cuda(), b["ilens"][:N], b["ds"][:N].cuda()
model.precision = "mix_mx"


def run():
    with torch.no_grad():
        r = model._run(xs, il, is_inference=True, d_override=ds, want=("after", "encoder_out", "decoder_out", "before"))
    return {k: r[k].cpu() for k in ("after", "encoder_out", "decoder_out", "before")}, r["olens"]


def cmp(tag, a, c, ol):
    for k in ("encoder_out", "decoder_out", "before", "after"):
        x, y = a[k], c[k]
        d = (x - y).abs().flatten(1).amax(1)
        bad = torch.nonzero(d > 1e-4).flatten().tolist()
        print("%-28s %-12s worst %.3e  utterances beyond 1e-4: %s" % (tag, k, float(d.max()), bad[:10]), flush=True)


_lib.set_option("FS2_ATTN_W32", 0)
good, ol = run()
_lib.set_option("FS2_ATTN_W32", -1)
dflt, _ = run()
cmp("default vs W32=0", dflt, good, ol)
i = 130
T, L = int(il[i]), int(ol[i])
print("utterance %d: %d tokens, %d frames; neighbours' tokens %s frames %s" % (i, T, L, il[i - 2:i + 3].tolist(), ol[i - 2:i + 3].tolist()))
dd = (dflt["decoder_out"][i, :L] - good["decoder_out"][i, :L]).abs().amax(-1)
print("decoder_out rows that differ:", torch.nonzero(dd > 1e-4).flatten().tolist()[:40])
de = (dflt["encoder_out"][i, :T] - good["encoder_out"][i, :T]).abs().amax(-1)
print("encoder_out rows that differ:", torch.nonzero(de > 1e-5).flatten().tolist()[:40])
# which attention: force w32 off via the regime? -- run the encoder-only switches
for opt, val in (("FS2_MT8", 2), ("FS2_MT8", 3), ("FS2_MT4", 4), ("FS2_MT4", 5), ("FS2_QKV_SPLIT", 0), ("FS2_QKV_SPLIT", 1), ("FS2_BM", 128)):
    _lib.set_option(opt, val)
    r, _ = run()
    cmp("%s=%d vs W32=0" % (opt, val), r, good, ol)
    _lib.set_option(opt, -1)
# run-to-run
r2, _ = run()
print("default, second run identical to the first:", all(torch.equal(r2[k], dflt[k]) for k in r2))
