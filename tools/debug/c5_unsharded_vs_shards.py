#!/usr/bin/env python3
"""Diagnostic (round 5): the unsharded 1,024-utterance c5 call against the eight LPT shards run alone -- which utterances differ, by how much, under which
kernel switches, and which side agrees with the CPU oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS, _lib   # noqa: E402
from fastspeech2_amd.synthetic import portable_state_dict, make_batch                         # noqa: E402
from fastspeech2_amd.parallel import shard_indices                                            # noqa: E402
from oracle import fs2_oracle as O                                                            # noqa: E402

hp = default_hparams()
model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
sd = portable_state_dict(model.state_dict(), seed=0)
model.load_state_dict(sd)
model = model.to("cuda:0")
cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, hp.audio.num_mels)
b = make_batch("c5")
xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()
parts = shard_indices(il.tolist(), 8)


def run(nutt=1024):
    with torch.no_grad():
        pk, ol = model.inference_batch(xs[:nutt], il[:nutt], d_override=ds[:nutt], packed=True)
    st = (torch.cumsum(ol, 0) - ol).tolist()
    h = pk.cpu()
    return [h[st[i]:st[i] + int(ol[i])] for i in range(nutt)]


def shards():
    out = [None] * 1024
    with torch.no_grad():
        for p_ in parts:
            sel = torch.as_tensor(p_)
            Tm = int(il[sel].max())
            pk, ol = model.inference_batch(xs[sel.cuda()][:, :Tm], il[sel], d_override=ds[sel.cuda()][:, :Tm], packed=True)
            h, off = pk.cpu(), 0
            for j, g in enumerate(p_):
                out[g] = h[off:off + int(ol[j])]
                off += int(ol[j])
    return out


def report(tag, a, c):
    d = [float((x - y).abs().max()) for x, y in zip(a, c)]
    bad = [i for i, v in enumerate(d) if v > 1e-4]
    print("%-44s worst %.3e; %d utterance(s) beyond 1e-4: %s" % (tag, max(d), len(bad), bad[:12]), flush=True)
    return bad


for prec in ("mix_mx", "bf16x3", "fp32"):
    model.precision = prec
    ref = shards()
    bad = report("[%s] unsharded 1024 vs shards alone" % prec, run(), ref)
    if prec == "mix_mx":
        for opt in ("FS2_ROW4", "FS2_ATTN_W32", "FS2_FFN2_MX", "FS2_QKV8", "FS2_ROW8"):
            _lib.set_option(opt, 0)
            report("[mix_mx] %s=0" % opt, run(), ref)
            _lib.set_option(opt, -1)
        for n in (512, 768, 896):
            report("[mix_mx] first %d utterances in one call" % n, run(n), ref[:n])
        if bad:
            i = bad[0]
            T = int(il[i])
            o = O.padded_forward(sd, cfg, b["xs"][i:i + 1, :T], il[i:i + 1], is_inference=True, d_override=b["ds"][i:i + 1, :T])["after"][0]
            un = run()[i]
            print("   utterance %d (%d frames): unsharded vs oracle %.3e, shard-alone vs oracle %.3e" % (i, o.shape[0], float((un - o).abs().max()), float((ref[i] - o).abs().max())))
            dd = (un - ref[i]).abs().amax(-1)
            nz = torch.nonzero(dd > 1e-4).flatten()
            print("   frames that differ: %d of %d, first %s last %s" % (len(nz), len(dd), nz[:5].tolist(), nz[-5:].tolist()))
