#!/usr/bin/env python3
"""CPU simulation of the "mx" arithmetic (fp16 x fp16 + block-scaled e4m3 cross terms, gemm_mx.h) on the SECOND FFN GEMM (w_2, K = dunits = 1024)
of the decoder -- VERDICT r04 item 1 "while there" -- run before any kernel is written.  The hidden layer h = relu(w_1 * x + b_1) gets the STATIC scale
2^kh from the a-priori bound |h_n| <= sum |w_1[n]| xmax + |b_1[n]| (xmax = the LayerNorm bound sqrt(D) max|gamma| + max|beta| of the layer's input),
the weights 2^kw from max |w_2|; operands are rounded exactly as store_planes4_mx / repack_weight_mx round them (fp16 main part, e4m3 of the fp16
copy and of the residual x 2^11), products accumulate in fp32 (the simulation: fp64).  Teacher-forced c2 batch through the CPU oracle with _ffn
replaced; mel max-abs against the unmodified oracle.  Test infrastructure: imports oracle/.

  python tools/arith_sim_ffn2.py
"""
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


def e4m3(x):
    """round to nearest e4m3 (saturating at 448), as v_cvt_pk_fp8_f32 does"""
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def split_mx(x, k):
    xh = x.half().float()
    return xh, e4m3(xh * 2.0 ** k), e4m3((x - xh) * 2.0 ** (k + 11))


def exp_for(bound):
    return int(math.floor(math.log2(448.0 / max(bound, 1e-30))))


def main():
    torch.set_num_threads(8)
    hp = default_hparams()
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
    sd = portable_state_dict(model.state_dict(), seed=0)
    cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, hp.audio.num_mels)
    b = make_batch("c2", B=8)
    run = lambda: O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])["after"]
    ref = run()
    orig = O._ffn
    stats = {}

    def make(mode, which):
        def ffn(sd_, p, x, cfg_):
            if not any(p.startswith(w) for w in which):
                return orig(sd_, p, x, cfg_)
            w1, b1 = sd_[p + ".w_1.weight"], sd_[p + ".w_1.bias"]
            w2, b2 = sd_[p + ".w_2.weight"], sd_[p + ".w_2.bias"]
            k = w1.shape[-1]
            h = torch.relu(F.conv1d(x.transpose(1, 2), w1, b1, padding=(k - 1) // 2))            # [B, H, T]
            # static bound of the hidden layer: LayerNorm bound of x (the norm1 of this layer) times the l1 norm of the w_1 rows
            lnp = p.replace(".feed_forward", ".norm1")
            xmax = math.sqrt(x.shape[-1]) * float(sd_[lnp + ".weight"].abs().max()) + float(sd_[lnp + ".bias"].abs().max())
            hb = float((w1.abs().sum(dim=(1, 2)) * xmax + b1.abs()).max())
            kh, kw = exp_for(hb), exp_for(float(w2.abs().max()))
            W = w2[:, :, 0].double()
            Hd = h.double()
            if mode == "mx":
                hh, h8, rh8 = split_mx(h, kh)
                wh, w8, rw8 = split_mx(w2[:, :, 0], kw)
                y = torch.einsum("nc,bct->bnt", wh.double(), hh.double())
                y = y + (torch.einsum("nc,bct->bnt", w8.double(), rh8.double()) + torch.einsum("nc,bct->bnt", rw8.double(), h8.double())) * 2.0 ** -(kh + kw + 11)
            elif mode == "f16x1":
                y = torch.einsum("nc,bct->bnt", W.half().double() if False else w2[:, :, 0].half().double(), h.half().double())
            else:       # bf16x3: hi.hi + hi.lo + lo.hi
                sp = lambda v: (v.bfloat16().float(), (v - v.bfloat16().float()).bfloat16().float())
                hh, hl = sp(h)
                wh, wl = sp(w2[:, :, 0])
                y = (torch.einsum("nc,bct->bnt", wh.double(), hh.double()) + torch.einsum("nc,bct->bnt", wh.double(), hl.double())
                     + torch.einsum("nc,bct->bnt", wl.double(), hh.double()))
            exact = torch.einsum("nc,bct->bnt", W, Hd)
            st = stats.setdefault((mode, p), [])
            st.append((float((y - exact).abs().max()), float(exact.abs().max()), float(h.max()), hb, kh, kw))
            return (y.float() + b2.view(1, -1, 1)).transpose(1, 2)
        return ffn

    print("second FFN GEMM (w_2) in a reduced arithmetic; mel max-abs vs the unmodified oracle, c2 B=8 teacher-forced, %d frames" % int(b["olens"].sum()))
    for mode in ("bf16x3", "mx", "f16x1"):
        for which, name in ((("decoder.",), "decoder"), (("decoder.", "encoder."), "decoder + encoder")):
            stats.clear()
            O._ffn = make(mode, which)
            try:
                d = float((run() - ref).abs().max())
            finally:
                O._ffn = orig
            op = max(v[0] for vs in stats.values() for v in vs)
            mag = max(v[1] for vs in stats.values() for v in vs)
            hm = max(v[2] for vs in stats.values() for v in vs)
            hb = max(v[3] for vs in stats.values() for v in vs)
            kh = min(v[4] for vs in stats.values() for v in vs)
            print("  %-7s %-18s mel %.2e | operator: max |err| %.2e at max |y| %.2f; hidden layer max %.2f against the static bound %.0f (2^%d)"
                  % (mode, name, d, op, mag, hm, hb, kh))


if __name__ == "__main__":
    main()
