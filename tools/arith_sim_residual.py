#!/usr/bin/env python3
"""CPU simulation of a residual stream that exists ONLY as the activation planes the GEMMs consume, run before the kernels
change.  Today every LayerNorm-fused launch of the bf16 modes reads its residual as fp32 rows and writes fp32 rows beside the planes of the same
values; planes-only means the residual that enters `x = residual + sublayer(x)` (reference core/encoder.py:60-69) is the value the planes hold:
  * "bf16x2": hi = bf16(x), lo = bf16(x - hi)                                     (split-bf16 planes: FFN2 + LN2's output, the decoder input layer's)
  * "mx":     fp16(x) + e4m3((x - fp16(x)) 2^(ka+11)) 2^-(ka+11)                  (mx planes: out-proj + LN1's output in mix_mx mode, with the static scale
              2^ka of the LayerNorm's a-priori bound sqrt(D) max|gamma| + max|beta|, exactly as store_planes4_mx rounds)
  * "bf16x3": a third bf16 plane (the fallback the review names: 6 B/element)
Everything else stays the exact fp32 oracle, so the number printed is the error this one change ADDS (today's total, mix_mx vs the oracle: 3.0e-5).
Teacher-forced c2 batch; mel max-abs against the unmodified oracle.  Test infrastructure: imports oracle/.

  python tools/arith_sim_residual.py
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


def bf16_planes(x, n):
    acc = torch.zeros_like(x)
    for _ in range(n):
        acc = acc + (x - acc).bfloat16().float()
    return acc


def mx_planes(x, ka):
    xh = x.clamp(-65504.0, 65504.0).half().float()
    r8 = ((x - xh) * 2.0 ** (ka + 11)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float()
    return xh + r8 * 2.0 ** -(ka + 11)


def exp_for(bound):
    return int(math.floor(math.log2(448.0 / max(bound, 1e-30))))


def make_stack(fmt_ln1, fmt_ln2, which, hostile=None):
    """_fft_stack (post-LN form) with the residual inputs rounded: the residual of the attention sub-layer is the previous LN2 output (or the stack's
    input), the residual of the FFN sub-layer is the LN1 output."""
    def stack(sd, prefix, x, mask, nlayers, heads, cfg, pre_ln=False, concat=False):
        if pre_ln or concat or not prefix.startswith(which):
            return ORIG(sd, prefix, x, mask, nlayers, heads, cfg, pre_ln, concat)
        D = x.shape[-1]
        ln = lambda t, name: F.layer_norm(t, (D,), sd[name + ".weight"], sd[name + ".bias"], 1e-5)
        rnd = lambda t, fmt, ka: t if fmt == "fp32" else (mx_planes(t, ka) if fmt == "mx" else bf16_planes(t, int(fmt[-1])))
        for i in range(nlayers):
            p = "%s.encoders_.%d" % (prefix, i)
            x = ln(rnd(x, fmt_ln2, 0) + O._mha(sd, p + ".self_attn", x, mask, heads), p + ".norm1")
            ka = exp_for(math.sqrt(D) * float(sd[p + ".norm1.weight"].abs().max()) + float(sd[p + ".norm1.bias"].abs().max()))
            x = ln(rnd(x, fmt_ln1, ka) + O._ffn(sd, p + ".feed_forward", x, cfg), p + ".norm2")
        return x
    return stack


def hostile_weights(sd, kind, seed=1):
    g = torch.Generator().manual_seed(seed)
    sd = {k: v.clone() for k, v in sd.items()}
    for k, v in sd.items():
        if kind == "ln_wide" and (".norm1." in k or ".norm2." in k or k.startswith("decoder.embed.1.")):
            if k.endswith(".weight"):
                sd[k] = torch.exp(torch.empty_like(v).uniform_(math.log(0.1), math.log(8.0), generator=g))
            else:
                sd[k] = torch.empty_like(v).uniform_(-2.0, 2.0, generator=g)
        if kind == "student_t" and v.dim() >= 2 and v.is_floating_point() and "embed" not in k and "pe" not in k.split(".")[-1]:
            t = torch.distributions.StudentT(3.0).sample(v.shape)
            sd[k] = t * (v.pow(2).mean().sqrt() / t.pow(2).mean().sqrt())
    return sd


def main():
    global ORIG
    torch.set_num_threads(8)
    torch.manual_seed(0)
    hp = default_hparams()
    model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
    sd0 = portable_state_dict(model.state_dict(), seed=0)
    cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, hp.audio.num_mels)
    b = make_batch("c2", B=8)
    ORIG = O._fft_stack
    print("residual stream held only as activation planes; mel max-abs ADDED by that, vs the unmodified fp32 oracle; c2 B=8 teacher-forced, %d frames" % int(b["olens"].sum()))
    for wname, sd in (("default synthetic weights", sd0), ("LayerNorm gamma in [0.1, 8], beta in +-2", hostile_weights(sd0, "ln_wide")),
                      ("Student-t(3) weights, same rms", hostile_weights(sd0, "student_t"))):
        run = lambda: O.per_utterance_forward(sd, cfg, b["xs"], b["ilens"], b["ds"], b["es"], b["ps"])["after"]
        ref = run()
        print("  %s (max |mel| %.2f):" % (wname, float(ref.abs().max())))
        for name, f1, f2 in (("LN2 out bf16x2, LN1 out bf16x2 (bf16x3 mode)", "bf16x2", "bf16x2"), ("LN2 out bf16x2, LN1 out mx (mix_mx mode)", "mx", "bf16x2"),
                             ("LN2 out bf16x2, LN1 out fp32 kept", "fp32", "bf16x2"), ("both bf16x3 (6 B/element)", "bf16x3", "bf16x3")):
            for which in ("decoder", ("decoder", "encoder")):
                O._fft_stack = make_stack(f1, f2, which)
                try:
                    d = float((run() - ref).abs().max())
                finally:
                    O._fft_stack = ORIG
                print("    %-46s %-20s mel +%.2e" % (name, "decoder" if which == "decoder" else "decoder + encoder", d))


if __name__ == "__main__":
    main()
