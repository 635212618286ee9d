"""Find the duration-predictor bias that makes the FREE-RUNNING synthetic model emit LJSpeech-like durations (mean 7.87 frames per
phoneme) on config c3 with the portable seed-0 weights.  ln(1 + 7.87) alone gives 6.33 frames per phoneme: the
predictor's output has a spread around its bias and clamp(round(exp(x) - 1), 0) is not linear in it.  Uses the CPU oracle (test
infrastructure); the result is the constant DUR_BIAS_C3 in fastspeech2_amd/synthetic.py."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
from fastspeech2_amd.synthetic import portable_state_dict, make_batch
from oracle import fs2_oracle as O

hp = default_hparams()
model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp)
sd = portable_state_dict(model.state_dict(), 0)
cfg = O.config_from_hp(hp, N_PHONEME_SYMBOLS, hp.audio.num_mels)
sd0 = dict(sd)
sd0["duration_predictor.linear.bias"] = torch.zeros(1)
for wl in sys.argv[1:] or ["c3"]:
    b = make_batch(wl)
    xs, il = b["xs"], b["ilens"]
    logs = []
    for i in range(xs.shape[0]):
        T = int(il[i])
        h = torch.nn.functional.embedding(xs[i:i + 1, :T], sd0["encoder.embed.0.weight"])
        h = O._add_pos(sd0, "encoder.embed.1", h, cfg)
        hs = O._fft_stack(sd0, "encoder", h, None, cfg["elayers"], cfg["aheads"], cfg)
        logs.append(O._predictor(sd0, "duration_predictor", hs, cfg["dur_layers"])[0])
    x = torch.cat(logs)
    f = lambda bias: float(O.duration_from_log(x + bias).float().mean())
    lo, hi = 1.0, 4.0
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        lo, hi = (mid, hi) if f(mid) < 7.87 else (lo, mid)
    print(wl, "tokens", x.numel(), "spread of the bias-free output: mean %.4f std %.4f" % (float(x.mean()), float(x.std())),
          "| ln(1+7.87) = %.4f gives %.3f frames/phoneme | bias %.4f gives %.3f" % (np.log(8.87), f(float(np.log(8.87))), hi, f(hi)))
