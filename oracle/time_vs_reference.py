"""Wall time of the CPU oracle vs the REAL reference on identical inputs (build container only: imports /root/reference).
BASELINE.md section 3 requires this figure: the oracle is the `cpu_baseline` ("kind": "port") of bench.py on the GPU box,
where the reference cannot run, so it must cost what the reference costs.  TEST INFRASTRUCTURE ONLY.

  python oracle/time_vs_reference.py         # prints mel-frames/s of both, per-utterance loop, free-running forced durations
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.gen_golden import import_reference          # noqa: E402
from oracle import fs2_oracle as O                      # noqa: E402
from fastspeech2_amd.synthetic import portable_state_dict, make_batch   # noqa: E402


def main():
    torch.set_num_threads(int(os.environ.get("THREADS", "8")))
    hp, idim, Ref = import_reference()
    odim = hp.audio.num_mels
    ref = Ref(idim, odim, hp).eval()
    sd = portable_state_dict(ref.state_dict(), seed=0)
    ref.load_state_dict(sd)
    cfg = O.config_from_hp(hp, idim, odim)
    b = make_batch("c3", B=8)
    xs, il, ds = b["xs"], b["ilens"], b["ds"]
    frames = int(b["olens"].sum())

    def run_ref():
        worst = 0.0
        for i in range(xs.shape[0]):
            T = int(il[i])
            d = ds[i:i + 1, :T]
            ref.duration_predictor.inference = lambda hs, masks, d=d: d          # forced durations
            with torch.no_grad():
                out = ref._forward(xs[i:i + 1, :T], il[i:i + 1], is_inference=True)
            o = O.padded_forward(sd, cfg, xs[i:i + 1, :T], il[i:i + 1], is_inference=True, d_override=d)
            worst = max(worst, float((out[1] - o["after"]).abs().max()))
        return worst

    def t_ref():
        for i in range(xs.shape[0]):
            T = int(il[i])
            d = ds[i:i + 1, :T]
            ref.duration_predictor.inference = lambda hs, masks, d=d: d
            with torch.no_grad():
                ref._forward(xs[i:i + 1, :T], il[i:i + 1], is_inference=True)

    def t_orc():
        for i in range(xs.shape[0]):
            T = int(il[i])
            O.padded_forward(sd, cfg, xs[i:i + 1, :T], il[i:i + 1], is_inference=True, d_override=ds[i:i + 1, :T])

    print("max-abs oracle vs reference on these utterances: %.1e" % run_ref())
    for name, fn in (("reference", t_ref), ("oracle", t_orc), ("reference", t_ref), ("oracle", t_orc)):
        fn()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            best = min(best, time.perf_counter() - t0)
        print("%-9s %d utterances, %d frames, %d threads: %.3f s = %.0f mel-frames/s" % (name, xs.shape[0], frames, torch.get_num_threads(), best, frames / best))


if __name__ == "__main__":
    main()
