#!/usr/bin/env python3
"""Experiment (round 5): consecutive steps issued on alternating streams, with and without overlap_encoder -- does letting the dispatcher fill the
tail rounds of one step's kernels with the next step's workgroups pay?  c3 / c2 / c4 / one c5 shard, mix_mx, one line per schedule; every schedule's
mels are compared with the synchronous call's."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
from fastspeech2_amd.synthetic import portable_state_dict, ljspeech_durations, make_batch
hp = default_hparams()
model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
model.load_state_dict(ljspeech_durations(portable_state_dict(model.state_dict(), seed=0)))
model = model.to("cuda:0")
model.precision = "mix_mx"
for wl, K in (("c3", 40), ("c2", 60), ("c5", 40), ("c4", 8), ("c1", 200)):
    b = make_batch("c5", B=128) if wl == "c5" else make_batch(wl)
    xs, il = b["xs"].cuda(), b["ilens"]
    with torch.no_grad():
        mel, ol = model.inference_batch(xs, il)
        frames = int(ol.sum())
        for ov, nstreams in ((False, 1), (True, 1), (True, 2), (False, 2), (True, 3)):
            model.overlap_encoder = ov
            streams = [torch.cuda.Stream() for _ in range(nstreams)]
            def run(n):
                outs = []
                for i in range(n):
                    if nstreams == 1:
                        outs.append(model.inference_batch(xs, il, sync=False))
                    else:
                        with torch.cuda.stream(streams[i % nstreams]):
                            outs.append(model.inference_batch(xs, il, sync=False))
                return outs
            run(6); torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = run(K)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ok = model.async_ok()
            same = all(torch.equal(o[0][:, :mel.shape[1]], mel) for o in outs[-3:])
            print("%s overlap_encoder=%s streams=%d: %.3f ms/step, %.2f M frames/s, ok=%s identical=%s" % (wl, ov, nstreams, 1e3 * dt / K, frames * K / dt / 1e6, ok, same), flush=True)
            del outs
