#!/usr/bin/env python3
"""One schedule per process (streams are mapped to a handful of hardware queues, so schedules tried one after another in one process disturb each
other): stream_schedule_ab.py <workload> <overlap_encoder 0|1> <streams> <steps>  ->  one line."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
from fastspeech2_amd.synthetic import portable_state_dict, ljspeech_durations, make_batch
wl, ov, nstreams, K = sys.argv[1], bool(int(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4])
hp = default_hparams()
model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
model.load_state_dict(ljspeech_durations(portable_state_dict(model.state_dict(), seed=0)))
model = model.to("cuda:0")
model.precision = "mix_mx"
b = make_batch("c5", B=128) if wl == "c5" else make_batch(wl)
xs, il = b["xs"].cuda(), b["ilens"]
with torch.no_grad():
    mel, ol = model.inference_batch(xs, il)
    frames = int(ol.sum())
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
            if len(outs) > 4:
                outs.pop(0)
        return outs
    run(8); torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        outs = run(K)
        torch.cuda.synchronize()
        res.append(time.perf_counter() - t0)
    ok = model.async_ok()
    same = all(torch.equal(o[0][:, :mel.shape[1]], mel) for o in outs)
print("%s overlap_encoder=%d streams=%d: %s ms/step -> best %.2f M frames/s, ok=%s identical=%s"
      % (wl, ov, nstreams, " ".join("%.3f" % (1e3 * t / K) for t in res), frames * K / min(res) / 1e6, ok, same), flush=True)
