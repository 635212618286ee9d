"""End-to-end latency of one-utterance calls (host work + GPU + synchronisation) and a cProfile of the host side.
   python tools/latency_c1.py      (on an MI355X)"""
import time, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
from fastspeech2_amd.synthetic import portable_state_dict, ljspeech_durations, make_batch
hp = default_hparams()
model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
model.load_state_dict(ljspeech_durations(portable_state_dict(model.state_dict(), seed=0)))
model = model.cuda(); model.precision = "mix_mx4"
b = make_batch("c1")
x = b["xs"][0, : int(b["ilens"][0])].cuda()
with torch.no_grad():
    for _ in range(10): out = model.inference(x)
    torch.cuda.synchronize()
    for name, fn in [("inference(x) + sync", lambda: model.inference(x)),
                     ("inference_batch(sync=True)", lambda: model.inference_batch(x[None], b["ilens"][:1], sync=True)),
                     ("inference_batch(sync=False)", lambda: model.inference_batch(x[None], b["ilens"][:1], sync=False))]:
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); o = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts.sort()
        print("%-30s median %.3f ms  p10 %.3f  p90 %.3f" % (name, ts[100] * 1e3, ts[20] * 1e3, ts[180] * 1e3))
    # host-side cost only: enqueue without waiting
    t0 = time.perf_counter()
    for _ in range(200): o = model.inference_batch(x[None], b["ilens"][:1], sync=False)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("enqueue only: %.3f ms per call (host), drained after %.3f ms more" % ((t1 - t0) / 200 * 1e3, (t2 - t1) * 1e3))
import cProfile, pstats, io
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for _ in range(200): o = model.inference_batch(x[None], b["ilens"][:1], sync=False)
    pr.disable()
torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14); print(st.getvalue()[:3500])
