import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
from fastspeech2_amd.synthetic import portable_state_dict, make_batch
hp = default_hparams()
model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
model.load_state_dict(portable_state_dict(model.state_dict(), seed=0))
model = model.to("cuda:0")
model.precision = "mix_mx"
b = make_batch("c2", B=12)
xs, il, ds = b["xs"].cuda(), b["ilens"], b["ds"].cuda()
with torch.no_grad():
    ref, ol = model.inference_batch(xs, il, d_override=ds)
    print("ref", int(ol.sum()), int(ol.max()), model._frames_per_token)
    for ov in (False, True):
        model.overlap_encoder = ov
        for cap in (None, (int(ol.sum()) + 64 * 12, int(ol.max()) + 32), (int(ol.sum()) + 64 * 12, 1024), (12000, 995), (12000, 1024)):
            r = model.inference_batch(xs, il, d_override=ds, sync=False, capacity=cap)
            st = r.status.cpu().tolist()
            print("overlap", ov, "capacity", cap, "status", st[:5], "equal", bool(torch.equal(r[0][:, :ref.shape[1]], ref)))
    model.async_ok()
