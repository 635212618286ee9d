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
b1, b2 = make_batch("c3", B=24), make_batch("c2", B=12)
ins = [(b["xs"].cuda(), b["ilens"], b["ds"].cuda()) for b in (b1, b2)]
def st(r): return r.status.cpu().tolist()[:5]
with torch.no_grad():
    refs = [model.inference_batch(x, il, d_override=d) for x, il, d in ins]
    free = [model.inference_batch(x, il) for x, il, d in ins]
    print("ratio after refs+free", model._frames_per_token, [int(f[1].sum()) for f in free])
    model.async_ok(); torch.cuda.synchronize()
    model.overlap_encoder = True
    outs = [model.inference_batch(*ins[i & 1][:2], d_override=ins[i & 1][2], sync=False) for i in range(8)]
    outs_free = [model.inference_batch(*ins[i & 1][:2], sync=False, packed=bool(i & 2)) for i in range(8)]
    print("first 16:", [st(o)[2] for o in outs + outs_free])
    sel = torch.argsort(ins[0][1])[:8]
    m = int(ins[0][1][sel].max())
    model.overlap_encoder = False
    sub_ref = model.inference_batch(ins[0][0][sel.cuda()][:, :m], ins[0][1][sel], d_override=ins[0][2][sel.cuda()][:, :m])
    torch.cuda.synchronize()
    model.overlap_encoder = True
    log = []
    for _ in range(2):
        with torch.cuda.stream(model.input_stream(ins[0][0].device)):
            xs_s, ds_s = ins[0][0][sel.cuda()][:, :m], ins[0][2][sel.cuda()][:, :m]
        log.append(("sub", model.inference_batch(xs_s, ins[0][1][sel], d_override=ds_s, sync=False, capacity=(int(sub_ref[1].sum()) + 512, int(sub_ref[1].max()) + 32))))
        for k in (1, 0):
            cap_k = (int(refs[k][1].sum()) + 64 * len(refs[k][1]), int(refs[k][1].max()) + 32)
            log.append(("b%d cap %s" % (k + 1, cap_k), model.inference_batch(*ins[k][:2], d_override=ins[k][2], sync=False, capacity=cap_k)))
    torch.cuda.synchronize()
    print([(n, st(r)) for n, r in log])
    print("pe rows", model.decoder.embed[-1].pe.shape if hasattr(model.decoder.embed[-1], 'pe') else None)
