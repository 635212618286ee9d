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
with torch.no_grad():
    refs = [model.inference_batch(x, il, d_override=d) for x, il, d in ins]
    sel = torch.argsort(ins[0][1])[:8]
    m = int(ins[0][1][sel].max())
    sub_ref = model.inference_batch(ins[0][0][sel.cuda()][:, :m], ins[0][1][sel], d_override=ins[0][2][sel.cuda()][:, :m])
    print("refs", [(int(r[1].sum()), int(r[1].max())) for r in refs], "sub", int(sub_ref[1].sum()), int(sub_ref[1].max()), "m", m, "ratio", model._frames_per_token)
    for ov in (False, True):
        for use_in_stream in (False, True):
            model.overlap_encoder = ov
            torch.cuda.synchronize()
            log = []
            for it in range(3):
                if use_in_stream:
                    with torch.cuda.stream(model.input_stream(ins[0][0].device)):
                        xs_s, ds_s = ins[0][0][sel.cuda()][:, :m], ins[0][2][sel.cuda()][:, :m]
                else:
                    xs_s, ds_s = ins[0][0][sel.cuda()][:, :m], ins[0][2][sel.cuda()][:, :m]
                    torch.cuda.synchronize()
                r = model.inference_batch(xs_s, ins[0][1][sel], d_override=ds_s, sync=False, capacity=(int(sub_ref[1].sum()) + 512, int(sub_ref[1].max()) + 32))
                log.append(("sub", r))
                for k in (1, 0):
                    cap_k = (int(refs[k][1].sum()) + 64 * len(refs[k][1]), int(refs[k][1].max()) + 32)
                    log.append(("b%d" % (k + 1), model.inference_batch(*ins[k][:2], d_override=ins[k][2], sync=False, capacity=cap_k)))
            torch.cuda.synchronize()
            print("overlap", ov, "inputs on the input stream", use_in_stream, [(n, r.status.cpu().tolist()[:5]) for n, r in log[:6]])
            model.async_ok()
