"""c4 batch in several arithmetic modes / attention kernels against the fp32 device result (itself within 4e-6 of the oracle):
per-utterance mel error and bucket-decision flips.  python tools/debug/c4_modes.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fastspeech2_amd import _lib
from fastspeech2_amd import FeedForwardTransformer, default_hparams, N_PHONEME_SYMBOLS
from fastspeech2_amd.synthetic import make_batch, portable_state_dict

hp = default_hparams()
model = FeedForwardTransformer(N_PHONEME_SYMBOLS, hp.audio.num_mels, hp).eval()
model.load_state_dict(portable_state_dict(model.state_dict(), seed=0))
model = model.to("cuda:0")
b = make_batch("c4")


def run(precision, w32):
    _lib.set_option("FS2_ATTN_W32", w32)
    model.precision = precision
    with torch.no_grad():
        r = model._run(b["xs"].cuda(), b["ilens"], is_inference=True, d_override=b["ds"].cuda(), want=("after", "qe", "qp", "e_outs", "p_outs"))
    model.precision = "fp32"
    return {k: r[k].cpu() for k in ("after", "qe", "qp", "e_outs", "p_outs")}


ref = run("fp32", -1)
for prec, w32 in (("bf16x3", 1), ("bf16x3", 0), ("mix_mx", -1)):
    r = run(prec, w32)
    bad = []
    for i in range(256):
        L = int(b["olens"][i])
        d = float((r["after"][i, :L] - ref["after"][i, :L]).abs().max())
        fl = int((r["qe"][i, :L] != ref["qe"][i, :L]).sum() + (r["qp"][i, :L] != ref["qp"][i, :L]).sum())
        de = float((r["e_outs"][i, :L] - ref["e_outs"][i, :L]).abs().max())
        dp = float((r["p_outs"][i, :L] - ref["p_outs"][i, :L]).abs().max())
        if d > 1e-3 or fl:
            bad.append((i, L, d, fl, de, dp))
    print("%s w32=%d: %d utterances beyond 1e-3 or with flips" % (prec, w32, len(bad)))
    for t in bad:
        print("   utt %3d L %4d mel err %.3e flips %d  predictor err energy %.2e pitch %.2e" % t)
        if t[3] == 0:
            i, L = t[0], t[1]
            e = (r["after"][i, :L] - ref["after"][i, :L]).abs().amax(-1)
            nz = torch.nonzero(e > 1e-3).flatten()
            print("      frames beyond 1e-3: %d, first %d last %d; max at frame %d" % (len(nz), int(nz[0]), int(nz[-1]), int(e.argmax())))

# run-to-run: the same mode ten times, bit-identical?
base = run("bf16x3", 1)["after"]
ndiff = 0
for k in range(10):
    o = run("bf16x3", 1)["after"]
    if not torch.equal(o, base):
        ndiff += 1
        e = (o - base).abs().amax(-1)
        nz = torch.nonzero(e > 0)
        print("   run %d differs: %d frames, utterances %s" % (k, len(nz), sorted(set(nz[:, 0].tolist()))))
print("bf16x3 w32=1, 10 repeats: %d differ from the first" % ndiff)
