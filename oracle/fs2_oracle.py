"""CPU oracle for the FastSpeech2 mel-generation forward pass.

TEST INFRASTRUCTURE ONLY.  Nothing under ``fastspeech2_amd/`` may import this file; it is
used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` as the *checker* and as the timed CPU port, never as the product path.

It restates, in plain fp32 PyTorch on the CPU, the arithmetic of the reference's
``FeedForwardTransformer._forward`` (reference fastspeech.py:169-243) from the spec in
SURVEY.md Appendix A.  It is pinned against the real reference by ``oracle/gen_golden.py``
(run in the build container, where /root/reference is importable), which writes the
fixtures under ``tests/golden/`` that ``tests/test_oracle_golden.py`` replays everywhere.
Parity status: PINNED by those fixtures (G1-G5); the reference's own test-suite holds no
golden vectors for this path, so that is the only pin there is.

Inputs are a reference-layout ``state_dict`` (same key names, Appendix C) and a small
config dict; no nn.Module is involved.
"""
import math

import torch
import torch.nn.functional as F


def config_from_hp(hp, idim, odim, script_twin=False):
    """Collect the fields the path reads (reference fastspeech.py:53-160).  script_twin: the architecture of
    reference utils/fastspeech2_script.py:112-145 (decoder dim = adim, PE-only decoder input layer)."""
    m = hp.model
    return dict(dec_input_linear=not script_twin,
        idim=idim, odim=odim, adim=m.adim, aheads=m.aheads, elayers=m.elayers, eunits=m.eunits,
        ddim=m.ddim, dlayers=m.dlayers, dunits=m.dunits,
        ffn_kernel=(m.positionwise_conv_kernel_size if m.positionwise_layer_type == "conv1d" else 1),
        ffn_linear=(m.positionwise_layer_type == "linear"),
        postnet_layers=m.postnet_layers, use_batch_norm=bool(m.use_batch_norm),
        use_scaled_pos_enc=bool(m.use_scaled_pos_enc), reduction_factor=m.reduction_factor,
        dur_layers=m.duration_predictor_layers, var_layers=2,
        enc_pre_ln=bool(m.encoder_normalize_before), dec_pre_ln=bool(m.decoder_normalize_before),
        enc_concat=bool(m.encoder_concat_after), dec_concat=bool(m.decoder_concat_after),
    )


def positional_table(n, d):
    """pe[t,2i]=sin(t*w_i), pe[t,2i+1]=cos(t*w_i)   (reference core/embedding.py:57-66)."""
    pos = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(n, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _pe(sd, prefix, n, d):
    pe = sd.get(prefix + ".pe")
    if pe is None or pe.shape[1] < n:
        return positional_table(n, d)
    return pe[0, :n].float()


def _add_pos(sd, prefix, x, cfg):
    d = x.shape[-1]
    pe = _pe(sd, prefix, x.shape[1], d)
    if cfg["use_scaled_pos_enc"]:
        return x + sd[prefix + ".alpha"] * pe          # embedding.py:115-120 (no xscale)
    return x * math.sqrt(d) + pe                       # embedding.py:77-80


def _mha(sd, p, x, mask, heads):
    """reference core/attention.py:30-74.  mask: [B,T,T] bool (True=keep) or None."""
    B, T, D = x.shape
    dk = D // heads
    q = F.linear(x, sd[p + ".linear_q.weight"], sd[p + ".linear_q.bias"]).view(B, T, heads, dk).transpose(1, 2)
    k = F.linear(x, sd[p + ".linear_k.weight"], sd[p + ".linear_k.bias"]).view(B, T, heads, dk).transpose(1, 2)
    v = F.linear(x, sd[p + ".linear_v.weight"], sd[p + ".linear_v.bias"]).view(B, T, heads, dk).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    if mask is not None:
        drop = ~mask.unsqueeze(1)
        s = s.masked_fill(drop, float("-inf"))
        a = torch.softmax(s, dim=-1).masked_fill(drop, 0.0)   # NaN rows (all -inf) become 0
    else:
        a = torch.softmax(s, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).contiguous().view(B, T, D)
    return F.linear(o, sd[p + ".linear_out.weight"], sd[p + ".linear_out.bias"])


def _ffn(sd, p, x, cfg):
    """conv: reference core/modules.py:237-248; linear: modules.py:186-201."""
    w1, b1 = sd[p + ".w_1.weight"], sd[p + ".w_1.bias"]
    w2, b2 = sd[p + ".w_2.weight"], sd[p + ".w_2.bias"]
    if w1.dim() == 2:
        return F.linear(torch.relu(F.linear(x, w1, b1)), w2, b2)
    k = w1.shape[-1]
    h = torch.relu(F.conv1d(x.transpose(1, 2), w1, b1, padding=(k - 1) // 2))
    return F.conv1d(h, w2, b2).transpose(1, 2)


def _fft_stack(sd, prefix, x, mask, nlayers, heads, cfg, pre_ln=False, concat=False):
    """FFT blocks, reference core/encoder.py:46-71 (default: post-LN, no concat_after) and the stack's after_norm (:201-202)."""
    D = x.shape[-1]
    ln = lambda t, name: F.layer_norm(t, (D,), sd[name + ".weight"], sd[name + ".bias"], 1e-5)
    for i in range(nlayers):
        p = "%s.encoders_.%d" % (prefix, i)
        residual = x
        if pre_ln:
            x = ln(x, p + ".norm1")
        att = _mha(sd, p + ".self_attn", x, mask, heads)
        if concat:
            x = residual + F.linear(torch.cat((x, att), dim=-1), sd[p + ".concat_linear.weight"], sd[p + ".concat_linear.bias"])
        else:
            x = residual + att
        if not pre_ln:
            x = ln(x, p + ".norm1")
        residual = x
        if pre_ln:
            x = ln(x, p + ".norm2")
        x = residual + _ffn(sd, p + ".feed_forward", x, cfg)
        if not pre_ln:
            x = ln(x, p + ".norm2")
    if pre_ln:
        x = ln(x, prefix + ".after_norm")
    return x


def _predictor(sd, p, x, nlayers):
    """conv-k -> ReLU -> channel-LN(eps 1e-12) stack + Linear(.,1)
    (reference variance_predictor.py:46-51, duration_predictor.py:70-75, modules.py:112-120)."""
    h = x.transpose(1, 2)
    for l in range(nlayers):
        w, b = sd["%s.conv.%d.0.weight" % (p, l)], sd["%s.conv.%d.0.bias" % (p, l)]
        h = torch.relu(F.conv1d(h, w, b, padding=(w.shape[-1] - 1) // 2))
        h = F.layer_norm(h.transpose(1, 2), (h.shape[1],),
                         sd["%s.conv.%d.2.layer_norm.weight" % (p, l)],
                         sd["%s.conv.%d.2.layer_norm.bias" % (p, l)], 1e-12).transpose(1, 2)
    return F.linear(h.transpose(1, 2), sd[p + ".linear.weight"], sd[p + ".linear.bias"]).squeeze(-1)


def duration_from_log(y):
    """clamp(round_half_even(exp(y) - 1), 0) -> int64  (duration_predictor.py:77-81)."""
    return torch.clamp(torch.round(y.exp() - 1.0), min=0).long()


def bucketize(x, bins):
    """first i with x <= bins[i]; NaN -> len(bins)   (variance_predictor.py:158,231)."""
    return torch.bucketize(x, bins)


def lr_indices(d):
    """Token index of every output frame for one utterance's durations d[T] (int64).

    Restates length_regulator.py:85-95: an all-zero row is replaced by ones, tokens with
    d==0 are skipped, token i is repeated d[i] times."""
    d = d.clone()
    if int(d.sum()) == 0:
        d.fill_(1)
    return torch.repeat_interleave(torch.arange(d.numel()), d)


def length_regulate(hs, ds, ilens):
    """-> (padded [B,Lmax,D], olens[B], list of index vectors)."""
    outs, idxs = [], []
    for b in range(hs.shape[0]):
        T = int(ilens[b])
        idx = lr_indices(ds[b, :T])
        idxs.append(idx)
        outs.append(hs[b, :T][idx])
    olens = torch.tensor([o.shape[0] for o in outs], dtype=torch.long)
    Lmax = int(olens.max())
    out = hs.new_zeros(len(outs), Lmax, hs.shape[-1])
    for b, o in enumerate(outs):
        out[b, : o.shape[0]] = o
    return out, olens, idxs


def _len_mask(lens, n):
    return torch.arange(n).unsqueeze(0) < torch.as_tensor(lens).view(-1, 1)


def _postnet(sd, x, cfg):
    """reference core/modules.py:285-358, eval-mode BatchNorm (eps 1e-5)."""
    n = cfg["postnet_layers"]
    for l in range(n):
        w = sd["postnet.postnet.%d.0.weight" % l]
        x = F.conv1d(x, w, None, padding=(w.shape[-1] - 1) // 2)
        if cfg["use_batch_norm"]:
            p = "postnet.postnet.%d.1" % l
            x = F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                             sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)
        if l != n - 1:
            x = torch.tanh(x)
    return x


@torch.no_grad()
def padded_forward(sd, cfg, xs, ilens, olens=None, ds=None, es=None, ps=None, is_inference=False,
                   d_override=None):
    """Exactly the reference's padded-batch ``_forward`` (fastspeech.py:169-243).

    Returns a dict with the 5-tuple members plus intermediates used by parity tests.
    ``d_override`` (int64 [B,Tmax]) replaces the duration predictor's output in inference
    mode (what the benchmark uses to force LJSpeech-like durations)."""
    ilens = torch.as_tensor(ilens, dtype=torch.long)
    B, Tmax = xs.shape
    heads = cfg["aheads"]
    valid = _len_mask(ilens, Tmax)
    x_mask = valid.unsqueeze(-2) & valid.unsqueeze(-1)
    h = F.embedding(xs, sd["encoder.embed.0.weight"])
    h = _add_pos(sd, "encoder.embed.1", h, cfg)
    hs = _fft_stack(sd, "encoder", h, x_mask, cfg["elayers"], heads, cfg, cfg.get("enc_pre_ln", False), cfg.get("enc_concat", False))
    out = {"encoder_out": hs}
    d_pad = ~valid
    dlog = _predictor(sd, "duration_predictor", hs, cfg["dur_layers"])
    if is_inference:
        d_outs = duration_from_log(dlog).masked_fill(d_pad, 0)
        if d_override is not None:
            d_outs = d_override.clone()
        hs_f, olens_c, idxs = length_regulate(hs, d_outs, ilens)
        e_pred = _predictor(sd, "energy_predictor.predictor", hs_f, cfg["var_layers"])
        p_pred = _predictor(sd, "pitch_predictor.predictor", hs_f, cfg["var_layers"])
        qe = bucketize(e_pred, sd["energy_predictor.energy_bins"])
        qp = bucketize(p_pred, sd["pitch_predictor.pitch_bins"])
        e_outs, p_outs = e_pred, p_pred
    else:
        qe = bucketize(es, sd["energy_predictor.energy_bins"])
        qp = bucketize(ps, sd["pitch_predictor.pitch_bins"])
        d_outs = dlog.masked_fill(d_pad, 0.0)
        hs_f, olens_c, idxs = length_regulate(hs, ds, ilens)
        mel_pad = ~_len_mask(olens, hs_f.shape[1])
        e_outs = _predictor(sd, "energy_predictor.predictor", hs_f, cfg["var_layers"]).masked_fill(mel_pad, 0.0)
        p_outs = _predictor(sd, "pitch_predictor.predictor", hs_f, cfg["var_layers"]).masked_fill(mel_pad, 0.0)
    out.update(lr_index=idxs, olens=olens_c, lr_out=hs_f, qe=qe, qp=qp)
    # one_hot(q) @ W^T + b  ==  W[:, q] + b   (bit-equal)
    hs_f = hs_f + (sd["pitch_embed.weight"].t()[qp] + sd["pitch_embed.bias"])
    hs_f = hs_f + (sd["energy_embed.weight"].t()[qe] + sd["energy_embed.bias"])
    out["decoder_in"] = hs_f
    if olens is not None:
        ov = _len_mask(torch.as_tensor(olens, dtype=torch.long), hs_f.shape[1])
        h_mask = ov.unsqueeze(-2) & ov.unsqueeze(-1)
    else:
        h_mask = None
    if cfg.get("dec_input_linear", True):
        z = F.linear(hs_f, sd["decoder.embed.0.weight"], sd["decoder.embed.0.bias"])
        z = F.layer_norm(z, (z.shape[-1],), sd["decoder.embed.1.weight"], sd["decoder.embed.1.bias"], 1e-5)
        z = _add_pos(sd, "decoder.embed.4", torch.relu(z), cfg)
    else:   # utils/fastspeech2_script.py:112-127: input_layer=None -> embed = Sequential(pos_enc)
        z = _add_pos(sd, "decoder.embed.0", hs_f, cfg)
    z = _fft_stack(sd, "decoder", z, h_mask, cfg["dlayers"], heads, cfg, cfg.get("dec_pre_ln", False), cfg.get("dec_concat", False))
    out["decoder_out"] = z
    before = F.linear(z, sd["feat_out.weight"], sd["feat_out.bias"]).view(B, -1, cfg["odim"])
    if cfg["postnet_layers"] > 0:
        after = before + _postnet(sd, before.transpose(1, 2), cfg).transpose(1, 2)
    else:
        after = before
    out.update(before=before, after=after, d_outs=d_outs, e_outs=e_outs, p_outs=p_outs)
    return out


@torch.no_grad()
def per_utterance_forward(sd, cfg, xs, ilens, ds=None, es=None, ps=None, is_inference=False,
                          d_override=None):
    """Batch-invariant semantics: every utterance is pushed through ``padded_forward`` alone
    (B=1, no padding), i.e. exactly what the reference's ``inference()`` /
    ``_forward(B=1)`` computes, and the results are re-padded with zeros."""
    ilens = torch.as_tensor(ilens, dtype=torch.long)
    B = xs.shape[0]
    res = []
    for b in range(B):
        T = int(ilens[b])
        kw = {}
        if not is_inference:
            idx = lr_indices(ds[b, :T])
            L = idx.numel()
            kw = dict(olens=torch.tensor([L]), ds=ds[b:b + 1, :T], es=es[b:b + 1, :L], ps=ps[b:b + 1, :L])
        elif d_override is not None:
            kw = dict(d_override=d_override[b:b + 1, :T])
        res.append(padded_forward(sd, cfg, xs[b:b + 1, :T], ilens[b:b + 1], is_inference=is_inference, **kw))
    olens = torch.cat([r["olens"] for r in res])
    Lmax, Tmax = int(olens.max()), xs.shape[1]

    def pad(key, n):
        first = res[0][key]
        o = first.new_zeros((B, n) + tuple(first.shape[2:]))
        for b, r in enumerate(res):
            o[b, : r[key].shape[1]] = r[key][0]
        return o

    out = dict(olens=olens, lr_index=[r["lr_index"][0] for r in res])
    rf = int(cfg.get("reduction_factor", 1))     # feat_out emits r mel frames per decoder frame (reference fastspeech.py:228-230)
    for key in ("before", "after"):
        out[key] = pad(key, Lmax * rf)
    for key in ("e_outs", "p_outs", "qe", "qp", "decoder_out", "lr_out", "decoder_in"):
        out[key] = pad(key, Lmax)
    for key in ("d_outs", "encoder_out"):
        out[key] = pad(key, Tmax)
    return out


def flops(T, L):
    """Algorithmic FLOPs of one utterance, default dims."""
    return T * (23855616 + 4096 * T) + L * (40383488 + 6144 * L)


def loss_report(out, ys, ilens, olens, ds, es, ps, use_masking=True, use_weighted_masking=False):
    """The reference's ``forward()`` loss algebra (fastspeech.py:281-333).  ``use_weighted_masking`` re-weights the already
    mean-reduced l1 / duration losses exactly as the reference does (:308-325; it needs the un-selected 3-D ``ys``, i.e.
    ``use_masking=False`` -- with both flags the reference raises IndexError on ``ys.size(2)``)."""
    im = _len_mask(ilens, out["d_outs"].shape[1])
    om = _len_mask(olens, out["before"].shape[1])
    before, after, y, d_outs, ds_t, e_outs, p_outs = out["before"], out["after"], ys, out["d_outs"], ds, out["e_outs"], out["p_outs"]
    es, ps = es[:, : before.shape[1]], ps[:, : before.shape[1]]
    if use_masking:
        before = before.masked_select(om.unsqueeze(-1))
        after = after.masked_select(om.unsqueeze(-1))
        y = ys.masked_select(om.unsqueeze(-1))
        d_outs, ds_t = d_outs.masked_select(im), ds.masked_select(im)
        e_outs, p_outs, es, ps = e_outs.masked_select(om), p_outs.masked_select(om), es.masked_select(om), ps.masked_select(om)
    before_loss = F.l1_loss(before, y)
    after_loss = F.l1_loss(after, y)
    l1 = before_loss + after_loss
    dur = F.mse_loss(d_outs, torch.log(ds_t.float() + 1.0))
    en = F.mse_loss(e_outs, es)
    pi = F.mse_loss(p_outs, ps)
    if use_weighted_masking:
        out_masks = om.unsqueeze(-1)
        out_weights = out_masks.float() / out_masks.sum(dim=1, keepdim=True).float()
        out_weights = out_weights / (y.size(0) * y.size(2))
        dw = im.float() / im.sum(dim=1, keepdim=True).float()
        dw = dw / ds_t.size(0)
        l1 = l1.mul(out_weights).masked_select(out_masks).sum()
        dur = dur.mul(dw).masked_select(im).sum()
    loss = l1 + dur + en + pi
    return loss, [{"l1_loss": l1.item()}, {"before_loss": before_loss.item()}, {"after_loss": after_loss.item()},
                  {"duration_loss": dur.item()}, {"energy_loss": en.item()}, {"pitch_loss": pi.item()},
                  {"loss": loss.item()}]
