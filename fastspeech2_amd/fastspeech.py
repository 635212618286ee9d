"""Drop-in ``FeedForwardTransformer`` whose tensor work runs in libfs2_hip.so on an MI355X.

Mirrors the public surface of the reference module (reference fastspeech.py:28-387):
``FeedForwardTransformer(idim, odim, hp)``, ``forward(xs, ilens, ys, olens, ds, es, ps)``,
``inference(x)``, ``_forward(xs, ilens, olens, ds, es, ps, is_inference)``, and the same
``state_dict()`` key names / shapes, so checkpoints and the
reference's ``inference.py`` / ``evaluation.py`` call sites work unchanged.

The sub-modules below are *parameter containers only* (standard torch layers used for
their parameter shapes, default init and state-dict names); none of their ``forward``
methods is ever called.  All arithmetic happens in the HIP library through the C ABI in
``include/fs2.h``; there is no CPU / eager fallback: a CPU tensor, a missing library or
training-mode autograd raises.
"""
import ctypes as C
import math

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib

__all__ = ["FeedForwardTransformer", "AsyncMels", "Fs2CapacityError", "StepStreams"]


class Fs2CapacityError(RuntimeError):
    """A capacity of the device-driven frame layout was too small: the mels of that call are NaN-filled (invalid)."""


FS2_OVF_BAD_ID = 16      # include/fs2.h: an utterance holds a phoneme id outside [0, idim)


class Fs2IndexError(IndexError):
    """A phoneme id outside [0, idim): what torch.nn.Embedding raises for in the reference (fastspeech.py:65-67, core/encoder.py:196).  The
    synchronous entry points raise it before any mel is returned; an asynchronous call reports it through ``AsyncMels.check()`` /
    ``model.async_ok()`` / ``ShardedSynthesizer.ok()`` (its mels are NaN-filled)."""


class AsyncMels(tuple):
    """What ``inference_batch(sync=False)`` returns: unpacks like ``(mels, olens_dev)`` and carries the call's own validity
    record.  ``status`` is the device int32[8] of fs2_decode ({rows, work items, overflow flags, longest utterance, valid frames,
    ...}); ``ok()`` waits for THIS call (its own event and pinned copy of the flags) and tells whether its capacities
    sufficed; ``check()`` raises ``Fs2CapacityError`` instead.  On overflow the mels are NaN-filled on the device, so a caller
    that never looks still cannot mistake them for audio."""

    def __new__(cls, mels, olens, status, record):
        self = super().__new__(cls, (mels, olens))
        self.status, self._record = status, record
        return self

    def ok(self):
        return self._record is None or self._record.flags(block=True) == 0

    def check(self):
        if not self.ok():
            fl = self._record.flags(block=True)
            if fl & FS2_OVF_BAD_ID:
                raise Fs2IndexError("a phoneme id of this batch lies outside [0, idim) (status flags %d); the mels of the call are NaN-filled" % fl)
            raise Fs2CapacityError("device-driven layout overflow (flags %d): rerun this batch with sync=True or a larger "
                                   "capacity" % fl)
        return self


class _AsyncRecord:
    """Frame counts and flags of one asynchronous call, copied to pinned host memory behind its kernels.  The pinned slot belongs
    to a ring and is re-used by later calls: the first look after the event has fired moves the call's own values into the
    record (``_record_async`` folds a record before it hands its slot to another call), so ``flags()`` / ``olens`` never show a
    later call's status."""
    __slots__ = ("il", "event", "olens_pin", "status_pin", "harvested", "_flags", "olens")

    def __init__(self, il, event, olens_pin, status_pin):
        self.il, self.event, self.olens_pin, self.status_pin, self.harvested = il, event, olens_pin, status_pin, False
        self._flags, self.olens = None, None

    def flags(self, block):
        if self._flags is None:
            if block:
                self.event.synchronize()
            elif not self.event.query():
                return None
            self._flags = int(self.status_pin[2])
            self.olens = self.olens_pin[: self.il[0].numel()].clone()
            self.olens_pin = self.status_pin = None
        return self._flags


# ----------------------------------------------------------------------------------------------
# parameter containers (names fixed by the reference's state-dict layout)
# ----------------------------------------------------------------------------------------------
class StepStreams:
    """Throughput mode for a caller with independent batches queued: ``n`` streams handed out in turn, so that ``n`` whole forwards are in flight and
    the tail round, HBM burst and launch gap of one forward's kernels are filled by another's workgroups (c3: 7.28 -> 8.84 M mel frames/s with
    n = 3, profiles/r05_ab_stream_schedules.txt; more streams than the runtime has hardware queues -- 4 by default, the null stream included -- lose
    again).  Nothing in the library knows about it -- a sync-free call runs on its caller's current stream::

        rot = StepStreams(3)
        for xs, ilens in batches:
            with rot.next():
                results.append(model.inference_batch(xs, ilens, sync=False))
        rot.join()                    # the current stream waits for all of them; then model.async_ok()

    ``next()`` makes the stream it hands out wait for what the current stream has queued so far (the batch's inputs); results belong to the stream
    they were computed on until ``join()`` (keep them alive, or ``record_stream`` them, while another stream reads them)."""

    def __init__(self, n=3, device=None):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(int(n), 1))]
        self._i = 0

    def next(self):
        s = self.streams[self._i % len(self.streams)]
        self._i += 1
        s.wait_stream(torch.cuda.current_stream(s.device))
        return torch.cuda.stream(s)

    def join(self):
        for s in self.streams:
            torch.cuda.current_stream(s.device).wait_stream(s)


def sinusoid_table(n, d):
    """pe[t,2i]=sin(t*exp(-2i*ln(1e4)/d)), pe[t,2i+1]=cos(.)  (reference core/embedding.py:57-66).
    Built on the host once (trig tables belong on the host, not in the streaming kernel)."""
    pos = torch.arange(0, n, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe = torch.zeros(n, d)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


class _Holder(nn.Module):
    @torch.jit.unused          # keeps the containers (and the nn.Sequential wrappers around them) scriptable
    def forward(self, x: torch.Tensor) -> torch.Tensor:  # pragma: no cover
        raise RuntimeError("parameter container: the forward pass runs in libfs2_hip.so")


class _PositionalTable(_Holder):
    """`pe` persistent buffer [1, 5000, d] (+ learnable `alpha` when scaled), embedding.py:28-120."""

    def __init__(self, d, scaled, max_len=5000):
        super().__init__()
        self.d_model = d
        self.register_buffer("pe", sinusoid_table(max_len, d))
        if scaled:
            self.alpha = nn.Parameter(torch.tensor(1.0))

    def ensure(self, n):
        """reference extend_pe(): recompute a longer table when needed; returns True if the table changed."""
        if self.pe.shape[1] < n:
            self.pe = sinusoid_table(n, self.d_model).to(self.pe.device)
            return True
        return False


class _SelfAttention(_Holder):
    def __init__(self, d):
        super().__init__()
        self.linear_q, self.linear_k = nn.Linear(d, d), nn.Linear(d, d)
        self.linear_v, self.linear_out = nn.Linear(d, d), nn.Linear(d, d)


class _FeedForward(_Holder):
    def __init__(self, d, units, kernel, conv):
        super().__init__()
        if conv:
            self.w_1 = nn.Conv1d(d, units, kernel, padding=(kernel - 1) // 2)
            self.w_2 = nn.Conv1d(units, d, 1)
        else:
            self.w_1, self.w_2 = nn.Linear(d, units), nn.Linear(units, d)


class _FFTBlock(_Holder):
    def __init__(self, d, units, kernel, conv):
        super().__init__()
        self.self_attn = _SelfAttention(d)
        self.feed_forward = _FeedForward(d, units, kernel, conv)
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)
        self.concat_linear = nn.Linear(2 * d, d)      # present in checkpoints, unused (concat_after=False)


class _FFTStack(_Holder):
    """core/encoder.py:74-204: `embed` + `encoders_` (+ unused `after_norm`)."""

    def __init__(self, embed, d, units, nblocks, kernel, conv):
        super().__init__()
        self.after_norm = nn.LayerNorm(d)
        self.embed = embed
        self.encoders_ = nn.ModuleList([_FFTBlock(d, units, kernel, conv) for _ in range(nblocks)])


class _ChannelLayerNorm(_Holder):
    def __init__(self, n):
        super().__init__()
        self.layer_norm = nn.LayerNorm(n, eps=1e-12)


class _ConvStackPredictor(_Holder):
    """duration_predictor.py:14-86 / variance_predictor.py:7-95."""

    def __init__(self, idim, n_layers=2, n_chans=256, kernel_size=3):
        super().__init__()
        self.conv = nn.ModuleList()
        for i in range(n_layers):
            self.conv.append(nn.Sequential(
                nn.Conv1d(idim if i == 0 else n_chans, n_chans, kernel_size, padding=(kernel_size - 1) // 2),
                nn.ReLU(), _ChannelLayerNorm(n_chans), nn.Dropout(0.5)))
        self.linear = nn.Linear(n_chans, 1)


class _QuantisingPredictor(_Holder):
    """Energy/Pitch predictor: a bins buffer + a default-shaped conv stack (the reference ignores
    its ctor args here, variance_predictor.py:125,198)."""

    def __init__(self, idim, bins_name, bins):
        super().__init__()
        self._bins_name = bins_name
        self.register_buffer(bins_name, bins)
        self.predictor = _ConvStackPredictor(idim)

    def to_one_hot(self, x):
        """bucketize + one-hot on the caller's device (variance_predictor.py:154-159,227-232)."""
        return F.one_hot(_bucketize(x, getattr(self, self._bins_name)).long(), 256).float()


class _Postnet(_Holder):
    def __init__(self, odim, n_layers, n_chans, n_filts, use_bn):
        super().__init__()
        self.postnet = nn.ModuleList()
        for l in range(n_layers):
            ic = odim if l == 0 else n_chans
            oc = odim if l == n_layers - 1 else n_chans
            mods = [nn.Conv1d(ic, oc, n_filts, padding=(n_filts - 1) // 2, bias=False)]
            if use_bn:
                mods.append(nn.BatchNorm1d(oc))
            if l != n_layers - 1:
                mods.append(nn.Tanh())
            mods.append(nn.Dropout(0.5))
            self.postnet.append(nn.Sequential(*mods))


def _bucketize(x, bins):
    """torch.bucketize semantics through the HIP kernel (device tensors only)."""
    _require_device(x)
    x = x.contiguous().float()
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    L = _lib.lib()
    with torch.cuda.device(x.device):
        _lib.check(L.fs2_op_bucketize(_stream(x.device), x.data_ptr(), x.numel(), bins.contiguous().data_ptr(),
                                      bins.numel(), out.data_ptr()))
    return out


def _require_device(t):
    if not t.is_cuda:
        raise RuntimeError("fastspeech2_amd runs on an MI355X only: got a %s tensor (no CPU fallback; "
                           "move the module and its inputs to a HIP device)" % t.device)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# ----------------------------------------------------------------------------------------------
class FeedForwardTransformer(nn.Module):
    """Feed-forward Transformer TTS acoustic model (FastSpeech 2) on libfs2_hip.

    Extra (non-reference) knobs, all optional:
      * ``precision``: "fp32" (f32-input MFMA, default), "bf16x3", "bf16".
      * ``batch_semantics``: "per_utterance" (default; every utterance is computed exactly as if it
        were alone -- what ``inference()`` produces; batch invariant, shardable) or "padded_compat"
        (bit-for-bit the reference's padded-batch behaviour where convolutions and unmasked attention
        see pad rows).  ``forward()`` (the loss path) always uses "padded_compat".
      * ``overlap_encoder`` (default False): throughput mode of the sync-free entry points -- each call's token-level half (encoder +
        duration predictor) runs on a side stream and overlaps the previous call's frame-level kernels; results are unchanged; the
        caller prepares the ids of a call under ``input_stream(device)``, guarantees they are complete, or hands the call an event
        recorded behind the work that produces them (``inference_batch(..., inputs_ready=event)``: the side stream waits for it).

    Pickling / ``copy.deepcopy`` / ``torch.save(model)`` carry the parameters and the knobs above, never the per-process runtime state (the
    library handle, side streams, pinned staging slots, calls in flight): the copy builds its own on first use.
    """

    def __init__(self, idim: int, odim: int, hp, _script_twin: bool = False):
        super().__init__()
        m = hp.model
        # _script_twin: the architecture of reference utils/fastspeech2_script.py (decoder dim = adim, positional
        # encoding as the only decoder input layer, feat_out Linear(adim, odim)); see fastspeech2_script.py here
        ddim = m.adim if _script_twin else m.ddim
        self.idim, self.odim = idim, odim
        # throughput mode of the sync-free entry points (inference_batch(sync=False), ShardedSynthesizer): run each call's token-level half on a side
        # stream so that it overlaps the previous call's frame-level kernels.  The caller then guarantees that the input ids (and d_override) of a call
        # are COMPLETE when the call is made -- not the product of work still queued on the current stream.  Off by default; results are unchanged.
        self.overlap_encoder = False
        self._enc_streams = {}          # (device, caller's stream) -> the side stream of the calls made on that stream
        self._enc_gen_seen = {}
        self.use_scaled_pos_enc = bool(m.use_scaled_pos_enc)
        self.use_masking = bool(m.use_masking)
        self.use_weighted_masking = bool(m.use_weighted_masking)
        if m.positionwise_layer_type not in ("conv1d", "linear"):
            raise NotImplementedError("Support only linear or conv1d.")
        if not 1 <= int(m.reduction_factor) <= 8:
            raise NotImplementedError("reduction_factor outside [1, 8]")
        self.reduction_factor = int(m.reduction_factor)      # feat_out emits r mel frames per decoder frame (reference fastspeech.py:153,228-230)
        conv = m.positionwise_layer_type == "conv1d"
        kernel = m.positionwise_conv_kernel_size if conv else 1
        self._cfg = dict(
            idim=idim, odim=odim, adim=m.adim, aheads=m.aheads, elayers=m.elayers, eunits=m.eunits, ddim=ddim,
            dlayers=m.dlayers, dunits=m.dunits, ffn_kernel=kernel,
            dur_layers=m.duration_predictor_layers, dur_chans=m.duration_predictor_chans,
            dur_kernel=m.duration_predictor_kernel_size, var_layers=2, var_chans=256, var_kernel=3, n_bins=256,
            postnet_layers=m.postnet_layers, postnet_chans=m.postnet_chans, postnet_filts=m.postnet_filts,
            use_batch_norm=int(bool(m.use_batch_norm)), use_scaled_pos_enc=int(self.use_scaled_pos_enc),
            reduction_factor=m.reduction_factor, decoder_input_layer=0 if _script_twin else 1,
            enc_normalize_before=int(bool(m.encoder_normalize_before)), dec_normalize_before=int(bool(m.decoder_normalize_before)),
            enc_concat_after=int(bool(m.encoder_concat_after)), dec_concat_after=int(bool(m.decoder_concat_after)))

        enc_embed = nn.Sequential(nn.Embedding(idim, m.adim, padding_idx=0), _PositionalTable(m.adim, self.use_scaled_pos_enc))
        self.encoder = _FFTStack(enc_embed, m.adim, m.eunits, m.elayers, kernel, conv)
        self.duration_predictor = _ConvStackPredictor(m.adim, m.duration_predictor_layers, m.duration_predictor_chans,
                                                      m.duration_predictor_kernel_size)
        self.energy_predictor = _QuantisingPredictor(m.adim, "energy_bins",
                                                     torch.linspace(hp.data.e_min, hp.data.e_max, 255))
        self.energy_embed = nn.Linear(m.adim, m.adim)
        self.pitch_predictor = _QuantisingPredictor(
            m.adim, "pitch_bins",
            torch.exp(torch.linspace(torch.log(torch.tensor(float(hp.data.p_min))),
                                     torch.log(torch.tensor(float(hp.data.p_max))), 255)))
        self.pitch_embed = nn.Linear(m.adim, m.adim)
        if _script_twin:
            dec_embed = nn.Sequential(_PositionalTable(ddim, self.use_scaled_pos_enc))
        else:
            dec_embed = nn.Sequential(nn.Linear(m.adim, ddim), nn.LayerNorm(ddim), nn.Dropout(0.2), nn.ReLU(),
                                      _PositionalTable(ddim, self.use_scaled_pos_enc))
        self.decoder = _FFTStack(dec_embed, ddim, m.dunits, m.dlayers, kernel, conv)
        self.postnet = None if m.postnet_layers == 0 else _Postnet(odim, m.postnet_layers, m.postnet_chans,
                                                                   m.postnet_filts, bool(m.use_batch_norm))
        self.feat_out = nn.Linear(ddim, odim * m.reduction_factor)
        self._reset_parameters(m.transformer_init, m.initial_encoder_alpha, m.initial_decoder_alpha)

        self.precision = "fp32"
        self.batch_semantics = "per_utterance"
        self._handle = None
        self._handle_device = None
        self._fingerprint = None
        self._weights_generation = 0
        self._pending = []             # _AsyncRecord of asynchronous calls not yet folded into the capacity predictor
        self._pin_ring, self._pin_next = [], 0
        self.last_olens = None

    # ------------------------------------------------------------------ init (fastspeech.py:378-387)
    def _reset_parameters(self, init_type, init_enc_alpha=1.0, init_dec_alpha=1.0):
        if init_type != "pytorch":
            fn = {"xavier_uniform": nn.init.xavier_uniform_, "xavier_normal": nn.init.xavier_normal_,
                  "kaiming_uniform": lambda p: nn.init.kaiming_uniform_(p, nonlinearity="relu"),
                  "kaiming_normal": lambda p: nn.init.kaiming_normal_(p, nonlinearity="relu")}.get(init_type)
            if fn is None:
                raise ValueError("Unknown initialization: " + init_type)
            for p in self.parameters():
                if p.dim() > 1:
                    fn(p.data)
                elif p.dim() == 1:
                    p.data.zero_()
            for mod in self.modules():
                if isinstance(mod, (nn.Embedding, nn.LayerNorm)):
                    mod.reset_parameters()
        if self.use_scaled_pos_enc:
            self.encoder.embed[-1].alpha.data = torch.tensor(float(init_enc_alpha))
            self.decoder.embed[-1].alpha.data = torch.tensor(float(init_dec_alpha))

    # per-process runtime state: never pickled / deep-copied (a ctypes handle, HIP streams and events, pinned host memory; round-5 advisor finding)
    _RUNTIME_STATE = dict(_handle=None, _handle_device=None, _fingerprint=None, _weights_generation=0, _fp_refs=None, _fp_pes=None,
                          last_async=None, last_olens=None)

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in list(self._RUNTIME_STATE) + ["_enc_streams", "_enc_gen_seen", "_pending", "_pin_ring", "_pin_next"]:
            st.pop(k, None)
        return st

    def __setstate__(self, st):
        self.__dict__.update(st)
        self.__dict__.update(self._RUNTIME_STATE)
        self.__dict__.update(_enc_streams={}, _enc_gen_seen={}, _pending=[], _pin_ring=[], _pin_next=0)

    # ------------------------------------------------------------------ library handle / weights
    def __del__(self):
        try:
            self._drop_handle()
        except Exception:       # interpreter shutdown: torch's own module machinery may already be torn down
            pass

    def _drop_handle(self):
        h = getattr(self, "_handle", None)
        if h is not None:
            try:
                _lib.lib().fs2_destroy(h)      # restores the caller's current device itself (DeviceGuard in the library)
            except Exception:   # interpreter shutdown
                pass
            self.__dict__["_handle"] = None

    def _weights_fingerprint(self):
        """(storage, version) of every parameter / buffer.  Catches load_state_dict, .to(), optimizer steps and every in-place
        op autograd sees; writes through ``.data`` bypass the version counter: call ``refresh_weights()`` after those.
        Runs on every call, so the tensor list is cached (walking ``state_dict()`` cost 0.25 ms of a 1.3-ms single-utterance
        call); ``load_state_dict`` / ``_apply`` (``.to()``, ``.cuda()``, ``.float()``) and ``refresh_weights`` drop the cache.
        Assigning a new ``nn.Parameter`` object to a submodule needs ``refresh_weights()``."""
        refs = self.__dict__.get("_fp_refs")
        # the positional tables are REPLACED when they grow (_PositionalTable.ensure, the reference's extend_pe): a cached list
        # would keep the old tensors alive and hide the change (round-2 advisor finding)
        pes = (self.encoder.embed[-1].pe, self.decoder.embed[-1].pe)
        if refs is not None and (self.__dict__["_fp_pes"][0] is not pes[0] or self.__dict__["_fp_pes"][1] is not pes[1]):
            refs = None
        if refs is None:
            refs = list(self.state_dict(keep_vars=True).values())
            self.__dict__["_fp_refs"], self.__dict__["_fp_pes"] = refs, pes
        return tuple([(v.data_ptr(), v._version) for v in refs])

    def load_state_dict(self, *args, **kwargs):
        self.__dict__["_fp_refs"] = None
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self.__dict__["_fp_refs"] = None
        return super()._apply(fn, *args, **kwargs)

    def _ensure_ready(self, device, need_tok, need_frames=0):
        """Create the device handle and (re)upload weights whenever any parameter changed."""
        L = _lib.lib()
        p0 = self.feat_out.weight
        if p0.device != device:
            raise RuntimeError("model is on %s but the input is on %s" % (p0.device, device))
        self.encoder.embed[-1].ensure(need_tok)
        self.decoder.embed[-1].ensure(max(need_frames, 1))
        if self._handle is None or self._handle_device != device:
            self._drop_handle()
            cfg = _lib.Config(**self._cfg, device=device.index if device.index is not None else torch.cuda.current_device())
            h = C.c_void_p()
            with torch.cuda.device(device):      # (the library also restores the caller's current device itself)
                _lib.check(L.fs2_create(C.byref(cfg), C.byref(h)))
            self._handle, self._handle_device, self._fingerprint = h, device, None
        fp = self._weights_fingerprint()
        if fp != self._fingerprint:
            sd = self.state_dict()
            keep, descs = [], []
            for name, t in sd.items():
                if t.dtype != torch.float32:
                    continue   # num_batches_tracked (int64) is not used in eval mode
                t = t.detach().contiguous()
                keep.append(t)
                d = _lib.TensorDesc(name.encode(), t.data_ptr(), t.dim(), (C.c_int64 * 4)(*(list(t.shape) + [0] * (4 - t.dim()))))
                descs.append(d)
            arr = (_lib.TensorDesc * len(descs))(*descs)
            with torch.cuda.device(device):
                _lib.check(L.fs2_load_weights(self._handle, arr, len(descs), _stream(device)), self._handle)
            self._fingerprint = fp
            self._weights_generation += 1      # device copies were re-allocated: captured graphs of older generations are stale
        return L

    def refresh_weights(self):
        """Force a re-upload of the parameters on the next call (normally automatic)."""
        self._fingerprint = None
        self.__dict__["_fp_refs"] = None

    # ------------------------------------------------------------------ the path
    def _run(self, xs, ilens, olens=None, ds=None, es=None, ps=None, is_inference=False, compat=False,
             want=("before", "after"), d_override=None, capacity=None, alpha=1.0, packed_out=None, regime=None, inputs_ready=None):
        """Runs fs2_encode -> (olens readback) -> fs2_decode.  Returns a dict of device tensors.

        ``capacity=(total_frames_bound, per_utterance_bound)`` selects the device-driven frame layout instead: no host
        read-back between the two calls, nothing in this method waits for the GPU.  Outputs are then padded to
        ``per_utterance_bound`` frames, ``out["olens"]`` is the DEVICE int64 tensor and ``out["status"]`` a device
        int32[8] = {rows used, attention work items, overflow flags, longest utterance, valid frames, ...}: the results are
        valid only if ``status[2] == 0`` (see ``inference_batch(sync=False)``); ``after_packed`` then has
        ``fs2_row_capacity(total_frames_bound)`` x reduction_factor rows, of which the first ``status[4]`` x reduction_factor are filled; ``packed_out`` (a contiguous
        float32 [>= that many rows, odim] tensor) receives them in place instead of a new allocation.

        ``regime=(phonemes, utterances)`` of the batch the kernel variants are to be chosen for (include/fs2.h: fs2_batch.regime_*):
        a shard of a larger batch names the whole batch and is computed bit-identically to the one-call run of the whole batch."""
        _require_device(xs)
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError(
                "training (autograd / dropout / BatchNorm batch statistics) is outside the scope of this MI355X "
                "inference path: call model.eval() and wrap calls in torch.no_grad()")
        dev = xs.device
        # (overlap_encoder: whatever copy these conversions need -- a column-sliced shard of a batch is not contiguous -- must run on the stream the
        #  encoder runs on; on the current stream it would queue behind the previous call's kernels while the side stream reads the result at once)
        in_stream = self.input_stream(dev) if capacity is not None else torch.cuda.current_stream(dev)
        with torch.cuda.stream(in_stream):
            xs = xs.contiguous().long()
        B, Tmax = xs.shape
        il = torch.as_tensor(ilens).detach().to("cpu", torch.int64).contiguous()   # host lengths (reference: .tolist())
        if il.numel() != B:
            raise ValueError("ilens has %d entries for a batch of %d" % (il.numel(), B))
        prec = _lib.PRECISIONS[self.precision]
        L = self._ensure_ready(dev, Tmax)
        h = self._handle
        il_arr = (C.c_int64 * B)(*il.tolist())
        if regime is not None:
            rt, ru = int(regime[0]), int(regime[1])
            if rt <= 0 or ru <= 0:
                raise ValueError("regime=(phonemes, utterances) must both be positive, got %r" % (regime,))
            if compat:
                raise ValueError("regime applies to per-utterance semantics: a padded_compat batch depends on its batch-mates anyway")
            batch = _lib.Batch(B, Tmax, il_arr, int(compat), prec, rt, ru)
        else:
            batch = _lib.Batch(B, Tmax, il_arr, int(compat), prec)
        out = {}
        teacher = not is_inference
        ds_dev = None
        if teacher:
            if ds is None or es is None or ps is None:
                raise ValueError("teacher-forced _forward needs ds, es and ps")
            ds_dev = ds.to(dev).long().contiguous()
        elif d_override is not None:
            with torch.cuda.stream(in_stream):
                ds_dev = d_override.to(dev).long().contiguous()
        if ds_dev is not None and tuple(ds_dev.shape) != (B, Tmax):
            raise ValueError("ds must be [B, Tmax] = [%d, %d], got %s" % (B, Tmax, tuple(ds_dev.shape)))
        with torch.cuda.device(dev):
            st = _stream(dev)
            # Throughput mode (``overlap_encoder``, sync-free calls only): the token-level half of the forward -- encoder, duration predictor:
            # ~30 launches that leave most of the chip idle -- runs on a side stream, so that call i + 1's encoder executes while call i's
            # frame-level kernels still run on the caller's stream; an event hands the encoder's results to this call's fs2_decode.  The side
            # stream does NOT wait for the caller's stream: the caller promises that xs (and d_override) are complete (see ``overlap_encoder``).
            cur = torch.cuda.current_stream(dev)
            side = None
            if capacity is not None and self.overlap_encoder and not torch.cuda.is_current_stream_capturing():
                side = self.input_stream(dev)
                if self._enc_gen_seen.get(side.cuda_stream) != self._weights_generation:      # the weights were (re-)uploaded on some stream since this side stream last looked
                    torch.cuda.synchronize(dev)
                    self._enc_gen_seen[side.cuda_stream] = self._weights_generation
            if side is not None and inputs_ready is not None:
                side.wait_event(inputs_ready)          # the caller's marker behind whatever produces xs / d_override (otherwise: its promise that they are complete)
            with torch.cuda.stream(side if side is not None else cur):
                st_e = _stream(dev)
                tok_ws = torch.empty(L.fs2_token_workspace_bytes(h, C.byref(batch)), dtype=torch.uint8, device=dev)
                d_log = torch.empty(B, Tmax, dtype=torch.float32, device=dev) if teacher else None
                d_int = torch.empty(B, Tmax, dtype=torch.int64, device=dev) if is_inference else None
                olens_dev = torch.empty(B, dtype=torch.int64, device=dev)
                enc_out = torch.empty(B, Tmax, self._cfg["adim"], device=dev) if "encoder_out" in want else None
                eio = _lib.EncodeIO(batch, xs.data_ptr(), ds_dev.data_ptr() if ds_dev is not None else None,
                                    d_log.data_ptr() if d_log is not None else None,
                                    d_int.data_ptr() if d_int is not None else None, olens_dev.data_ptr(),
                                    enc_out.data_ptr() if enc_out is not None else None, tok_ws.data_ptr(), tok_ws.numel(), float(alpha))
                _lib.check(L.fs2_encode(h, st_e, C.byref(eio)), h)
            if side is not None:
                cur.wait_stream(side)                      # (this call's fs2_decode and everything behind it on the caller's stream)
                for t_ in (tok_ws, d_log, d_int, olens_dev, enc_out):      # allocated on the side stream, used on the caller's
                    if t_ is not None:
                        t_.record_stream(cur)
                xs.record_stream(side)                                      # ... and the other way round
                if ds_dev is not None:
                    ds_dev.record_stream(side)
            if capacity is not None:
                total_cap, Lcap = int(capacity[0]), int(capacity[1])
                self.decoder.embed[-1].ensure(Lcap)      # (a grown table is picked up by the next call's fingerprint check)
                rows = int(L.fs2_row_capacity(C.byref(batch), total_cap))
                frm_ws = torch.empty(L.fs2_frame_workspace_bytes_cap(h, C.byref(batch), rows, Lcap), dtype=torch.uint8, device=dev)
                status = torch.empty(8, dtype=torch.int32, device=dev)
                odim = self.odim

                def cbuf(key, shape, dtype=torch.float32):
                    if key in want:
                        out[key] = torch.empty(shape, dtype=dtype, device=dev)
                        return out[key].data_ptr()
                    return None

                if teacher:
                    raise ValueError("the device-driven layout serves free-running synthesis")
                dio = _lib.DecodeIO(
                    batch, None, Lcap, 0, None, None, 0, 0,
                    cbuf("before", (B, Lcap * self.reduction_factor, odim)), cbuf("after", (B, Lcap * self.reduction_factor, odim)),
                    cbuf("e_outs", (B, Lcap)), cbuf("p_outs", (B, Lcap)),
                    cbuf("qe", (B, Lcap), torch.int32), cbuf("qp", (B, Lcap), torch.int32),
                    cbuf("lr_index", (B, Lcap), torch.int32), cbuf("decoder_out", (B, Lcap, self._cfg["ddim"])),
                    tok_ws.data_ptr(), frm_ws.data_ptr(), frm_ws.numel(), None, rows, status.data_ptr())
                if "after_packed" in want:
                    prows = rows * self.reduction_factor       # mel frames: r per decoder row
                    if packed_out is not None:
                        if (not packed_out.is_contiguous() or packed_out.dtype != torch.float32 or packed_out.device != dev
                                or packed_out.dim() != 2 or packed_out.shape[1] != odim or packed_out.shape[0] < prows):
                            raise ValueError("packed_out must be a contiguous float32 [>= %d, %d] tensor on %s" % (prows, odim, dev))
                        out["after_packed"] = packed_out[:prows]
                    else:
                        out["after_packed"] = torch.empty((prows, odim), dtype=torch.float32, device=dev)
                    dio.after_packed = out["after_packed"].data_ptr()
                if dio.after is None and dio.after_packed is None:
                    raise ValueError("'after' or 'after_packed' must be requested")
                _lib.check(L.fs2_decode(h, st, C.byref(dio)), h)
                out["olens"], out["status"] = olens_dev, status
                if d_int is not None:
                    out["d_int"] = d_int if d_override is None else ds_dev
                if enc_out is not None:
                    out["encoder_out"] = enc_out
                return out
            ol = olens_dev.cpu()                     # the one host sync of the path (frame counts)
            if int(ol.min()) < 0:                    # fs2_encode's marker (include/fs2.h: olens = -1)
                bad = [i for i, v in enumerate(ol.tolist()) if v < 0]
                raise Fs2IndexError("index out of range in self: utterance(s) %s hold a phoneme id outside [0, %d)" % (bad, self.idim))
            self.last_olens = ol
            if olens is not None:
                given = torch.as_tensor(olens).detach().to("cpu", torch.int64)
                if not torch.equal(given, ol):
                    raise ValueError("olens %s do not match the sums of the durations %s" % (given.tolist(), ol.tolist()))
            Lmax = int(ol.max())
            if self.decoder.embed[-1].ensure(Lmax):                # pe table grew: reload and redo the encoder
                return self._run(xs, ilens, olens, ds, es, ps, is_inference, compat, want, d_override, alpha=alpha, regime=regime)
            ol_arr = (C.c_int64 * B)(*ol.tolist())
            frm_ws = torch.empty(L.fs2_frame_workspace_bytes(h, C.byref(batch), ol_arr), dtype=torch.uint8, device=dev)
            odim = self.odim

            def buf(key, shape, dtype=torch.float32):
                if key in want:
                    out[key] = torch.empty(shape, dtype=dtype, device=dev)
                    return out[key].data_ptr()
                return None

            es_dev = es.to(dev).float().contiguous() if teacher else None
            ps_dev = ps.to(dev).float().contiguous() if teacher else None
            masked = int(teacher and olens is not None) if compat else 0
            dio = _lib.DecodeIO(
                batch, ol_arr, Lmax, masked,
                es_dev.data_ptr() if es_dev is not None else None, ps_dev.data_ptr() if ps_dev is not None else None,
                es_dev.shape[1] if es_dev is not None else 0, ps_dev.shape[1] if ps_dev is not None else 0,
                buf("before", (B, Lmax * self.reduction_factor, odim)), buf("after", (B, Lmax * self.reduction_factor, odim)),
                buf("e_outs", (B, Lmax)), buf("p_outs", (B, Lmax)),
                buf("qe", (B, Lmax), torch.int32), buf("qp", (B, Lmax), torch.int32),
                buf("lr_index", (B, Lmax), torch.int32), buf("decoder_out", (B, Lmax, self._cfg["ddim"])),
                tok_ws.data_ptr(), frm_ws.data_ptr(), frm_ws.numel(),
                buf("after_packed", (int(ol.sum()) * self.reduction_factor, odim)))
            if dio.after is None:
                raise ValueError("'after' must be requested")
            _lib.check(L.fs2_decode(h, st, C.byref(dio)), h)
        out["olens"] = ol
        if d_log is not None:
            out["d_log"] = d_log
        if d_int is not None:
            out["d_int"] = d_int if d_override is None else ds_dev
        if enc_out is not None:
            out["encoder_out"] = enc_out
        return out

    def _forward(self, xs, ilens, olens=None, ds=None, es=None, ps=None, is_inference=False):
        """reference fastspeech.py:169-243: returns (before, after, d_outs, e, p).

        inference: d_outs int64 durations, e/p the one-hot energy/pitch codes [B, Lmax, 256];
        otherwise: d_outs log-durations (pads 0), e/p the predictor outputs (pads 0)."""
        compat = self.batch_semantics == "padded_compat"
        if is_inference:
            r = self._run(xs, ilens, is_inference=True, compat=compat, want=("before", "after", "qe", "qp"))
            one_hot = lambda q: F.one_hot(q.clamp(min=0).long(), 256).float() * (q >= 0).unsqueeze(-1)
            return r["before"], r["after"], r["d_int"], one_hot(r["qe"]), one_hot(r["qp"])
        r = self._run(xs, ilens, olens, ds, es, ps, is_inference=False, compat=compat,
                      want=("before", "after", "e_outs", "p_outs"))
        return r["before"], r["after"], r["d_log"], r["e_outs"], r["p_outs"]

    def forward(self, xs, ilens, ys, olens, ds, es, ps):
        """reference fastspeech.py:245-337: losses of the teacher-forced pass; returns (loss, report_keys).

        The forward pass runs on the device (padded_compat semantics, as the reference computes it); the
        loss algebra on its outputs is plain PyTorch, as in the reference."""
        _require_device(xs)
        dev = xs.device
        il = torch.as_tensor(ilens).to("cpu", torch.int64)
        if self.reduction_factor > 1:
            raise NotImplementedError("loss path with reduction_factor > 1 (the reference's own length handling for it is commented "
                                      "out, fastspeech.py:275-276); _forward / inference are implemented")
        ol = torch.as_tensor(olens).to("cpu", torch.int64)
        xs = xs[:, : int(il.max())]
        ys = ys[:, : int(ol.max())].to(dev)
        ds, es, ps = ds.to(dev), es.to(dev).float(), ps.to(dev).float()
        ds_t = ds[:, : xs.shape[1]]
        r = self._run(xs, il, ol, ds_t, es, ps, is_inference=False, compat=True,
                      want=("before", "after", "e_outs", "p_outs"))
        before, after, d_outs, e_outs, p_outs = r["before"], r["after"], r["d_log"], r["e_outs"], r["p_outs"]
        ar_t = torch.arange(xs.shape[1], device=dev).unsqueeze(0)
        ar_l = torch.arange(before.shape[1], device=dev).unsqueeze(0)
        in_masks = ar_t < il.to(dev).unsqueeze(1)
        mel_masks = ar_l < ol.to(dev).unsqueeze(1)
        out_masks = mel_masks.unsqueeze(-1)
        es, ps = es[:, : before.shape[1]], ps[:, : before.shape[1]]
        if self.use_masking:
            d_outs, ds_t = d_outs.masked_select(in_masks), ds_t.masked_select(in_masks)
            before, after = before.masked_select(out_masks), after.masked_select(out_masks)
            es, ps = es.masked_select(mel_masks), ps.masked_select(mel_masks)
            e_outs, p_outs = e_outs.masked_select(mel_masks), p_outs.masked_select(mel_masks)
            ys = ys.masked_select(out_masks)
        before_loss = F.l1_loss(before, ys)
        after_loss = F.l1_loss(after, ys)
        l1_loss = before_loss + after_loss
        duration_loss = F.mse_loss(d_outs, torch.log(ds_t.float() + 1.0))
        energy_loss = F.mse_loss(e_outs, es)
        pitch_loss = F.mse_loss(p_outs, ps)
        if self.use_weighted_masking:
            # reference fastspeech.py:308-325, quirks included: the weights multiply the already mean-reduced scalars, and they are built
            # from ys.size(2) -- with use_masking as well, ys is 1-D by now and this raises IndexError exactly as the reference does
            out_weights = out_masks.float() / out_masks.sum(dim=1, keepdim=True).float()
            out_weights = out_weights / (ys.size(0) * ys.size(2))
            duration_weights = in_masks.float() / in_masks.sum(dim=1, keepdim=True).float()
            duration_weights = duration_weights / ds_t.size(0)
            l1_loss = l1_loss.mul(out_weights).masked_select(out_masks).sum()
            duration_loss = duration_loss.mul(duration_weights).masked_select(in_masks).sum()
        loss = l1_loss + duration_loss + energy_loss + pitch_loss
        report_keys = [{"l1_loss": l1_loss.item()}, {"before_loss": before_loss.item()}, {"after_loss": after_loss.item()},
                       {"duration_loss": duration_loss.item()}, {"energy_loss": energy_loss.item()},
                       {"pitch_loss": pitch_loss.item()}, {"loss": loss.item()}]
        return loss, report_keys

    def inference(self, x, alpha=1.0):
        """reference fastspeech.py:339-357: x [T] int64 phoneme ids -> mel [L, odim].  ``alpha`` (not in the reference's
        ``inference``, but in its LengthRegulator, length_regulator.py:57-59) scales the durations: > 1 slower speech."""
        xs, il = x.unsqueeze(0), torch.tensor([x.shape[0]])
        if not float(alpha) > 0.0:
            raise ValueError("alpha must be > 0 (reference length_regulator.py:57), got %r" % (alpha,))
        if self._frames_per_token is not None:
            # device-driven layout inside capacities learnt from earlier utterances: the GPU runs the whole forward without
            # waiting for the host; the frame count is read once, at the end (it is needed for the shape of the result)
            total, Lcap = self.predict_capacity(il, alpha)
            r = self._run(xs, il, is_inference=True, want=("after",), capacity=(max(total, Lcap), Lcap), alpha=alpha)
            st = r["status"].cpu()
            if int(st[2]) == 0:
                L = int(st[3])
                self._learn_ratio(il, torch.tensor([L]), alpha)
                return r["after"][0, :L * self.reduction_factor]
        r = self._run(xs, il, is_inference=True, want=("after",), alpha=alpha)
        self._learn_ratio(il, r["olens"], alpha)
        return r["after"][0]

    def inference_batch(self, xs, ilens, d_override=None, packed=False, sync=True, capacity=None, alpha=1.0, packed_out=None, regime=None,
                        inputs_ready=None):
        """Batched free-running synthesis (not in the reference, which only has single-utterance
        ``inference``): per-utterance semantics; returns (mels [B, Lmax, odim], olens [B] on the host = MEL frames per utterance,
        i.e. decoder frames x reduction_factor), or with
        ``packed=True`` (valid frames back to back [sum(olens), odim], olens): the form the multi-GPU gather ships.
        An empty batch (B = 0, e.g. a rank that got no utterance) returns empty tensors without touching the GPU.

        ``sync=False``: nothing waits for the GPU.  The frame layout is built on the device inside capacities predicted
        from earlier calls (frames per phoneme seen so far, with headroom; or ``capacity=(total_frames, per_utterance)``
        given by the caller, e.g. agreed between ranks) and the call returns an :class:`AsyncMels`, which unpacks like
        (mels [B, Lcap, odim] zero-padded - or, with ``packed=True``, [rows_cap, odim] whose first sum(olens) rows are the
        valid frames - , olens as a DEVICE int64 tensor) and carries ``status`` / ``ok()`` / ``check()`` for THIS call.
        If a capacity did not suffice the mels of that call are NaN-filled on the device (never silently wrong);
        ``ok()`` is then False: repeat the batch with ``sync=True`` (which learns the exact sizes).  ``packed_out``: a
        caller-owned [>= row capacity, odim] buffer that receives the pack in place (the multi-GPU send buffer).  Any number of
        asynchronous calls may be in flight; ``async_ok()`` waits for all of them and tells whether every one since the last
        ``async_ok()`` was valid.  The first call of a model is always synchronous.  ``alpha``: duration scale.

        ``regime=(phonemes, utterances)``: this batch is a SHARD of a batch of that size -- choose the kernel variants (which differ in
        summation order, i.e. in the last bits) as the one-call run of the whole batch would, so that every utterance comes out bit-identical
        to that run (what ``ShardedSynthesizer`` passes on every rank; include/fs2.h: fs2_batch.regime_tokens / regime_utterances).

        ``inputs_ready`` (a ``torch.cuda.Event``; ``overlap_encoder`` mode only, ignored otherwise): recorded by the caller behind the work that
        produces ``xs`` / ``d_override``; the encoder's side stream waits for it.  Without it the mode relies on the caller's promise that the
        inputs are complete when the call is made (the side stream does not wait for the caller's stream: that wait is the overlap)."""
        il = torch.as_tensor(ilens).detach().to("cpu", torch.int64).reshape(-1)
        if not float(alpha) > 0.0:
            raise ValueError("alpha must be > 0 (reference length_regulator.py:57), got %r" % (alpha,))
        if il.numel() == 0:
            _require_device(xs)
            empty = torch.zeros((0, self.odim) if packed else (0, 1, self.odim), device=xs.device)
            return empty, (torch.zeros(0, dtype=torch.int64) if sync else torch.zeros(0, dtype=torch.int64, device=xs.device))
        if not sync and (self._frames_per_token is not None or capacity is not None):
            self._harvest_async(block=False)
            total, Lcap = capacity if capacity is not None else self.predict_capacity(il, alpha)
            key = "after_packed" if packed else "after"         # (packed: the padded mels are neither built nor written)
            r = self._run(xs, il, is_inference=True, compat=False, want=(key,), d_override=d_override, capacity=(total, Lcap), alpha=alpha,
                          packed_out=packed_out if packed else None, regime=regime, inputs_ready=inputs_ready)
            mel_lens = r["olens"] if self.reduction_factor == 1 else r["olens"] * self.reduction_factor     # mel frames per utterance
            if torch.cuda.is_current_stream_capturing():     # graph capture: no host-side bookkeeping inside the graph
                return AsyncMels(r[key], mel_lens, r["status"], None)
            return AsyncMels(r[key], mel_lens, r["status"], self._record_async(il, r, xs.device, alpha))
        want = ("after", "after_packed") if packed else ("after",)
        r = self._run(xs, il, is_inference=True, compat=False, want=want, d_override=d_override, alpha=alpha, regime=regime)
        if d_override is None:
            self._learn_ratio(il, r["olens"], alpha)
        else:
            self._learn_ratio(il, r["olens"], 1.0)
        return (r["after_packed"] if packed else r["after"]), r["olens"] * self.reduction_factor

    def capture_graph(self, xs, ilens, d_override=None):
        """HIP-graph replay of the whole free-running forward for a fixed batch shape (``xs.shape`` and ``ilens``): the
        launch-bound small-batch case (one utterance: 85 launches) becomes one graph launch.  Returns ``run(new_xs) ->
        (mels [B, Lcap, odim], olens_dev, status_dev)``; the tensors are the graph's static outputs (overwritten by the
        next replay).  ``status_dev[2] != 0`` means the capacities captured with the graph were too small for that input
        (the mels are then NaN): fall back to ``inference_batch``.  Capacities come from one synchronous run on ``xs``
        (x 1.5 head-room).  The graph holds pointers into the library's weight copies: ``run`` raises if the weights were
        re-uploaded since the capture (load_state_dict, a grown positional table, .to()): capture again."""
        with torch.no_grad():
            il = torch.as_tensor(ilens).detach().to("cpu", torch.int64)
            _, ol = self.inference_batch(xs, il, d_override=d_override)            # builds the handle, learns the sizes
            total = int(float(ol.sum()) * 1.5) + 64 * int(il.numel())
            Lcap = -(-int(float(ol.max()) * 1.5 + 64) // 32) * 32
            self.decoder.embed[-1].ensure(Lcap)
            self._ensure_ready(xs.device, xs.shape[1], Lcap)                        # a grown table is uploaded BEFORE the capture
            static_xs = xs.clone()
            static_ds = d_override.clone() if d_override is not None else None
            run_once = lambda: self._run(static_xs, il, is_inference=True, compat=False, want=("after",), d_override=static_ds,
                                         capacity=(total, Lcap))
            side = torch.cuda.Stream(device=xs.device)
            side.wait_stream(torch.cuda.current_stream(xs.device))
            with torch.cuda.stream(side):
                run_once()                                                         # warm-up on the capture stream
            torch.cuda.current_stream(xs.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            # "relaxed": the library pins a small host staging block for the token layout while capturing (hipHostMalloc)
            with torch.cuda.graph(graph, capture_error_mode="relaxed"):
                r = run_once()
        after, olens_dev, status = r["after"], r["olens"], r["status"]
        generation, handle = self._weights_generation, self._handle

        def run(new_xs, new_ds=None):
            if self._handle is not handle or self._weights_generation != generation or self._weights_fingerprint() != self._fingerprint:
                raise RuntimeError("the model's weights changed since capture_graph(): the graph points at released device "
                                   "copies; capture a new graph")
            static_xs.copy_(new_xs)
            if static_ds is not None and new_ds is not None:
                static_ds.copy_(new_ds)
            graph.replay()
            return after, olens_dev, status

        run.graph = graph
        return run

    # frames-per-phoneme statistics for the capacities of the asynchronous path: (batch mean, max over utterances)
    _frames_per_token = None
    _overflow_seen = False
    last_async = None           # the most recent _AsyncRecord (kept for introspection)

    def predict_capacity(self, ilens, alpha=1.0):
        """(total frames, frames of the longest utterance) to reserve for a batch with these phoneme counts: the
        frames-per-phoneme ratios seen so far (x the duration scale ``alpha``) plus 15 % / 25 % head-room."""
        il = torch.as_tensor(ilens).detach().to("cpu", torch.int64)
        a = float(alpha)
        mean_r, max_r = self._frames_per_token[0] * a + (0.5 if a != 1.0 else 0.0), self._frames_per_token[1] * a + (0.5 if a != 1.0 else 0.0)
        total = int(float(il.sum()) * mean_r * 1.15) + 64 * int(il.numel())
        Lcap = -(-int(float(il.max()) * max_r * 1.25 + 64) // 32) * 32
        return total, Lcap

    def _learn_ratio(self, il, ol, alpha=1.0):
        a = max(float(alpha), 1e-6)
        mean_r = float(ol.sum()) / max(float(il.sum()), 1.0) / a
        max_r = float((ol.float() / il.float().clamp(min=1)).max()) / a
        old = self._frames_per_token
        self._frames_per_token = (mean_r, max_r) if old is None else (max(mean_r, 0.9 * old[0]), max(max_r, 0.9 * old[1]))

    _PIN_SLOTS = 16

    def _record_async(self, il, r, device, alpha):
        """Queue the frame counts and flags of an asynchronous call: they travel to pinned host memory behind the kernels, an
        event tells when they are there.  A ring of pinned slots serves any number of calls in flight (a slot still in use
        when the ring wraps is waited for and folded first)."""
        B = int(il.numel())
        if len(self._pin_ring) < self._PIN_SLOTS:
            self._pin_ring.append([torch.empty(max(B, 64), dtype=torch.int64).pin_memory(), torch.empty(8, dtype=torch.int32).pin_memory(), None])
            slot = self._pin_ring[-1]
        else:
            slot = self._pin_ring[self._pin_next]
            self._pin_next = (self._pin_next + 1) % self._PIN_SLOTS
            if slot[2] is not None and not slot[2].harvested:
                self._fold(slot[2], block=True)
            if slot[0].numel() < B:
                slot[0] = torch.empty(B, dtype=torch.int64).pin_memory()
        slot[0][:B].copy_(r["olens"], non_blocking=True)
        slot[1].copy_(r["status"], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        rec = _AsyncRecord((il, alpha), ev, slot[0], slot[1])
        slot[2] = rec
        self._pending.append(rec)
        self.last_async = rec
        return rec

    def _fold(self, rec, block):
        """Fold one asynchronous call into the capacity predictor.  Returns its flags, or None if it has not finished."""
        flags = rec.flags(block)
        if flags is None:
            return None
        if not rec.harvested:
            rec.harvested = True
            il, alpha = rec.il
            if flags == 0:
                self._learn_ratio(il, rec.olens, alpha)
            else:
                self._overflow_seen = True
            if rec in self._pending:
                self._pending.remove(rec)
        return flags

    def _harvest_async(self, block):
        """Fold every finished asynchronous call (all of them if ``block``) into the capacity predictor."""
        for rec in list(self._pending):
            if self._fold(rec, block) is None:
                break

    def input_stream(self, device):
        """The stream a sync-free call's inputs should be prepared on: the encoder's side stream in ``overlap_encoder`` mode (so that slicing /
        gathering the ids of call i + 1 does not queue behind call i's frame-level kernels), else the current stream."""
        cur = torch.cuda.current_stream(device)
        if not self.overlap_encoder or torch.cuda.is_current_stream_capturing():
            return cur
        key = (device, cur.cuda_stream)          # one side stream per stream the caller works on: steps issued on alternating streams overlap their encoders too
        side = self._enc_streams.get(key)
        if side is None:
            side = self._enc_streams[key] = torch.cuda.Stream(device=device)
        return side

    def async_ok(self):
        """Waits for every ``inference_batch(sync=False)`` call still in flight; True if the capacities of ALL asynchronous
        calls since the previous ``async_ok()`` sufficed (an overflowed call returned NaN-filled mels)."""
        self._harvest_async(block=True)
        ok = not self._overflow_seen
        self._overflow_seen = False
        return ok

    def _source_mask(self, ilens):
        """reference fastspeech.py:359-376 (kept for API parity; the kernels take lengths, not masks)."""
        il = torch.as_tensor(ilens)
        m = torch.arange(int(il.max()), device=il.device).unsqueeze(0) < il.unsqueeze(1)
        m = m.to(self.feat_out.weight.device)
        return m.unsqueeze(-2) & m.unsqueeze(-1)


def _profile_methods():
    def set_profiling(self, on=True, only=None):
        """hipEvent timing of kernel launches on the caller's stream (fs2_set_profiling); ``only`` = name of the
        single launch site to bracket (keeps the event overhead out of a timed region)."""
        if self._handle is None:
            raise RuntimeError("run one forward before enabling profiling (the handle is created lazily)")
        _lib.check(_lib.lib().fs2_set_profile_filter(self._handle, only.encode() if only else None), self._handle)
        _lib.check(_lib.lib().fs2_set_profiling(self._handle, int(bool(on))), self._handle)

    def get_profile(self, cap=65536):
        """-> list of (name, ms, flops, bytes) per launch since set_profiling(True)."""
        L = _lib.lib()
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        fl = (C.c_double * cap)()
        by = (C.c_double * cap)()
        n = L.fs2_get_profile(self._handle, names, ms, fl, by, cap)
        return [(names[i].decode(), float(ms[i]), float(fl[i]), float(by[i])) for i in range(n)]

    def counter(self, name, reset=False):
        """Cumulative event counter of this model's device handle (include/fs2.h: fs2_get_counter; waits for the current stream):
        "attn_slow_path_waves" = waves of attn_w32 that left the fast path and were recomputed by the plain fp32 loop."""
        if self._handle is None:
            return 0
        v = C.c_int64(0)
        with torch.cuda.device(self._handle_device):
            _lib.check(_lib.lib().fs2_get_counter(self._handle, _stream(self._handle_device), name.encode(), C.byref(v), int(bool(reset))), self._handle)
        return int(v.value)

    FeedForwardTransformer.set_profiling = set_profiling
    FeedForwardTransformer.get_profile = get_profile
    FeedForwardTransformer.counter = counter


_profile_methods()
