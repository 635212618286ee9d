"""TorchScript-able inference twin, counterpart of reference utils/fastspeech2_script.py.

The reference keeps a second, inference-only ``FeedForwardTransformer`` for ``export_torchscript.py``
(reference export_torchscript.py:46-58): *different architecture* from fastspeech.py -- the decoder runs at
``adim`` with a positional-encoding-only input layer and ``feat_out`` is ``Linear(adim, odim)``
(utils/fastspeech2_script.py:112-127,145) -- and ``forward(x: [T] int64) -> [L, odim]``.

Here the same HIP library runs it (``fs2_config.decoder_input_layer = 0``).  To be scriptable the forward
cannot go through ctypes, so it is a registered dispatcher op with a schema,

    fs2::twin_inference(Tensor x, Tensor flat_weights, str config_json) -> Tensor

whose implementation rebuilds (and caches) the eager module from the flat weight buffer + JSON config that
the scripted module carries.  ``torch.jit.script(model)``, ``torch.jit.trace`` and ``torch.jit.save/load``
therefore work; a reloaded archive runs in any process that has imported this package (which registers the op).
"""
import json

import torch

from .fastspeech import FeedForwardTransformer as _Base
from .hparams import DotDict

__all__ = ["FeedForwardTransformer"]

_LIBDEF = torch.library.Library("fs2", "DEF")
_LIBDEF.define("twin_inference(Tensor x, Tensor flat_weights, str config_json) -> Tensor")

_CACHE = {}


def _float_keys(module):
    return [k for k, v in sorted(module.state_dict().items()) if v.dtype == torch.float32]


def _twin_inference(x, flat_weights, config_json):
    # The cache entry keeps a reference to the flat buffer it was built from: its storage therefore cannot be released and handed
    # to another tensor while the entry lives, so (data_ptr, version) identifies the weights without reading them back.
    key = (flat_weights.data_ptr(), flat_weights._version, flat_weights.numel(), flat_weights.device, config_json)
    entry = _CACHE.get(key)
    inner = entry[0] if entry is not None else None
    if inner is None:
        cfg = json.loads(config_json)
        inner = _Base(cfg["idim"], cfg["odim"], DotDict(cfg["hp"]), _script_twin=True)
        sd, off = inner.state_dict(), 0
        flat = flat_weights.detach()
        for k in _float_keys(inner):
            n = sd[k].numel()
            sd[k] = flat[off:off + n].view(sd[k].shape).clone()
            off += n
        if off != flat.numel():
            raise RuntimeError("flat weight buffer has %d elements, the architecture needs %d" % (flat.numel(), off))
        inner.load_state_dict(sd)
        inner = inner.to(flat_weights.device).eval()
        _CACHE.clear()          # one live model per process is the export use case
        _CACHE[key] = (inner, flat_weights)
    with torch.no_grad():
        return inner.inference(x)


torch.library.impl(_LIBDEF, "twin_inference", "CompositeExplicitAutograd")(_twin_inference)


class FeedForwardTransformer(_Base):
    """Same ctor as reference utils/fastspeech2_script.py:29; ``forward(x)`` as :201-219; state_dict keys as the
    reference twin's (``decoder.embed.0.{alpha,pe}``, ``feat_out.weight [odim, adim]``)."""

    def __init__(self, idim: int, odim: int, hp):
        super().__init__(idim, odim, hp, _script_twin=True)
        hp_plain = {"model": dict(hp.model), "data": {k: hp.data[k] for k in ("e_min", "e_max", "p_min", "p_max")}}
        self.config_json = json.dumps({"idim": idim, "odim": odim, "hp": hp_plain}, sort_keys=True)
        self.register_buffer("flat_weights", torch.zeros(0), persistent=False)
        self.pack_weights()

    def pack_weights(self):
        """(Re)build the flat fp32 buffer the scripted forward ships to the op; call after changing parameters
        by hand (``load_state_dict`` does it for you)."""
        sd = self.state_dict()
        keys = [k for k in _float_keys(self) if k != "flat_weights"]
        self.flat_weights = torch.cat([sd[k].detach().reshape(-1).float() for k in keys]).to(self.feat_out.weight.device)

    def load_state_dict(self, state_dict, strict: bool = True):
        r = super().load_state_dict(state_dict, strict)
        self.pack_weights()
        return r

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.ops.fs2.twin_inference(x, self.flat_weights, self.config_json)
