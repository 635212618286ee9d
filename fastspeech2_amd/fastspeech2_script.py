"""TorchScript-able inference twin, counterpart of reference utils/fastspeech2_script.py.

The reference keeps a second, inference-only ``FeedForwardTransformer`` for ``export_torchscript.py``
(reference export_torchscript.py:46-58): *different architecture* from fastspeech.py -- the decoder runs at
``adim`` with a positional-encoding-only input layer and ``feat_out`` is ``Linear(adim, odim)``
(utils/fastspeech2_script.py:112-127,145) -- and ``forward(x: [T] int64) -> [L, odim]``.

Here the same HIP library runs it (``fs2_config.decoder_input_layer = 0``).  To be scriptable the forward
cannot go through ctypes, so it is a dispatcher op with a schema,

    fs2::twin_inference(Tensor x, Tensor flat_weights, str config_json) -> Tensor

implemented in C++ (``csrc/fs2_torch_op.cpp`` -> ``libfs2_torch.so``, built by ``__graft_entry__.build()``): it cuts the flat
weight buffer the scripted module carries by the manifest inside ``config_json``, keeps a libfs2_hip handle per weight buffer
(an LRU of four) and runs fs2_encode / fs2_decode on torch's current stream.  ``torch.jit.script(model)``, ``torch.jit.trace``
and ``torch.jit.save/load`` work, and a saved archive runs in ANY process that has loaded the op library -- Python or C++,
without this package:

    torch.ops.load_library("<repo>/fastspeech2_amd/libfs2_torch.so")
    mel = torch.jit.load("fs2_twin.pt")(ids)
"""
import json
import os

import torch

from . import _lib
from .fastspeech import FeedForwardTransformer as _Base

__all__ = ["FeedForwardTransformer", "OP_LIBRARY"]

OP_LIBRARY = _lib.TORCH_OP_PATH


def _load_op_library():
    """Load libfs2_torch.so (registers fs2::twin_inference).  There is no Python implementation to fall back to."""
    if hasattr(torch.ops, "fs2") and hasattr(torch.ops.fs2, "twin_inference"):
        try:
            torch.ops.fs2.twin_inference.default      # already registered (library loaded earlier in this process)
            return
        except (AttributeError, RuntimeError):
            pass
    if not os.path.exists(OP_LIBRARY):
        raise _lib.Fs2LibraryError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` (the TorchScript twin's op is "
            "implemented in C++; there is no Python fallback)" % OP_LIBRARY)
    torch.ops.load_library(OP_LIBRARY)


_load_op_library()


def _float_keys(module):
    return [k for k, v in sorted(module.state_dict().items()) if v.dtype == torch.float32 and k != "flat_weights"]


class FeedForwardTransformer(_Base):
    """Same ctor as reference utils/fastspeech2_script.py:29; ``forward(x)`` as :201-219; state_dict keys as the
    reference twin's (``decoder.embed.0.{alpha,pe}``, ``feat_out.weight [odim, adim]``)."""

    __jit_unused_properties__ = ["precision"]

    @property
    def precision(self):
        """The arithmetic mode (as on the base class).  On the twin it is frozen into ``config_json`` when the weights are packed, so
        setting it repacks: `twin.precision = "mix_mx"` followed by scripting / saving exports what was asked for, not fp32."""
        return self.__dict__.get("_precision", "fp32")

    @precision.setter
    def precision(self, value):
        self.__dict__["_precision"] = value
        if self.__dict__.get("config_json"):      # (not yet during __init__: the first pack_weights() comes at its end)
            self.pack_weights()

    def __init__(self, idim: int, odim: int, hp):
        super().__init__(idim, odim, hp, _script_twin=True)
        self._hp_plain = {"model": dict(hp.model), "data": {k: hp.data[k] for k in ("e_min", "e_max", "p_min", "p_max")}}
        self.config_json = ""
        self.register_buffer("flat_weights", torch.zeros(0), persistent=False)
        self.pack_weights()

    def pack_weights(self):
        """(Re)build what the scripted forward ships to the op -- the flat fp32 weight buffer and ``config_json`` (hyper-parameters,
        arithmetic mode, and the manifest [[name, shape], ...] that says how to cut the buffer); call after changing parameters or
        ``precision`` by hand (``load_state_dict`` does it for you)."""
        sd = self.state_dict()
        keys = _float_keys(self)
        self.flat_weights = torch.cat([sd[k].detach().reshape(-1).float() for k in keys]).to(self.feat_out.weight.device)
        self.config_json = json.dumps({"idim": self.idim, "odim": self.odim, "hp": self._hp_plain, "precision": self.precision,
                                       "tensors": [[k, list(sd[k].shape)] for k in keys]}, sort_keys=True)

    def load_state_dict(self, state_dict, strict: bool = True):
        r = super().load_state_dict(state_dict, strict)
        self.pack_weights()
        return r

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.ops.fs2.twin_inference(x, self.flat_weights, self.config_json)
