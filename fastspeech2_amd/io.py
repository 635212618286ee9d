"""Callers' side of the path: checkpoint loading and the vocoder hand-off.

* ``load_checkpoint`` mirrors what the reference's scripts do around the model (inference.py:149-166,
  train_fastspeech.py:42-64,235-244): checkpoints are ``{"model": state_dict, "optim", "step", "hp_str", "githash"}``;
  ``--old_model`` checkpoints are bare state dicts loaded with ``strict=False``.
* ``vocoder_input`` is the step right after the path (inference.py:173-193, utils/plot.py:96-105): the per-sentence mels
  ``[L_i, 80]`` are transposed and concatenated to ``[1, 80, sum L_i]`` for MelGAN -- here one HIP kernel over the packed
  frames, no host round trip.
"""
import ctypes as C
import io
import os

import torch
import yaml

from . import _lib
from .hparams import DotDict


def hparams_from_str(hp_str):
    """The YAML text the reference embeds in checkpoints (train_fastspeech.py:417-418,240; utils/hparams.py:5-11)."""
    merged = {}
    for doc in yaml.safe_load_all(io.StringIO(hp_str)):
        merged.update(doc or {})
    return DotDict(merged)


def load_checkpoint(model, checkpoint, old_model=False, map_location="cpu", trust_checkpoint=False):
    """Load reference weights into ``model``.  ``checkpoint``: a path or an already loaded object.
    Returns the dict of extras found (``step``, ``hp_str``, ``githash``) -- empty for bare state dicts.

    Files are read with ``torch.load(weights_only=True)``: the reference's checkpoints hold only tensors, str and int fields
    (``model``, ``optim``, ``step``, ``hp_str``, ``githash``), so nothing else needs unpickling and a downloaded file cannot
    run code.  ``trust_checkpoint=True`` opts into full unpickling for legacy files that carry other objects."""
    obj = torch.load(checkpoint, map_location=map_location, weights_only=not trust_checkpoint) if isinstance(checkpoint, (str, os.PathLike)) else checkpoint
    extras = {}
    if isinstance(obj, dict) and "model" in obj and isinstance(obj["model"], dict):
        sd = obj["model"]
        extras = {k: obj[k] for k in ("step", "hp_str", "githash") if k in obj}
    else:
        sd = obj
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}     # utils/util.py:533 (DataParallel prefix)
    missing = model.load_state_dict(sd, strict=not old_model)
    if old_model and (missing.missing_keys or missing.unexpected_keys):
        extras["missing_keys"], extras["unexpected_keys"] = list(missing.missing_keys), list(missing.unexpected_keys)
    return extras


def vocoder_input(packed_mels):
    """packed mel frames [N, 80] (``inference_batch(..., packed=True)`` or ``torch.cat`` of per-sentence mels) ->
    [1, 80, N] on the same device (MelGAN's input layout)."""
    if not packed_mels.is_cuda:
        raise RuntimeError("fastspeech2_amd runs on an MI355X only (no CPU fallback)")
    x = packed_mels.contiguous().float()
    n, w = x.shape
    out = torch.empty(1, w, n, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().fs2_op_transpose(C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), x.data_ptr(), n, w, out.data_ptr()))
    return out
