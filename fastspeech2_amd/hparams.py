"""YAML -> dot-dict hyper-parameters.

The reference passes an ``HParam`` dot-dict (reference utils/hparams.py:55-65) into
``FeedForwardTransformer(idim, odim, hp)``.  The module in this package consumes any
object with ``hp.model.<field>`` / ``hp.data.<field>`` attribute access, so the
reference's own HParam works unchanged; this file only provides an equivalent loader
for environments where the reference is absent (tests, bench, the GPU box).
"""
import os

import yaml


class DotDict(dict):
    """dict with attribute access, nested dicts converted recursively."""

    def __init__(self, src=None):
        super().__init__()
        for k, v in (src or {}).items():
            self[k] = DotDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # keep hasattr() semantics sane
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def load_hparams(path):
    with open(path, "r") as f:
        merged = {}
        for doc in yaml.safe_load_all(f):
            merged.update(doc or {})
    return DotDict(merged)


_DEFAULT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "configs", "default.yaml")


def default_hparams():
    """The reference's default model (configs/default.yaml:38-66,98-102)."""
    return load_hparams(_DEFAULT)


N_PHONEME_SYMBOLS = 68  # len(valid_symbols), reference dataset/texts/__init__.py:25-94
