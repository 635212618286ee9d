"""MI355X-native FastSpeech2 mel-generation forward pass.

``FeedForwardTransformer`` is a drop-in for the reference's ``fastspeech.FeedForwardTransformer``
(reference fastspeech.py:28); its arithmetic runs in ``libfs2_hip.so`` (C ABI: include/fs2.h).
"""
from .hparams import default_hparams, load_hparams, N_PHONEME_SYMBOLS  # noqa: F401
from .fastspeech import FeedForwardTransformer, StepStreams  # noqa: F401
from .io import load_checkpoint, vocoder_input, hparams_from_str  # noqa: F401
