"""MI355X-native frame-importance scorer (drop-in for the reference ``model.SimNet``).

The directory name carries a hyphen, so import it with
``importlib.import_module("video-summarization_amd")`` or through the root-level alias
module ``video_summarization_amd``.
"""
from . import _lib, synth  # noqa: F401
from .simnet import SimNet, score_frames  # noqa: F401
from .pretrain import PretrainModel  # noqa: F401
from .losses import mse_with_mask_loss  # noqa: F401

__all__ = ["SimNet", "PretrainModel", "score_frames", "synth", "mse_with_mask_loss"]
