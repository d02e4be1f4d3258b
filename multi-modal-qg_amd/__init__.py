"""MI355X-native hot path of ksg14/multi-modal-qg: the multimodal encoder -> attention-decoder
training step on hand-written gfx950 HIP kernels behind the reference's ``model.encoder`` /
``model.decoder`` class API.

The directory name is not a Python identifier; import it with
``importlib.import_module("multi-modal-qg_amd")`` or through the ``mmqg_amd`` alias module at the
repository root.
"""
from . import _lib  # noqa: F401
from .config import Config  # noqa: F401

__all__ = ["Config", "_lib"]
__version__ = "0.1.0"
