"""Drop-in for the reference's utils/custom_transforms.py — see multi-modal-qg_amd/data.py."""
import importlib as _il

_m = _il.import_module("multi-modal-qg_amd.data")
prepare_sequence, Resize, ToFloatTensor, Normalize = _m.prepare_sequence, _m.Resize, _m.ToFloatTensor, _m.Normalize
