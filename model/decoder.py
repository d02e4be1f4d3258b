"""Drop-in for the reference's model/decoder.py — see multi-modal-qg_amd/model/decoder.py."""
import importlib as _il

_m = _il.import_module("multi-modal-qg_amd.model.decoder")
AttnDecoder = _m.AttnDecoder
Decoder = _m.Decoder
