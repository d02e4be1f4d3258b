"""Drop-in for the reference's model/encoder.py — see multi-modal-qg_amd/model/encoder.py."""
import importlib as _il

_m = _il.import_module("multi-modal-qg_amd.model.encoder")
AudioEncoder = _m.AudioEncoder
VideoResnetEncoder = _m.VideoResnetEncoder
VideoConvLstmEncoder = _m.VideoConvLstmEncoder
TextEncoder = _m.TextEncoder
AudioVideoEncoder = _m.AudioVideoEncoder
