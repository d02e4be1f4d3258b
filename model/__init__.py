"""Drop-in ``model`` package: ``from model.encoder import AudioVideoEncoder, TextEncoder`` and
``from model.decoder import AttnDecoder, Decoder`` (reference train.py:15-16) resolve to the
MI355X implementations in ``multi-modal-qg_amd/model``."""
