"""Drop-in for the reference's config.py — see multi-modal-qg_amd/config.py."""
import importlib as _il

Config = _il.import_module("multi-modal-qg_amd.config").Config
