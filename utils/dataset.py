"""Drop-in for the reference's utils/dataset.py — see multi-modal-qg_amd/data.py."""
import importlib as _il

VQGDataset = _il.import_module("multi-modal-qg_amd.data").VQGDataset
