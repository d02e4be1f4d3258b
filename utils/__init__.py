"""Drop-in ``utils`` package (reference train.py:19-20 imports utils.dataset / utils.custom_transforms)."""
