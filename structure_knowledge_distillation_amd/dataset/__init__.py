"""Device-side input pipeline (reference: dataset/datasets.py)."""
from .datasets import CSDataSet, CSTrainTransform, ID_TO_TRAINID, draw_sample_params, trainid_lut  # noqa: F401
