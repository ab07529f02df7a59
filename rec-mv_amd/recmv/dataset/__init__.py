"""Data path of the reference (dataset/dataset.py): the per-capture directory layout read into the tensors the loop consumes."""
from .dataset import (ClipSampler, People_Snapshot_SceneDataset, RandomSampler, SceneDataset,  # noqa: F401
                      getDatasetAndLoader, read_image_bgr)
