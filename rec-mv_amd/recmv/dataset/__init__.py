"""Data path of the reference (dataset/dataset.py): the per-capture directory layout read into the tensors the loop consumes."""
from .dataset import (ClipSampler, Init_Fl_SceneDataset, Large_Pose_SceneDataset, People_Snapshot_SceneDataset, RandomSampler,  # noqa: F401
                      SceneDataset, Synthe_SceneDataset, getDatasetAndLoader, one_euro_smooth, read_image_bgr)
