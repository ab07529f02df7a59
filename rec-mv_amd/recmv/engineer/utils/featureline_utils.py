"""Feature-line annotation files (engineer/utils/featureline_utils.py:19-43 of the reference): labelme-style json with a
`shapes` list, every shape a `label` and its 2-D `points`."""
import json

import numpy as np


def _shapes(name):
    with open(name) as fh:
        return json.load(fh)['shapes']


def check_feature_lines(name):
    """A file may annotate every feature line at most once."""
    seen = set()
    for shape in _shapes(name):
        assert shape['label'] not in seen, "label conflict"
        seen.add(shape['label'])


def obtain_feature_lines(name):
    """{label: float32 [n,2] polyline} of one annotation file."""
    return {shape['label']: np.asarray(shape['points']).astype(np.float32) for shape in _shapes(name)}
