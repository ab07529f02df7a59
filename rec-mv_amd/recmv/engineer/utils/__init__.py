"""engineer.utils of the reference: the two helpers the data path uses (feature-line annotation files, polyline resampling)."""
from . import featureline_utils, polygons  # noqa: F401
