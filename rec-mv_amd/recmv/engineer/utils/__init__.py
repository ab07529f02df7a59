"""engineer.utils of the reference: the helpers the data path uses (feature-line annotation files, polyline resampling) and the
rigid / scale transforms of the feature-line templates (start-up registration)."""
from . import featureline_utils, matrix_transform, polygons  # noqa: F401
