from .OptimGarmentNetwork import OptimGarmentNetwork  # noqa: F401
from .OptimGarmentNetwork_Large_Pose import OptimGarmentNetwork_LargePose  # noqa: F401
