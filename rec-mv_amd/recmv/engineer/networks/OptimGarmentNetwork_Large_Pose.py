"""`OptimGarmentNetwork_LargePose` (engineer/networks/OptimGarmentNetwork_Large_Pose.py:122-475 of the reference): the
same loop with the SDF nets frozen (`freeze_sdf`, :130-137), the curve losses zero-weighted (:219) and no SDF-parameter
term in the implicit differentiation (:440-452) — `HotLoop(large_pose=True)`."""
from .OptimGarmentNetwork import OptimGarmentNetwork


class OptimGarmentNetwork_LargePose(OptimGarmentNetwork):
    def __init__(self, *args, **kwargs):
        kwargs['large_pose'] = True
        super().__init__(*args, **kwargs)
