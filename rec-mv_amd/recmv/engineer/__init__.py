"""recmv.engineer — the `engineer` package surface of the reference that the per-frame optimisation path touches:
`engineer.networks.{OptimGarmentNetwork, OptimGarmentNetwork_Large_Pose}` (the objects train.py / train_large_pose.py
drive) and `engineer.core.{fl_optimizer, beta_optimizer}` (entry points named by the north star)."""
