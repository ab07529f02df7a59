"""train_large_pose.py — the reference's large-pose driver (train_large_pose.py:20-344) on the MI355X hot loop.

The loop body is train.py's; what the reference changes for this stage is kept: no `--a_pose` / `--resume` flags
(:20-37), the model resumes from `<data>/<save-folder>/a-pose.pth` (:39) and continues at epoch 60 (:210), the epochs
run to `nepoch` inclusive (:289), and the optimisation object is `OptimGarmentNetwork_LargePose`
(`getOptNet(..., opt_large=True)`, model/network.py:337-340): SDF nets frozen, curve losses zero-weighted, no
SDF-parameter term in the implicit differentiation (engineer/networks/OptimGarmentNetwork_Large_Pose.py:130-137, :219,
:440-452) — only the deformation field, the per-frame codes / poses, the colour net and the camera are optimised.
"""
import os.path as osp
import sys

sys.path.insert(0, osp.dirname(osp.abspath(__file__)))
import train  # noqa: E402


def main(argv=None):
    return train.main(argv, large_pose=True)


if __name__ == '__main__':
    main()
