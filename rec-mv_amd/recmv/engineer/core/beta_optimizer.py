"""engineer/core/beta_optimizer.py of the reference: `smpl_beta_optimizer` (:132) fits the SMPL shape to 2-D joints once,
before the loop (dataset preparation) — outside the hot-path scope.  Present under its name so that
an importer fails at the call with a clear message."""


def smpl_beta_optimizer(*args, **kwargs):
    raise NotImplementedError(
        "engineer.core.beta_optimizer.smpl_beta_optimizer (engineer/core/beta_optimizer.py:132) is a pre-processing "
        "step of the reference (SMPL shape fit before the optimisation loop); this package implements the per-frame "
        "optimisation hot path only (SURVEY.md §8).")
