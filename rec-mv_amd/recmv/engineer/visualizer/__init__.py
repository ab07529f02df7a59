"""engineer.visualizer of the reference: the logging sink `getOptNet(..., visualizer=)` hands to the optimisation object."""
from .wandb_visualizer import wandb_visualizer  # noqa: F401
