"""`model.hourglass` of the reference (model/hourglass.py:173-236): PoseNet on the HIP path."""
from keypointfusion_amd.model.hourglass import PoseNet  # noqa: F401
