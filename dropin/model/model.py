"""`model.model` of the reference (model/model.py:354-426): KPFusion on the HIP path."""
from keypointfusion_amd.model.model import KPFusion  # noqa: F401
