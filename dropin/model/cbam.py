"""`model.cbam` of the reference (model/cbam.py:84-94)."""
from keypointfusion_amd.model.cbam import CBAM  # noqa: F401
