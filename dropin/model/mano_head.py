"""`model.mano_head` of the reference (model/mano_head.py:177-250)."""
from keypointfusion_amd.model.mano_head import mano_regHead  # noqa: F401
