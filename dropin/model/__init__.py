"""Replacement for the reference's `model/` package directory (ru1ven/KeypointFusion): copy or symlink this directory over
`<reference>/model` (with the repository root on PYTHONPATH) and `from model.model import KPFusion` (train.py:5-6, demo_RGBD.py:46)
resolves to the MI355X implementation with no edit to the reference's scripts.  Each module re-exports the class of the same name
from keypointfusion_amd.model (same constructor, forward signature and state-dict keys)."""
