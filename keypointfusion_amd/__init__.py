"""MI355X-native implementation of the KeypointFusion (ru1ven/KeypointFusion) forward hot path.

    from keypointfusion_amd.model.model import KPFusion     # drop-in for `from model.model import KPFusion`
"""
__all__ = ["spec", "weights", "lib", "engine"]
