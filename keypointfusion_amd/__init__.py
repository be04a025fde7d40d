"""MI355X-native implementation of the KeypointFusion (ru1ven/KeypointFusion) forward hot path.

    from keypointfusion_amd.model.model import KPFusion     # drop-in for `from model.model import KPFusion`

Importing this package (and the inference path under it: model, engine, serving, heads) changes nothing in the host process.  The one
process-wide setting the library ever makes — DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, a HIP-runtime workaround that only captured TRAINING
iterations need — is made by `keypointfusion_amd.graphs.prepare_training_graphs()`, which runs when `keypointfusion_amd.training` is imported
and again when a `GraphedTrainStep` is built."""
__all__ = ["spec", "weights", "lib", "engine"]
