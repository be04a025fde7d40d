import os, sys
sys.path.insert(0, os.getcwd())
import torch
mode = sys.argv[1]
if mode == "late":      # HIP initialised before the package is imported
    torch.zeros(1, device="cuda")
import keypointfusion_amd
from keypointfusion_amd.graphs import replay_is_sound
print(mode, "env", os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE"), "sound", replay_is_sound())
