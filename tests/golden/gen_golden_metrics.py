"""Known answers of the reference's per-joint error metric (train.py:470-488 `Trainer.xyz2error`), including its NYU branch
(joint_num == 23: 14 of the 23 joints are scored).  Run in the build container only (imports /root/reference); writes
tests/golden/metrics_xyz2error.npz = inputs + the reference's outputs."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402


def main():
    if not ref_import.reference_available():
        sys.exit("reference tree not found; golden vectors can only be generated in the build container")
    ref_import.install_shims()
    for name in ("tensorboardX",):  # train.py:26 imports it at module level; nothing of it is used by xyz2error
        m = types.ModuleType(name)
        m.SummaryWriter = object
        sys.modules.setdefault(name, m)
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    cwd, argv = os.getcwd(), sys.argv
    os.chdir(ref_import.REF_ROOT)
    sys.argv = argv[:1]
    try:
        import train as ref_train
    finally:
        os.chdir(cwd)
        sys.argv = argv
    rng = np.random.default_rng(23)
    out = {}
    for J in (21, 23, 14):
        pred = rng.uniform(-0.6, 0.6, size=(5, J, 3)).astype(np.float32)
        gt = (pred + rng.normal(size=(5, J, 3)) * 0.05).astype(np.float32)
        center = (np.array([0, 0, 600.0]) + rng.uniform(-30, 30, size=(5, 3))).astype(np.float32)
        cube = np.tile(np.array([250.0, 250.0, 250.0], np.float32), (5, 1))
        cube[3] = (300.0, 280.0, 260.0)
        err = ref_train.Trainer.xyz2error(None, torch.from_numpy(pred), torch.from_numpy(gt), torch.from_numpy(center), torch.from_numpy(cube))
        out.update({"pred%d" % J: pred, "gt%d" % J: gt, "center%d" % J: center, "cube%d" % J: cube, "err%d" % J: np.asarray(err)})
        print("J = %d -> errors %s" % (J, np.asarray(err).shape))
    path = os.path.join(HERE, "metrics_xyz2error.npz")
    np.savez_compressed(path, **out)
    print("wrote", path)


if __name__ == "__main__":
    main()
