"""the crop preprocessing against the ONLY reference-authored known-answer files — the 128x128 depth and RGB
crops the authors committed from demo_RGBD.py:588-589 (tests/golden/demo_box_crop{D,RGB}.png).  The source frame fixture is a
460x500 window of the reference's 1920x1080 sample frame (data, not source), re-embedded into a zero frame at its offset."""
import os

import numpy as np
from PIL import Image

from conftest import GOLDEN
from keypointfusion_amd import preprocess as P


def _frame():
    meta = {}
    for line in open(os.path.join(GOLDEN, "demo_box_window.txt")):
        k, *v = line.split()
        meta[k] = [float(x) for x in v]
    y0, x0 = int(meta["window_y0"][0]), int(meta["window_x0"][0])
    H, W = int(meta["frame_h"][0]), int(meta["frame_w"][0])
    rgbw = np.array(Image.open(os.path.join(GOLDEN, "demo_box_rgb_window.png")))
    dw = np.array(Image.open(os.path.join(GOLDEN, "demo_box_depth_window.png")))
    rgb = np.zeros((H, W, 3), np.uint8)
    depth = np.zeros((H, W), np.uint16)
    rgb[y0:y0 + rgbw.shape[0], x0:x0 + rgbw.shape[1]] = rgbw
    depth[y0:y0 + dw.shape[0], x0:x0 + dw.shape[1]] = dw
    # visualization/box_bbox.txt holds the box normalised (cx, cy, w, h); demo_RGBD.py:578-580 turns a centre-format box into
    # top-left xywh.  (The script's hard-coded [885, 515.5, 178, 127] is this box rounded; the committed crops were produced from
    # the un-rounded values: int() of 451.99998 / 973.99999 moves two box edges by one pixel and the centre depth by 0.09 mm.)
    cx, cy, w, h = meta["bbox_norm"]
    bbox = [cx * W, cy * H, w * W, h * H]
    bbox[0] -= bbox[2] / 2
    bbox[1] -= bbox[3] / 2
    return rgb, depth, bbox, tuple(meta["cam"])


def test_crops_reproduce_the_reference_committed_files():
    rgb, depth, bbox, cam = _frame()
    out = P.prepare_rgbd(rgb, depth, bbox, cam)
    want_rgb = np.array(Image.open(os.path.join(GOLDEN, "demo_box_cropRGB.png")))
    want_d = np.array(Image.open(os.path.join(GOLDEN, "demo_box_cropD.png")))
    assert np.array_equal(out["crop_rgb"].astype(np.uint8), want_rgb)                      # bit-exact RGB crop
    got_d = ((out["img"][0] + 1) / 2 * 255.0).astype(np.uint8)                              # demo_RGBD.py:92-93
    assert np.array_equal(got_d, want_d[..., 0]) and np.array_equal(want_d[..., 0], want_d[..., 2])
    assert out["img"].shape == (1, 128, 128) and out["img_rgb"].shape == (3, 128, 128)
    assert float(out["img"].max()) == 1.0 and float(out["img"].min()) >= -1.0           # background exactly 1.0
    assert out["pcl"].shape == (1024, 3) and float(np.abs(out["pcl"]).max()) <= 1.0
    # the crop affine maps the centre of mass to the crop centre
    c = out["M"] @ np.array([out["com"][0], out["com"][1], 1.0])
    assert abs(c[0] - 64) < 1.5 and abs(c[1] - 64) < 1.5


def test_resize_nearest_and_uncrop():
    a = np.arange(35).reshape(5, 7)
    r = P.resize_nearest(a, (3, 2))  # (w, h) like cv2
    assert r.shape == (2, 3) and r[0, 0] == a[0, 0] and r[1, 2] == a[2, 4]
    M = np.array([[0.5, 0, -10.0], [0, 0.5, -20.0], [0, 0, 1.0]])
    pts = np.array([[64.0, 64.0, 500.0]])
    back = P.uncrop_points(pts, M)
    assert np.allclose(back[0, :2], [(64 + 10) / 0.5, (64 + 20) / 0.5]) and back[0, 2] == 500.0


def test_degenerate_inputs():
    depth = np.zeros((200, 200), np.float32)  # nothing in range: centre falls back to the box corner at 300 mm
    c = P.center_from_bbox(depth, [50, 60, 20, 20])
    assert list(c) == [50.0, 60.0, 300.0]
    pcl = P.sample_points(np.zeros((0, 3)), 1024, np.random.RandomState(0))
    assert pcl.shape == (1024, 3) and not pcl.any()
    few = P.sample_points(np.random.RandomState(1).rand(10, 3), 1024, np.random.RandomState(0))  # tiling branch
    assert few.shape == (1024, 3)


import pytest  # noqa: E402


@pytest.mark.gpu
def test_demo_crop_through_the_model_matches_oracle():
    """BASELINE configs[0]: the reference's sample frame, cropped as demo_RGBD.py does, through the full model on the GPU
    (synthetic weights: no checkpoint exists here) against the CPU oracle on the same crop, and back to image coordinates."""
    import torch
    from conftest import synthetic_sd
    from keypointfusion_amd.model.model import KPFusion
    from oracle.compare import oracle_with_device_decisions
    rgb, depth, bbox, cam = _frame()
    pre = P.prepare_rgbd(rgb, depth, bbox, cam)
    b = {k: torch.from_numpy(np.ascontiguousarray(pre[k]))[None] for k in ("img_rgb", "img", "pcl", "center", "M", "cube", "cam_para")}
    net = "KPFusion-convnext-tiny"
    sd = synthetic_sd(net)
    m = KPFusion(net, "", 21, "dexycb", "")
    m.load_state_dict(sd)
    dev = torch.device("cuda:0")
    m = m.to(dev).eval()
    plan = m._plan(dev)
    with torch.no_grad():
        res, sws, ctx = plan.forward(*[b[k].to(dev) for k in ("img_rgb", "img", "pcl", "center", "M", "cube", "cam_para")], 0.8, 128, 1, want_aux=True)
    ref, rsw, _, report = oracle_with_device_decisions(sd, b, ctx)
    for o, r in zip(res + sws, ref + rsw):
        assert float((o.cpu() - r).abs().max() / r.abs().max()) < 1e-3
    # joints of the last stage back in the 1920x1080 frame, as demo_RGBD.py:121-147 does: xyz -> crop pixels (xyz_nl2uvdnl_tensor) ->
    # full image (transformPoints2D with M^-1).  Un-cropping the crop-space projection must land on the direct pinhole projection of
    # the metric joints into the full frame, and the crop-space joints of the ORACLE must un-crop to the same pixels (< 0.05 px).
    xyz = res[5].cpu().numpy()[0]
    assert np.isfinite(xyz).all()
    crop_px = P.project_to_crop(xyz, pre["center"], pre["M"], pre["cube"], pre["cam_para"])
    full = P.uncrop_points(crop_px, pre["M"])
    world = xyz.astype(np.float64) * (np.asarray(pre["cube"], np.float64) / 2) + np.asarray(pre["center"], np.float64)
    fx, fy, u0, v0 = [float(c) for c in pre["cam_para"]]
    direct = np.stack([world[:, 0] * fx / world[:, 2] + u0, world[:, 1] * fy / world[:, 2] + v0], 1)
    assert np.abs(full[:, :2] - direct).max() < 1e-6 * np.abs(direct).max()
    full_ref = P.uncrop_points(P.project_to_crop(ref[5].numpy()[0], pre["center"], pre["M"], pre["cube"], pre["cam_para"]), pre["M"])
    assert np.abs(full[:, :2] - full_ref[:, :2]).max() < 0.05
