"""RGB-D crop preprocessing in front of the forward path: bounding box -> centre of mass -> metric crop ->
nearest-neighbour resize -> depth normalisation -> point-cloud sampling, and the un-crop of predicted joints.

Restates demo_RGBD.py:253-276 (get_center_from_bbx), :410-569 (Crop_Image_deep_pp[_RGB], comToBounds, getCrop), :378-385
(normalize_img), :345-376 (getpcl / depthToPCL), :300-343 (process_depth sampling), :192-212 (transformPoints2D) — the same
code lives in dataloader/loader.py:604-750, 843-893.  Host-side numpy like the reference's (the dataloader stays Python per
BASELINE.json north_star); OpenCV is not a dependency: cv2.resize(INTER_NEAREST) is restated (src = floor(dst * src/dst)).
Pinned by the reference's own committed crops (tests/test_preprocess.py).
"""
import math

import numpy as np


def center_from_bbox(depth, bbox, upper=1500, lower=171):
    """demo_RGBD.py:253-276 — (u, v, d_mm) centre of mass of the valid depth inside an xywh box (x,y = top-left)."""
    c = np.array([0.0, 0.0, 300.0])
    x0, x1 = int(bbox[0]), int(bbox[0] + bbox[2])
    y0, y1 = int(bbox[1]), int(bbox[1] + bbox[3])
    img = depth[y0:y1, x0:x1]
    flag = np.logical_and(img <= upper, img >= lower)
    xv, yv = np.meshgrid(np.linspace(0, img.shape[1], img.shape[1]), np.linspace(0, img.shape[0], img.shape[0]))
    if flag.any():
        c[0], c[1], c[2] = np.mean(xv[flag]), np.mean(yv[flag]), np.mean(img[flag])
        if c[2] <= 0:
            c[2] = 300.0
    else:
        c[:] = (0.0, 0.0, 300.0)
    c[0] += bbox[0]
    c[1] += bbox[1]
    return c


def com_to_bounds(com, size, cam):
    """demo_RGBD.py:520-530 — pixel bounds of a metric cube of `size` mm around the centre of mass."""
    fx, fy, _, _ = cam
    zs, ze = com[2] - size[2] / 2.0, com[2] + size[2] / 2.0
    xs = int(np.floor((com[0] * com[2] / fx - size[0] / 2.0) / com[2] * fx + 0.5))
    xe = int(np.floor((com[0] * com[2] / fx + size[0] / 2.0) / com[2] * fx + 0.5))
    ys = int(np.floor((com[1] * com[2] / fy - size[1] / 2.0) / com[2] * fy + 0.5))
    ye = int(np.floor((com[1] * com[2] / fy + size[1] / 2.0) / com[2] * fy + 0.5))
    return xs, xe, ys, ye, zs, ze


def get_crop(img, xs, xe, ys, ye, zs, ze, thresh_z=True, background=0):
    """demo_RGBD.py:532-569 — crop with zero padding outside the image; depth values clamped to the cube in z."""
    H, W = img.shape[:2]
    c = img[max(ys, 0):min(ye, H), max(xs, 0):min(xe, W)].copy()
    pad = ((abs(ys) - max(ys, 0), abs(ye) - min(ye, H)), (abs(xs) - max(xs, 0), abs(xe) - min(xe, W)))
    if img.ndim == 3:
        pad = pad + ((0, 0),)
    c = np.pad(c, pad, mode="constant", constant_values=background)
    if thresh_z:
        m1 = np.logical_and(c < zs, c != 0)
        m2 = np.logical_and(c > ze, c != 0)
        c[m1] = zs
        c[m2] = 0.0
    return c


def resize_nearest(src, dsize):
    """cv2.resize(src, (w, h), interpolation=cv2.INTER_NEAREST): dst[y, x] = src[floor(y * sh/h), floor(x * sw/w)]."""
    w, h = dsize
    sh, sw = src.shape[:2]
    xi = np.minimum(np.floor(np.arange(w) * (sw / float(w))).astype(np.int64), sw - 1)
    yi = np.minimum(np.floor(np.arange(h) * (sh / float(h))).astype(np.int64), sh - 1)
    return src[yi][:, xi]


def crop_image(img, com, size, dsize, cam, thresh_z):
    """demo_RGBD.py:410-518 — returns (dsize crop, 3x3 image->crop affine)."""
    xs, xe, ys, ye, zs, ze = com_to_bounds(com, size, cam)
    cropped = get_crop(img, xs, xe, ys, ye, zs, ze, thresh_z=thresh_z)
    wb, hb = xe - xs, ye - ys
    sz = (dsize[0], int(hb * dsize[0] / wb)) if wb > hb else (int(wb * dsize[1] / hb), dsize[1])
    trans = np.eye(3)
    trans[0, 2], trans[1, 2] = -xs, -ys
    if cropped.shape[0] > cropped.shape[1]:
        scale = np.eye(3) * sz[1] / float(cropped.shape[0])
    else:
        scale = np.eye(3) * sz[0] / float(cropped.shape[1])
    scale[2, 2] = 1
    rz = resize_nearest(cropped, sz)
    shape = (dsize[1], dsize[0]) + ((img.shape[2],) if img.ndim == 3 else ())
    ret = np.zeros(shape, np.float32)
    x0 = int(np.floor(dsize[0] / 2.0 - rz.shape[1] / 2.0))
    y0 = int(np.floor(dsize[1] / 2.0 - rz.shape[0] / 2.0))
    ret[y0:y0 + rz.shape[0], x0:x0 + rz.shape[1]] = rz
    off = np.eye(3)
    off[0, 2], off[1, 2] = x0, y0
    return ret, np.dot(off, np.dot(scale, trans))


def normalize_depth(crop, com, cube):
    """demo_RGBD.py:378-385 — background / out-of-cube -> far plane, then (d - com_z) / (cube_z/2) in [-1, 1].
    All scalars are rounded to float32 first: the reference runs under NumPy 1.x value-based casting, where a float64 scalar
    combined with a float32 array computes in float32 (NumPy 2 would silently compute in float64 and round differently)."""
    d = crop.astype(np.float32, copy=True)
    far = np.float32(com[2] + cube[2] / 2.0)
    near = np.float32(com[2] - cube[2] / 2.0)
    premax = d.max()
    d[d == premax] = far
    d[d == 0] = far
    d[d >= far] = far
    d[d <= near] = near
    d -= np.float32(com[2])
    d /= np.float32(cube[2] / 2.0)
    return d


def image_to_3d(uvd, cam, flip=1):
    """demo_RGBD.py:387-408 jointImgTo3D."""
    fx, fy, fu, fv = cam
    uvd = np.asarray(uvd, np.float64)
    out = np.zeros_like(uvd, np.float32)
    out[..., 0] = (uvd[..., 0] - fu) * uvd[..., 2] / fx
    out[..., 1] = flip * (uvd[..., 1] - fv) * uvd[..., 2] / fy
    out[..., 2] = uvd[..., 2]
    return out


def depth_to_pcl(img_n, com3d, cube, M, cam, flip=1):
    """demo_RGBD.py:345-376 — normalised crop -> all foreground points, normalised to the cube."""
    fx, fy, fu, fv = cam
    mask = np.isclose(img_n, 1)
    dpt = img_n * cube[2] / 2.0 + com3d[2]
    dpt[mask] = 0
    valid = ~np.isclose(dpt, 0.0)
    pts = np.asarray(np.where(valid)).transpose()
    pts = np.concatenate([pts[:, [1, 0]] + 0.5, np.ones((pts.shape[0], 1), dtype="float32")], axis=1)
    pts = np.dot(np.linalg.inv(np.asarray(M)), pts.T).T
    pts = (pts[:, 0:2] / pts[:, 2][:, None]).reshape((pts.shape[0], 2))
    depth = dpt[valid]
    xyz = np.column_stack(((pts[:, 0] - fu) / fx * depth, flip * (pts[:, 1] - fv) / fy * depth, depth))
    return (xyz - com3d) / (np.asarray(cube) / 2.0)


def sample_points(pcl, n, rng):
    """demo_RGBD.py:319-332 — n points without replacement (tiling the cloud first when it is smaller than n), clamped to [-1,1].
    `rng` is a numpy RandomState (the reference uses the global np.random seeded with 0)."""
    num = pcl.shape[0]
    if num == 0:
        return np.zeros([n, 3], np.float32)
    idx = np.arange(num)
    if num < n:
        idx = np.append(idx.repeat(math.floor(n / num)), rng.choice(idx, size=divmod(n, num)[1], replace=False))
    sel = rng.choice(idx, n, replace=False)
    return np.clip(pcl[sel, :], -1, 1).astype(np.float32)


def prepare_rgbd(rgb, depth, bbox, cam, cube=(250.0, 250.0, 250.0), img_size=128, sample_num=1024, rng=None):
    """Everything demo_RGBD.py:72-108 does before the forward.  rgb H x W x 3 (any channel order, kept), depth H x W (mm),
    bbox xywh with x,y the top-left corner.  Returns numpy arrays ready to batch: img_rgb 3xSxS in [0,1], img 1xSxS in [-1,1],
    pcl Nx3, center (3,) mm, M 3x3, cube (3,), cam_para (4,)."""
    rng = np.random.RandomState(0) if rng is None else rng
    depth = np.asarray(depth)  # keep the sensor dtype (uint16 mm from cv2.imread(..., IMREAD_ANYDEPTH)): the reference's means and
    com = center_from_bbox(depth, bbox)  # z-clamps are computed on it (float64 mean, integer truncation of the near plane)
    crop_rgb, _ = crop_image(np.asarray(rgb), com, cube, (img_size, img_size), cam, thresh_z=False)
    crop_d, M = crop_image(depth, com, cube, (img_size, img_size), cam, thresh_z=True)
    img_n = normalize_depth(crop_d, com, cube)
    com3d = image_to_3d(com, cam)
    pcl = sample_points(depth_to_pcl(img_n, com3d, np.asarray(cube, np.float64), M, cam), sample_num, rng)
    return dict(img_rgb=(crop_rgb.astype(np.float32) / 255.0).transpose(2, 0, 1), img=img_n[None].astype(np.float32), pcl=pcl,
                center=com3d.astype(np.float32), M=M.astype(np.float32), cube=np.asarray(cube, np.float32),
                cam_para=np.asarray(cam, np.float32), crop_rgb=crop_rgb, com=com)


def project_to_crop(xyz_nl, center, M, cube, cam, img_size=128, flip=1):
    """loader.xyz_nl2uvdnl_tensor (dataloader/loader.py:821-834; demo_RGBD.py:121-123): normalised xyz joints -> pixel coordinates in
    the crop (u, v in [0, img_size), d in mm): de-normalise, pinhole-project (points3DToImg), apply the crop matrix M."""
    xyz = np.asarray(xyz_nl, np.float64) * (np.asarray(cube, np.float64) / 2.0) + np.asarray(center, np.float64)
    fx, fy, u0, v0 = [float(c) for c in cam]
    u = xyz[:, 0] * fx / xyz[:, 2] + u0
    v = flip * xyz[:, 1] * fy / xyz[:, 2] + v0
    h = np.stack([u, v, np.ones_like(u)], 1) @ np.asarray(M, np.float64).T
    return np.stack([h[:, 0], h[:, 1], xyz[:, 2]], 1)


def uncrop_points(uv, M):
    """demo_RGBD.py:192-212 — crop pixel coordinates back to the original image (projective divide included)."""
    Mi = np.linalg.inv(np.asarray(M, np.float64))
    uv = np.asarray(uv, np.float64)
    h = np.concatenate([uv[:, :2], np.ones((uv.shape[0], 1))], 1) @ Mi.T
    out = uv.copy()
    out[:, :2] = h[:, :2] / h[:, 2:3]
    return out
