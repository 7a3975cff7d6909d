"""TEST INFRASTRUCTURE (never imported by the product): CPU restatement of the reference's occupancy ground-truth
loaders, the transforms that put ``gt_occ`` / ``points_occ`` / ``bda_rot`` beside ``img_inputs``:
    projects/mmdet3d_plugin/datasets/pipelines/loading_nusc_occ.py
        :47-57    sample_3d_augmentation (numpy global RNG: rot, scale, flip_dx, flip_dy, flip_dz -- draw order kept)
        :59-125   LoadNuscOccupancyAnnotations.__call__ (learning map, BEV augmentation of the points, voxelisation)
        :127-143  nb_process_label (majority label per voxel; numba there, the same loop vectorised here)
        :145-203  voxel_transform (bda matrix 3x3; flips of the label volume)
        :205-224  custom_rotate_3d (PIL nearest rotate of every height slice, fill 255)
    projects/mmdet3d_plugin/datasets/pipelines/loading_kitti_occ.py
        :17-26    sample_bda_augmentation
        :35-55    LoadSemKittiAnnotation.__call__
        :57-116   voxel_transform (bda matrix 4x4 about the centre of the point-cloud range)
in plain numpy.  ``Image.rotate`` is Pillow (third party, not vendored): restated as in oracle/image_pipeline_ref.py
(Geometry.c affine_fixed, 16.16 fixed point) plus the transpose fast paths Image.rotate takes for 90 / 180 / 270 degrees
and the pre-filled destination ``fillcolor`` gives; pinned against Pillow itself in tests/test_pipeline_ops.py and
against the reference's own functions imported through tests/refshim (tests/golden/make_golden_occ_loading.py ->
tests/golden/occ_loading.npz; numba, absent here, is replaced there by a pass-through ``jit``, i.e. the reference's loop
runs as plain Python)."""

import numpy as np

from .image_pipeline_ref import rotate_affine


def sample_bda(conf, rng=np.random):
    """loading_nusc_occ.py:47-57 / loading_kitti_occ.py:17-26 (flip_dz_ratio defaults to 0 in the nuScenes loader)"""
    rot = rng.uniform(*conf["rot_lim"])
    scale = rng.uniform(*conf["scale_lim"])
    fx = rng.uniform() < conf["flip_dx_ratio"]
    fy = rng.uniform() < conf["flip_dy_ratio"]
    fz = rng.uniform() < conf.get("flip_dz_ratio", 0.0)
    return rot, scale, fx, fy, fz


def bda_matrix(rotate_deg, flip_dx, flip_dy, flip_dz, center=None):
    """flip @ rotation in float32 (3x3, loading_nusc_occ.py:147-180), or denorm @ flip @ rot @ norm (4x4,
    loading_kitti_occ.py:59-101) when ``center`` is given"""
    a = np.float32(rotate_deg / 180 * np.pi)
    s, c = np.sin(a, dtype=np.float32), np.cos(a, dtype=np.float32)
    rot = np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    flip = np.eye(4, dtype=np.float32)
    for on, axis in ((flip_dx, 0), (flip_dy, 1), (flip_dz, 2)):
        if on:
            f = np.eye(4, dtype=np.float32)
            f[axis, axis] = -1
            flip = flip @ f
    if center is None:
        return (flip @ rot)[:3, :3]
    norm, denorm = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
    norm[:3, 3] = -np.asarray(center, np.float32)
    denorm[:3, 3] = np.asarray(center, np.float32)
    return denorm @ flip @ rot @ norm


def rotate_slice(lab, angle, fill=255):
    """``Image.fromarray(lab).rotate(angle, resample=NEAREST, fillcolor=fill)`` of one uint8 [rows, cols] slice"""
    H, W = lab.shape
    angle = angle % 360.0
    if angle == 0:
        return lab.copy()
    if angle == 180:
        return lab[::-1, ::-1].copy()
    if angle in (90, 270) and H == W:
        return np.rot90(lab, 1 if angle == 90 else 3).copy()
    _, a = rotate_affine(W, H, angle)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.int64)
    xin = (a[2] + a[1] * ys + a[0] * xs) >> 16
    yin = (a[5] + a[4] * ys + a[3] * xs) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    out = np.full_like(lab, fill)
    out[ok] = lab[yin[ok], xin[ok]]
    return out


def custom_rotate_3d(vox, angle):
    """loading_nusc_occ.py:205-224: every [X, Y] height slice rotated like an image"""
    vox = vox.astype(np.uint8)
    return np.stack([rotate_slice(vox[..., z], angle) for z in range(vox.shape[-1])], -1)


def voxel_transform(vox, rotate_deg, flip_dx, flip_dy, flip_dz, center=None):
    """label volume [X, Y, Z] -> (int64 volume, bda matrix): rotate (unless ~0), then flip z, y, x"""
    mat = bda_matrix(rotate_deg, flip_dx, flip_dy, flip_dz, center)
    if vox is None:
        return None, mat
    vox = np.asarray(vox).astype(np.uint8)
    if not np.isclose(rotate_deg, 0):
        vox = custom_rotate_3d(vox, rotate_deg)
    if flip_dz:
        vox = vox[:, :, ::-1]
    if flip_dy:
        vox = vox[:, ::-1]
    if flip_dx:
        vox = vox[::-1]
    return vox.astype(np.int64), mat


def majority_labels(grid_ind, labels, grid_size, empty_id):
    """nb_process_label (loading_nusc_occ.py:127-143): per voxel the label most points carry, the smallest label on a
    tie (np.argmax of the counter); voxels without points keep ``empty_id``"""
    gs = [int(v) for v in grid_size]
    lin = (grid_ind[:, 0] * gs[1] + grid_ind[:, 1]) * gs[2] + grid_ind[:, 2]
    labels = labels.astype(np.int64)
    counts = np.zeros((gs[0] * gs[1] * gs[2], int(labels.max()) + 1), np.int64)   # (columns past the largest label stay 0)
    np.add.at(counts, (lin, labels), 1)
    counts = counts.astype(np.uint16)                                # the reference's counter type (wraps at 65536)
    out = np.full(gs[0] * gs[1] * gs[2], empty_id, np.uint8)
    hit = np.zeros(gs[0] * gs[1] * gs[2], bool)
    hit[lin] = True
    out[hit] = counts[hit].argmax(1)
    return out.reshape(gs)


def nusc_occupancy(points, point_labels, learning_map, grid_size, pc_range, bda_rot, unoccupied_id=17):
    """loading_nusc_occ.py:76-123 from the decoded arrays: points float32 [P, 3], raw lidarseg labels uint8 [P]
    -> (gt_occ int64 [X, Y, Z], points_occ float32 [P, 4])"""
    grid_size, pc_range = np.array(grid_size), np.array(pc_range)
    voxel_size = (pc_range[3:] - pc_range[:3]) / grid_size
    lab = np.vectorize(learning_map.__getitem__)(point_labels.reshape(-1, 1))
    pts = points.astype(np.float32) @ np.asarray(bda_rot, np.float32).T
    lidarseg = np.concatenate([pts, lab], -1)                        # float64 from here on, as in the reference
    eps = 1e-5
    ind = np.floor((np.clip(lidarseg[:, :3], pc_range[:3], pc_range[3:] - eps) - pc_range[:3]) / voxel_size).astype(np.int64)
    out = majority_labels(ind, lidarseg[:, 3].astype(np.int64), grid_size, unoccupied_id)
    out[out == 0] = 255
    out[out == unoccupied_id] = 0
    return out.astype(np.int64), lidarseg.astype(np.float32)
