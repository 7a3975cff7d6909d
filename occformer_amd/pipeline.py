"""Input-pipeline and evaluation stages of the reference that sit on the HOST around the model, on the device
(kernels: csrc/pipeline.hip):

  * ``CreateDepthFromLiDAR``  -- projects/mmdet3d_plugin/datasets/pipelines/lidar2depth.py:9-87 (registered under the
    reference's pipeline name): LiDAR points -> sparse per-camera depth maps ``gt_depths`` (``img_inputs[6]``), what
    ``get_depth_loss`` supervises DepthNet with.  File decoding stays with the caller (``results['points']`` /
    ``from_points``): the reference reads ``.bin`` files with numpy inside the data-loader workers.
  * ``SSCMetrics``            -- projects/mmdet3d_plugin/utils/ssc_metric.py:13-175 (a torchmetrics ``Metric`` there):
    same ``update`` / ``compute`` / ``compute_single`` contract, state = one int64 confusion matrix on the device that the
    kernel accumulates into (no host synchronisation per sample; ``reduce`` all-reduces it over RCCL like
    ``dist_reduce_fx='sum'``).
  * ``LoadMultiViewImageFromFiles_OccFormer`` -- loading_nusc_imgs.py:9-193, the ``img_inputs`` producer (registered
    under the reference's pipeline name): augmentation draws in the reference's numpy order, ``img_transform_core``
    (PIL resize -> crop -> flip -> rotate) and ``mmlabNormalize`` on the device from decoded uint8 frames
    (``image_transform``: Pillow's antialiased bicubic resampling and its fixed-point nearest-neighbour rotation restated
    bit for bit), the post-homography (``image_post_homography``, :35-55) and the calibration tensors.  JPEG decoding is
    file I/O and stays with the caller (``cam['img']``) or PIL (``cam['data_path']``).

  * ``LoadMultiViewImageFromFiles_SemanticKitti`` -- loading_kitti_imgs.py:11-145, the monocular producer of the
    SemanticKITTI configs on the same image path.
  * ``LoadNuscOccupancyAnnotations`` / ``LoadSemKittiAnnotation`` -- loading_nusc_occ.py:13-224 /
    loading_kitti_occ.py:7-116: ``gt_occ``, ``points_occ`` and the BEV-augmentation matrix ``bda_rot`` (``img_inputs[6]``).
    The lidarseg voxelisation (majority label per voxel), the flips and the PIL-style rotation of the label volume are
    device tensor programs (integer sorts / scans / gathers: index work, bit-exact), no kernel of their own yet.
  * ``OccDefaultFormatBundle3D`` / ``Collect3D`` / ``Compose`` / ``collate`` -- the remaining stages of the configs'
    ``train_pipeline`` / ``test_pipeline`` lists, so that the lists build by name and feed ``forward_train`` directly.

The image / depth / metric kernels have no CPU implementation: host tensors raise in ``occformer_amd.ops``; the tensor
programs of the ground-truth loaders refuse them the same way under the product binding (``_device_only``)."""
import math

import torch

from .ops import get_ops
from .registry import Registry

PIPELINES = Registry("pipeline")


def pack_depth_cameras(rots, trans, intrins, post_rots, post_trans):
    """[N, 36] camera constants of ``occf_lidar_depth_fwd``: inv(rots) | trans | intrins (3x3 or 4x4, row-major, in a
    16-float slot) | post_rots[:2, :2] | post_trans[:2] | 2 pad.  ``rots.inverse()`` as in lidar2depth.py:24 (computed
    by torch on the tensors' own device)."""
    N = rots.shape[0]
    inv = torch.linalg.inv(rots.float())
    cam = torch.zeros((N, 36), dtype=torch.float32, device=rots.device)
    cam[:, 0:9] = inv.reshape(N, 9)
    cam[:, 9:12] = trans.float()
    k = intrins.shape[-1]
    cam[:, 12:12 + k * k] = intrins.float().reshape(N, k * k)
    cam[:, 28:32] = post_rots.float()[:, :2, :2].reshape(N, 4)
    cam[:, 32:34] = post_trans.float()[:, :2]
    return cam, k == 4


def create_depth_from_lidar(points, rots, trans, intrins, post_rots, post_trans, img_hw):
    """lidar2depth.py:21-41 (project_points) + :55-80 (valid mask, nearest return per pixel) for ONE sample:
    points [P, >=3] (x, y, z first), rots [N, 3, 3], trans [N, 3], intrins [N, 3, 3] or [N, 4, 4], post_rots [N, 3, 3],
    post_trans [N, 3] -> gt_depths [N, H, W]"""
    cam, kitti = pack_depth_cameras(rots, trans, intrins, post_rots, post_trans)
    H, W = img_hw
    return get_ops().lidar_depth(points, cam, rots.shape[0], int(H), int(W), kitti)


@PIPELINES.register_module()
class CreateDepthFromLiDAR:
    """Same constructor keys and ``results`` contract as the reference transform; the LiDAR points are taken from
    ``results['points']`` (a [P, >=3] tensor already on the device) -- reading ``pts_filename`` is the caller's I/O."""

    def __init__(self, data_root=None, dataset="kitti"):
        assert dataset in ("kitti", "nusc")
        self.data_root, self.dataset = data_root, dataset

    def __call__(self, results):
        imgs, rots, trans, intrins, post_rots, post_trans = results["img_inputs"][:6]
        pts = results["points"]
        dev = pts.device
        gt = create_depth_from_lidar(pts, rots.to(dev), trans.to(dev), intrins.to(dev), post_rots.to(dev),
                                     post_trans.to(dev), imgs.shape[-2:])
        rest = tuple(results["img_inputs"][7:])
        results["img_inputs"] = (imgs, rots, trans, intrins, post_rots, post_trans, gt) + rest
        return results


def image_post_homography(resize, crop, flip, rotate_deg):
    """loading_nusc_imgs.py:35-55: post_rot [2, 2] / post_tran [2] of one augmentation draw
    (resize scalar, crop = (x0, y0, x1, y1), flip bool, rotate in degrees), float32 as the reference computes it."""
    post_rot = torch.eye(2) * resize
    post_tran = -torch.tensor([float(crop[0]), float(crop[1])])
    if flip:
        A = torch.tensor([[-1.0, 0.0], [0.0, 1.0]])
        b = torch.tensor([float(crop[2] - crop[0]), 0.0])
        post_rot = A.matmul(post_rot)
        post_tran = A.matmul(post_tran) + b
    h = rotate_deg / 180 * math.pi
    A = torch.tensor([[math.cos(h), math.sin(h)], [-math.sin(h), math.cos(h)]], dtype=torch.float32)
    b = torch.tensor([float(crop[2] - crop[0]), float(crop[3] - crop[1])]) / 2
    b = A.matmul(-b) + b
    return A.matmul(post_rot), A.matmul(post_tran) + b


# ------------------------------------------------------------------------------------------ img_inputs producer
_PIL_BITS = 22
_COEFF_CACHE = {}


def _pil_bicubic_tables(in_size, out_size, device):
    """Pillow's antialiased bicubic taps for one axis (Resample.c precompute_coeffs + normalize_coeffs_8bpc), vectorised:
    -> (bounds int32 [out, 2], kk int32 [out, ksize]) on ``device``, cached per (sizes, device)"""
    key = (int(in_size), int(out_size), str(device))
    hit = _COEFF_CACHE.get(key)
    if hit is not None:
        return hit
    import numpy as np
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    centers = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    first = np.maximum((centers - support + 0.5).astype(np.int64), 0)          # C's (int) truncation: operands >= 0
    first = np.where(centers - support + 0.5 < 0, 0, first)
    last = np.minimum((centers + support + 0.5).astype(np.int64), in_size)
    count = last - first
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    x = np.abs((taps + first[:, None] - centers[:, None] + 0.5) / fscale)
    a = -0.5
    w = np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0,
                 np.where(x < 2.0, (((x - 5.0) * x + 8.0) * x - 4.0) * a, 0.0))
    w = np.where(taps < count[:, None], w, 0.0)
    # Pillow accumulates the weights left to right in double; numpy's cumulative sum along the row does the same
    tot = np.cumsum(w, axis=1)[:, -1:]
    w = np.where(tot != 0.0, w / np.where(tot != 0.0, tot, 1.0), w)
    q = np.where(w < 0, -0.5 + w * (1 << _PIL_BITS), 0.5 + w * (1 << _PIL_BITS)).astype(np.int64)    # (int) truncates
    bounds = torch.from_numpy(np.stack((first, count), 1).astype(np.int32)).to(device)
    kk = torch.from_numpy(q.astype(np.int32)).to(device)
    _COEFF_CACHE[key] = (bounds, kk)
    return bounds, kk


def image_resize(img, out_w, out_h):
    """``PIL.Image.resize((out_w, out_h))`` (default BICUBIC, antialiased) of a uint8 [H, W, C] frame on the device:
    horizontal pass, uint8 intermediate, vertical pass -- bit for bit (tests compare with Pillow)"""
    ops = get_ops()
    H, W, _ = img.shape
    if out_w != W:
        img = ops.image_resample(img.contiguous(), *_pil_bicubic_tables(W, out_w, img.device), out_w, False)
    if out_h != H:
        img = ops.image_resample(img.contiguous(), *_pil_bicubic_tables(H, out_h, img.device), out_h, True)
    return img


def _pil_rotate_fixed(w, h, angle):
    """``Image.rotate(angle)``'s affine matrix (centre of the frame, cos / sin rounded to 15 digits) in the 16.16 fixed
    point of Geometry.c's affine_fixed -> (mode, a[6]); mode 0 = identity, 1 = 180 degrees (transpose fast paths)"""
    angle = angle % 360.0
    if angle == 0:
        return 0, None
    if angle == 180:
        return 1, None
    cx, cy = w / 2.0, h / 2.0
    t = -math.radians(angle)
    c, s_ = round(math.cos(t), 15), round(math.sin(t), 15)
    m = [c, s_, 0.0, -s_ if s_ != 0 else round(-math.sin(t), 15), c, 0.0]
    m[3] = round(-math.sin(t), 15)
    m[2] = m[0] * -cx + m[1] * -cy + cx
    m[5] = m[3] * -cx + m[4] * -cy + cy
    fx = lambda v: int(math.floor(v * 65536.0 + 0.5))            # noqa: E731
    return 2, [fx(m[0]), fx(m[1]), fx(m[2] + 0.5 * m[0] + 0.5 * m[1]), fx(m[3]), fx(m[4]), fx(m[5] + 0.5 * m[3] + 0.5 * m[4])]


IMG_NORM_DEFAULT = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def image_transform(img, resize_dims, crop, flip, rotate, img_norm_cfg=None, want_canvas=False):
    """loading_nusc_imgs.py:57-64 + :179-193 on the device: decoded uint8 [H, W, 3] frame (channel order as mmcv.imread
    delivers it: BGR) -> (float32 [3, fH, fW] normalised image, uint8 [fH, fW, 3] canvas or None)"""
    import numpy as np
    cfg = img_norm_cfg or IMG_NORM_DEFAULT
    x0, y0, x1, y1 = (int(v) for v in crop)
    fW, fH = x1 - x0, y1 - y0
    r = image_resize(img, int(resize_dims[0]), int(resize_dims[1]))
    mode, aff = _pil_rotate_fixed(fW, fH, float(rotate))
    mean = np.asarray(cfg["mean"], np.float64).astype(np.float32)
    stdinv = (1.0 / np.asarray(cfg["std"], np.float64)).astype(np.float32)
    return get_ops().image_crop_rotate_normalize(r.contiguous(), (x0, y0), (fW, fH), bool(flip), mode, aff, mean, stdinv,
                                                 bool(cfg.get("to_rgb", True)), want_canvas)


@PIPELINES.register_module()
class LoadMultiViewImageFromFiles_OccFormer:
    """loading_nusc_imgs.py:9-177 with the image work on the device.  ``results['curr']['cams'][name]`` carries
    ``cam_intrinsic`` and either ``img`` (a decoded uint8 [H, W, 3] BGR array / tensor, what mmcv.imread returns) or
    ``data_path`` (decoded here with PIL: file I/O stays on the host); ``results['lidar2cam_dic'][name]`` the 4x4
    lidar -> camera matrix.  The augmentation draws come from numpy's global RNG in the reference's order (resize, crop_h,
    crop_w, flip, rotate per camera), so a seeded run samples the same augmentations."""

    def __init__(self, data_config, is_train=False, img_norm_cfg=None, device=None):
        self.is_train, self.data_config, self.img_norm_cfg, self.device = is_train, data_config, img_norm_cfg, device

    def choose_cams(self):
        import numpy as np
        if self.is_train and self.data_config["Ncams"] < len(self.data_config["cams"]):
            return np.random.choice(self.data_config["cams"], self.data_config["Ncams"], replace=False)
        return self.data_config["cams"]

    def sample_augmentation(self, H, W, flip=None, scale=None):
        import numpy as np
        fH, fW = self.data_config["input_size"]
        rs = float(fW) / float(W)
        if self.is_train:
            rs += np.random.uniform(*self.data_config["resize"])
            dims = (int(W * rs), int(H * rs))
            crop_h = int((1 - np.random.uniform(*self.data_config["crop_h"])) * dims[1]) - fH
            crop_w = int(np.random.uniform(0, max(0, dims[0] - fW)))
            flip = self.data_config["flip"] and np.random.choice([0, 1])
            rotate = np.random.uniform(*self.data_config["rot"])
        else:
            rs += self.data_config.get("resize_test", 0.0)
            if scale is not None:
                rs = scale
            dims = (int(W * rs), int(H * rs))
            crop_h = int((1 - np.mean(self.data_config["crop_h"])) * dims[1]) - fH
            crop_w = int(max(0, dims[0] - fW) / 2)
            flip = False if flip is None else flip
            rotate = 0
        return rs, dims, (crop_w, crop_h, crop_w + fW, crop_h + fH), flip, rotate

    def _frame(self, cam_data, device):
        import numpy as np
        img = cam_data.get("img")
        if img is None:
            from PIL import Image
            img = np.asarray(Image.open(cam_data["data_path"]))
            if img.ndim == 3 and img.shape[2] >= 3:
                img = img[:, :, 2::-1]                               # RGB(A) file order -> the BGR of mmcv.imread
        if not torch.is_tensor(img):
            img = torch.from_numpy(np.ascontiguousarray(img))
        return img.to(device=device, dtype=torch.uint8).contiguous()

    def get_inputs(self, results, flip=None, scale=None):
        dev = self.device or results.get("device") or "cuda"
        names = self.choose_cams()
        results["cam_names"] = names
        imgs, rots, trans, intrins, post_rots, post_trans, s2s, canvas = [], [], [], [], [], [], [], []
        for name in names:
            cam = results["curr"]["cams"][name]
            frame = self._frame(cam, dev)
            sensor2lidar = torch.tensor(results["lidar2cam_dic"][name]).inverse().float()
            rs, dims, crop, flip_, rot = self.sample_augmentation(frame.shape[0], frame.shape[1], flip=flip, scale=scale)
            flip = flip_                          # (the reference rebinds its argument: later cameras inherit the draw)
            x, cv = image_transform(frame, dims, crop, flip_, rot, self.img_norm_cfg, want_canvas=True)
            pr2, pt2 = image_post_homography(rs, crop, flip_, rot)
            pr, pt = torch.eye(3), torch.zeros(3)
            pr[:2, :2], pt[:2] = pr2, pt2
            imgs.append(x)
            canvas.append(cv)
            intrins.append(torch.Tensor(cam["cam_intrinsic"]))
            rots.append(sensor2lidar[:3, :3])
            trans.append(sensor2lidar[:3, 3])
            post_rots.append(pr)
            post_trans.append(pt)
            s2s.append(sensor2lidar)
        results["canvas"] = torch.stack(canvas)
        host = lambda ts: torch.stack(ts).to(dev)                   # noqa: E731
        return (torch.stack(imgs), host(rots), host(trans), host(intrins), host(post_rots), host(post_trans),
                torch.zeros((len(names), 1), device=dev), host(s2s))

    def __call__(self, results):
        results["img_inputs"] = self.get_inputs(results)
        return results


@PIPELINES.register_module()
class LoadMultiViewImageFromFiles_SemanticKitti(LoadMultiViewImageFromFiles_OccFormer):
    """loading_kitti_imgs.py:11-145, the monocular ``img_inputs`` producer of the SemanticKITTI configs: the same
    augmentation draws and device image path as the nuScenes loader, one frame.  ``results['img']`` (a list with the one
    decoded uint8 [H, W, 3] frame, as mmcv.imread(..., 'unchanged') returns it) or ``results['img_filename']`` (decoded
    with PIL here); ``results['cam_intrinsic'][0]`` and ``results['lidar2cam'][0]`` as in the reference."""

    def get_inputs(self, results, flip=None, scale=None):
        dev = self.device or results.get("device") or "cuda"
        names = results["img_filename"]
        assert len(names) == 1
        frames = results.get("img")
        frame = self._frame(dict(img=None if frames is None else frames[0], data_path=names[0]), dev)
        results["raw_img"] = frame
        rs, dims, crop, flip, rot = self.sample_augmentation(frame.shape[0], frame.shape[1], flip=flip, scale=scale)
        x, cv = image_transform(frame, dims, crop, flip, rot, self.img_norm_cfg, want_canvas=True)
        pr2, pt2 = image_post_homography(rs, crop, flip, rot)
        pr, pt = torch.eye(3), torch.zeros(3)
        pr[:2, :2], pt[:2] = pr2, pt2
        cam2lidar = torch.Tensor(results["lidar2cam"][0]).inverse()
        results["canvas"] = cv[None]
        host = [cam2lidar[:3, :3], cam2lidar[:3, 3], torch.Tensor(results["cam_intrinsic"][0]), pr, pt, torch.zeros(1)]
        rots, trans, intrin, pr, pt, depth = (t[None].to(dev) for t in host)
        return x[None], rots, trans, intrin, pr, pt, depth, cam2lidar[None].to(dev)


# --------------------------------------------------------------------------- occupancy ground truth (gt_occ, bda_rot)
def sample_bda_augmentation(conf):
    """loading_nusc_occ.py:47-57 / loading_kitti_occ.py:17-26: rot, scale, flip_dx, flip_dy, flip_dz from numpy's
    global RNG in the reference's order (the nuScenes configs carry no flip_dz_ratio: 0)"""
    import numpy as np
    rot = np.random.uniform(*conf["rot_lim"])
    scale = np.random.uniform(*conf["scale_lim"])
    fx = np.random.uniform() < conf["flip_dx_ratio"]
    fy = np.random.uniform() < conf["flip_dy_ratio"]
    fz = np.random.uniform() < conf.get("flip_dz_ratio", 0.0)
    return rot, scale, fx, fy, fz


def bda_matrix(rotate_deg, flip_dx, flip_dy, flip_dz, center=None):
    """The BEV-augmentation matrix ``bda_rot`` (``img_inputs[6]``), host float32 as the reference builds it:
    flip @ rot [3, 3] (loading_nusc_occ.py:147-180) or, about ``center``, denorm @ flip @ rot @ norm [4, 4]
    (loading_kitti_occ.py:59-101).  The scale draw is unused there as well."""
    a = torch.tensor(rotate_deg / 180 * math.pi)
    s, c = torch.sin(a), torch.cos(a)
    rot = torch.Tensor([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    flip = torch.eye(4)
    for on, axis in ((flip_dx, 0), (flip_dy, 1), (flip_dz, 2)):
        if on:
            f = torch.eye(4)
            f[axis, axis] = -1
            flip = flip @ f
    if center is None:
        return (flip @ rot)[:3, :3]
    norm, denorm = torch.eye(4), torch.eye(4)
    norm[:3, -1] = -torch.as_tensor(center, dtype=torch.float32)
    denorm[:3, -1] = torch.as_tensor(center, dtype=torch.float32)
    return denorm @ flip @ rot @ norm


def _device_only(t, what):
    """the tensor programs below are part of the device pipeline: like the kernels (``ops._ptr``) they refuse host tensors
    under the product binding (the test-only host emulation binding runs them on the CPU)"""
    if get_ops().strict and not t.is_cuda:
        from .ops import OccfError
        raise OccfError(f"{what}: GPU tensors expected (no CPU path exists)")


def rotate_label_volume(vox, angle, fill=255):
    """custom_rotate_3d (loading_nusc_occ.py:205-224): every [X, Y] height slice of the uint8 label volume rotated as
    ``Image.rotate(angle, NEAREST, fillcolor=255)`` does (16.16 fixed-point source index; the transpose fast paths for
    0 / 180 and, on square slices, 90 / 270 degrees), all slices at once on the device"""
    _device_only(vox, "rotate_label_volume")
    H, W, Z = vox.shape
    ang = angle % 360.0
    if ang in (90, 270) and H == W:
        return torch.rot90(vox, 1 if ang == 90 else 3, dims=(0, 1)).contiguous()
    mode, a = _pil_rotate_fixed(W, H, angle)
    if mode == 0:
        return vox.clone()
    if mode == 1:
        return vox.flip(0, 1)
    ys = torch.arange(H, device=vox.device, dtype=torch.int64)[:, None]
    xs = torch.arange(W, device=vox.device, dtype=torch.int64)[None, :]
    xin = (a[2] + a[1] * ys + a[0] * xs) >> 16
    yin = (a[5] + a[4] * ys + a[3] * xs) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    src = (yin.clamp(0, H - 1) * W + xin.clamp(0, W - 1)).reshape(-1)
    out = vox.reshape(H * W, Z).index_select(0, src).reshape(H, W, Z)
    return torch.where(ok[..., None], out, torch.full_like(out, fill))


def voxel_transform(vox, rotate_deg, flip_dx, flip_dy, flip_dz, center=None):
    """voxel_transform of both loaders: (int64 label volume rotated unless the angle is ~0, then flipped along z, y, x;
    the bda matrix).  ``vox`` None -> only the matrix (the nuScenes loader augments the points instead)."""
    mat = bda_matrix(rotate_deg, flip_dx, flip_dy, flip_dz, center)
    if vox is None:
        return None, mat
    _device_only(vox, "voxel_transform")
    vox = vox.to(torch.uint8)
    if abs(rotate_deg) > 1e-8:                                       # np.isclose(rotate_degree, 0)
        vox = rotate_label_volume(vox, rotate_deg)
    dims = [d for d, on in ((2, flip_dz), (1, flip_dy), (0, flip_dx)) if on]
    if dims:
        vox = vox.flip(*dims)
    return vox.long(), mat


def voxelize_point_labels(points, labels, grid_size, pc_range, num_labels, empty_id=17):
    """loading_nusc_occ.py:99-121 on the device: points float32 [P, 3] (augmented), labels int64 [P] in [0, num_labels)
    -> int64 [X, Y, Z]: the label most points of a voxel carry (the smallest on a tie, as np.argmax of
    nb_process_label's counter; the counter is uint16 there and wraps here too), ``empty_id`` where no point falls;
    then 0 (noise) -> 255, ``empty_id`` -> 0.  The grid index is computed in float64 like the reference's numpy
    (points promoted, clip to [lo, hi - 1e-5], floor of the quotient).  No host synchronisation."""
    import numpy as np
    _device_only(points, "voxelize_point_labels")
    dev = points.device
    gs = [int(v) for v in grid_size]
    rng = np.array(pc_range, np.float64)
    vs = (rng[3:] - rng[:3]) / np.array(gs)
    lo = torch.tensor(rng[:3], dtype=torch.float64, device=dev)
    hi = torch.tensor(rng[3:] - 1e-5, dtype=torch.float64, device=dev)
    ind = torch.floor((torch.minimum(torch.maximum(points.double(), lo), hi) - lo)
                      / torch.tensor(vs, dtype=torch.float64, device=dev)).long()
    V, L = gs[0] * gs[1] * gs[2], int(num_labels)
    lin = (ind[:, 0] * gs[1] + ind[:, 1]) * gs[2] + ind[:, 2]
    P = int(points.shape[0])
    if P == 0:
        return torch.zeros(gs, dtype=torch.int64, device=dev)
    # per (voxel, label) run of the sorted keys its length (running maxima of the run-start / run-end positions: static
    # shapes, no histogram of [voxels x labels] -- 1.5 GB at 512 x 512 x 40 x 18), then per voxel the entry with the
    # largest (count, -label): a second sort puts it first in the voxel's run; everything else goes to a dump slot
    key = torch.sort(lin * L + labels).values
    pos = torch.arange(P, device=dev)
    edge = torch.ones(P + 1, dtype=torch.bool, device=dev)
    edge[1:P] = key[1:] != key[:-1]                                   # edge[i]: a run starts at i; edge[i + 1]: one ends at i
    run_first = torch.cummax(torch.where(edge[:P], pos, torch.zeros_like(pos)), 0).values
    run_last = (P - 1) - torch.cummax(torch.where(edge[1:].flip(0), pos, torch.zeros_like(pos)), 0).values.flip(0)
    cnt = (run_last - run_first + 1) & 0xFFFF
    big = 65536 * L + L
    key2 = torch.sort((key // L) * big + (big - 1 - (cnt * L + (L - 1 - key % L)))).values
    lin2 = key2 // big
    lab = (L - 1) - (big - 1 - key2 % big) % L
    first = torch.ones_like(lin2, dtype=torch.bool)
    first[1:] = lin2[1:] != lin2[:-1]
    lab = torch.where(lab == 0, torch.full_like(lab, 255), lab)
    lab = torch.where(lab == empty_id, torch.zeros_like(lab), lab)
    out = torch.zeros(V + 1, dtype=torch.int64, device=dev)
    out.scatter_(0, torch.where(first, lin2, torch.full_like(lin2, V)), lab)
    return out[:V].view(gs)


@PIPELINES.register_module()
class LoadNuscOccupancyAnnotations:
    """loading_nusc_occ.py:13-125 (registered under the reference's pipeline name): lidarseg points -> ``gt_occ``
    (int64 [X, Y, Z]), ``points_occ`` (float32 [P, 4]) and ``bda_rot`` inserted as ``img_inputs[6]``.  The decoded arrays
    come in ``results['points']`` (float32 [P, >=3]) and ``results['points_label']`` (raw uint8 lidarseg labels [P]), or
    are read from ``results['pts_filename']`` / ``data_root + results['lidarseg']`` as the reference does (file I/O on the
    host); learning map, BEV augmentation of the points and the voxelisation run on the device."""

    def __init__(self, data_root="data/nuscenes", is_train=False, is_test_submit=False, grid_size=None,
                 point_cloud_range=None, bda_aug_conf=None, unoccupied_id=17, cls_metas="nuscenes.yaml", device=None):
        import numpy as np
        import yaml
        self.is_train, self.is_test_submit, self.data_root = is_train, is_test_submit, data_root
        if isinstance(cls_metas, dict):
            self.learning_map = cls_metas.get("learning_map", cls_metas)
        else:
            with open(cls_metas, "r") as stream:
                self.learning_map = yaml.safe_load(stream)["learning_map"]
        lut = np.zeros(256, np.int64)
        for k, v in self.learning_map.items():
            lut[int(k)] = int(v)
        self._lut_host, self._lut = lut, {}
        self.num_labels = max(int(lut.max()), int(unoccupied_id)) + 1
        self.bda_aug_conf, self.unoccupied_id, self.device = bda_aug_conf, unoccupied_id, device
        self.grid_size = [int(v) for v in grid_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]

    def sample_3d_augmentation(self):
        return sample_bda_augmentation(self.bda_aug_conf)

    def _points(self, results, dev):
        import numpy as np
        pts = results.get("points")
        if pts is None:
            pts = np.fromfile(results["pts_filename"], dtype=np.float32, count=-1).reshape(-1, 5)
        if not torch.is_tensor(pts):
            pts = torch.from_numpy(np.ascontiguousarray(pts))
        return pts.to(device=dev, dtype=torch.float32)[:, :3]

    def __call__(self, results):
        import os

        import numpy as np
        dev = self.device or results.get("device") or results["img_inputs"][0].device
        head, tail = tuple(results["img_inputs"][:6]), tuple(results["img_inputs"][6:])
        points = self._points(results, dev)
        if self.is_test_submit:
            results["img_inputs"] = head + (torch.eye(3, device=dev),) + tail
            results["points_occ"] = torch.cat([points, points.new_zeros(points.shape[0], 1)], 1)
            return results
        raw = results.get("points_label")
        if raw is None:
            raw = np.fromfile(os.path.join(self.data_root, results["lidarseg"]), dtype=np.uint8)
        if not torch.is_tensor(raw):
            raw = torch.from_numpy(np.ascontiguousarray(raw))
        if dev not in self._lut:
            self._lut[dev] = torch.from_numpy(self._lut_host).to(dev)
        labels = self._lut[dev][raw.to(dev).reshape(-1).long()]
        if self.is_train:
            rot, _, fx, fy, fz = self.sample_3d_augmentation()
            bda = bda_matrix(rot, fx, fy, fz)
        else:
            bda = torch.eye(3)
        bda = bda.to(dev)
        points = points @ bda.t()
        results["gt_occ"] = voxelize_point_labels(points, labels, self.grid_size, self.point_cloud_range, self.num_labels,
                                                  self.unoccupied_id)
        results["points_occ"] = torch.cat([points, labels[:, None].float()], 1)
        results["img_inputs"] = head + (bda,) + tail
        return results


@PIPELINES.register_module()
class LoadSemKittiAnnotation:
    """loading_kitti_occ.py:7-55 (registered under the reference's pipeline name): the voxel labels the dataset loaded
    (``results['gt_occ']``, [X, Y, Z]) through the BEV augmentation on the device, ``bda_rot`` [4, 4] (about the centre
    of the point-cloud range) inserted as ``img_inputs[6]``; ``gt_occ`` None = the test split."""

    def __init__(self, bda_aug_conf, is_train=True, point_cloud_range=(0, -25.6, -2, 51.2, 25.6, 4.4), device=None):
        self.bda_aug_conf, self.is_train, self.device = bda_aug_conf, is_train, device
        rng = torch.tensor(point_cloud_range)
        self.point_cloud_range = rng
        self.transform_center = (rng[:3] + rng[3:]) / 2

    def sample_bda_augmentation(self):
        return sample_bda_augmentation(self.bda_aug_conf)

    def __call__(self, results):
        dev = self.device or results.get("device") or results["img_inputs"][0].device
        head, tail = tuple(results["img_inputs"][:6]), tuple(results["img_inputs"][6:])
        gt = results["gt_occ"]
        if gt is None:
            results["img_inputs"] = head + (torch.eye(4, device=dev),) + tail
            return results
        gt = torch.as_tensor(gt).to(dev)
        if self.is_train:
            rot, _, fx, fy, fz = self.sample_bda_augmentation()
            gt, bda = voxel_transform(gt, rot, fx, fy, fz, self.transform_center)
        else:
            bda = torch.eye(4)
        results["img_inputs"] = head + (bda.to(dev),) + tail
        results["gt_occ"] = gt.long()
        return results


@PIPELINES.register_module()
class OccDefaultFormatBundle3D:
    """formating.py:7-45: ``gt_occ`` / ``points_occ`` / ``points_uv`` become tensors.  The reference wraps them in mmcv
    DataContainers for its multi-worker collate; here one process per GPU feeds its own model and ``collate`` below
    applies the same stacking rules (gt_occ stacked, points lists kept per sample) without the wrapper."""

    def __init__(self, class_names=None, with_gt=True, with_label=True):
        self.class_names, self.with_gt, self.with_label = class_names, with_gt, with_label

    def __call__(self, results):
        gt = results.get("gt_occ")
        if gt is not None:
            results["gt_occ"] = tuple(torch.as_tensor(x) for x in gt) if isinstance(gt, list) else torch.as_tensor(gt)
        for key in ("points_occ", "points_uv"):
            if key in results:
                results[key] = torch.as_tensor(results[key])
        return results


@PIPELINES.register_module()
class Collect3D:
    """mmdet3d's Collect3D as the configs use it: ``img_metas`` = the ``meta_keys`` present in ``results``, plus ``keys``"""

    def __init__(self, keys, meta_keys=("pc_range", "occ_size")):
        self.keys, self.meta_keys = keys, meta_keys

    def __call__(self, results):
        data = dict(img_metas={k: results[k] for k in self.meta_keys if k in results})
        for key in self.keys:
            data[key] = results[key]
        return data


class Compose:
    """the configs' ``train_pipeline`` / ``test_pipeline`` lists: every stage built from PIPELINES by its ``type``;
    ``defaults`` (e.g. ``device``) is handed to the stages whose constructor takes it"""

    def __init__(self, stages, **defaults):
        import inspect
        self.stages = []
        for cfg in stages:
            if isinstance(cfg, dict):
                cfg = dict(cfg)
                name = cfg.pop("type")
                cls = PIPELINES.get(name)
                if cls is None:
                    raise KeyError(f"pipeline stage {name!r} is not registered")
                accepted = inspect.signature(cls.__init__).parameters
                cfg.update({k: v for k, v in defaults.items() if k in accepted and k not in cfg})
                self.stages.append(cls(**cfg))
            else:
                self.stages.append(cfg)

    def __call__(self, results):
        for stage in self.stages:
            results = stage(results)
            if results is None:
                return None
        return results


def collate(samples):
    """The batch a model call takes from per-sample pipeline outputs, with mmcv collate's rules for these keys: tensors
    and the ``img_inputs`` tuple stacked along a new batch axis, ``points_occ`` / ``points_uv`` kept as per-sample lists,
    ``img_metas`` a list of dicts"""
    out = {}
    for key, first in samples[0].items():
        vals = [s[key] for s in samples]
        if key == "img_metas" or key in ("points_occ", "points_uv") or first is None:
            out[key] = vals if first is not None else None
        elif isinstance(first, (tuple, list)):
            out[key] = [torch.stack([v[i] for v in vals]) for i in range(len(first))]
        else:
            out[key] = torch.stack(vals)
    return out


SEMANTIC_KITTI_CLASS_NAMES = ["unlabeled", "car", "bicycle", "motorcycle", "truck", "other-vehicle", "person", "bicyclist",
                              "motorcyclist", "road", "parking", "sidewalk", "other-ground", "building", "fence",
                              "vegetation", "trunk", "terrain", "pole", "traffic-sign"]


class SSCMetrics:
    """ssc_metric.py:13-106.  ``update(y_pred, y_true, nonempty, nonsurface)``: y_pred int64 labels [B, X, Y, Z] (or
    ``scores=`` the class volume [B, C, X, Y, Z], arg-max taken inside the kernel -- apis/test.py:64), y_true labels
    with 255 = ignore.  Unlike the reference the inputs are NOT modified in place (its in-place edits are reproduced
    in the counting rule, csrc/pipeline.hip)."""

    def __init__(self, class_names=None, device=None):
        self.class_names = list(class_names) if class_names is not None else list(SEMANTIC_KITTI_CLASS_NAMES)
        self.n_classes = len(self.class_names)
        self.device = device
        self.counts = None

    def _state(self, device):
        if self.counts is None:
            self.counts = torch.zeros(self.n_classes * self.n_classes + 3, dtype=torch.int64, device=device)
        return self.counts

    def reset(self):
        self.counts = None

    def _count(self, counts, y_pred, y_true, nonempty, nonsurface, scores):
        return get_ops().ssc_confusion(counts, y_true, pred=y_pred, scores=scores, nonempty=nonempty,
                                       nonsurface=nonsurface, num_classes=self.n_classes)

    def update(self, y_pred=None, y_true=None, nonempty=None, nonsurface=None, scores=None):
        self._count(self._state(y_true.device), y_pred, y_true, nonempty, nonsurface, scores)

    @staticmethod
    def _scores(counts, C):
        conf = counts[:C * C].view(C, C)
        tps = torch.diagonal(conf)
        fps = conf.sum(0) - tps
        fns = conf.sum(1) - tps
        return counts[C * C], counts[C * C + 1], counts[C * C + 2], tps, fps, fns

    def compute_single(self, y_pred=None, y_true=None, nonempty=None, nonsurface=None, scores=None):
        """ssc_metric.py:41-60: the six count arrays of ONE sample as numpy (tp, fp, fn, tp_sum, fp_sum, fn_sum)"""
        c = torch.zeros(self.n_classes * self.n_classes + 3, dtype=torch.int64, device=y_true.device)
        self._count(c, y_pred, y_true, nonempty, nonsurface, scores)
        tp, fp, fn, tps, fps, fns = self._scores(c.cpu(), self.n_classes)
        return (tp.numpy(), fp.numpy(), fn.numpy(), tps.float().numpy(), fps.float().numpy(), fns.float().numpy())

    def reduce(self, dist=None):
        """sum the state over the ranks (torchmetrics' dist_reduce_fx='sum'); RCCL all-reduce of C*C + 3 int64"""
        if dist is not None and self.counts is not None:
            dist.all_reduce(self.counts)
        return self

    def compute(self):
        """ssc_metric.py:86-101"""
        C = self.n_classes
        tp, fp, fn, tps, fps, fns = (t.double() for t in self._scores(self.counts, C))
        precision = tp / (tp + fp)
        recall = tp / (tp + fn)
        iou = tp / (tp + fp + fn)
        iou_ssc = tps / (tps + fps + fns + 1e-5)
        return {"precision": precision.float().reshape(1), "recall": recall.float().reshape(1), "iou": float(iou),
                "iou_ssc": iou_ssc.float(), "iou_ssc_mean": float(iou_ssc[1:].mean())}
