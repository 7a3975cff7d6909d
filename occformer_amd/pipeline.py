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
  * ``image_post_homography`` -- the calibration half of ``LoadMultiViewImageFromFiles_OccFormer.img_transform``
    (loading_nusc_imgs.py:35-55): the post-rotation / post-translation the view transformer consumes, for a batch of
    augmentation draws.  (JPEG decoding and PIL resampling are I/O: out of scope.)

There is no CPU implementation: host tensors raise in ``occformer_amd.ops``."""
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
