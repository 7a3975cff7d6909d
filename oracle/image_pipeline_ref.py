"""TEST INFRASTRUCTURE (never imported by the product): CPU restatement of the image half of the reference's
``img_inputs`` producer,
    projects/mmdet3d_plugin/datasets/pipelines/loading_nusc_imgs.py
        :57-64   img_transform_core   (PIL: Image.resize -> crop -> FLIP_LEFT_RIGHT -> rotate)
        :66-103  choose_cams / sample_augmentation (numpy global RNG, draw order kept)
        :35-55   img_transform (post-homography; restated in occformer_amd.pipeline.image_post_homography)
        :179-193 mmlabNormalize (mmcv.image.imnormalize: BGR -> RGB, (x - mean) * (1 / std) in float32)
in plain numpy.  The PIL calls are a third-party dependency of the reference (Pillow; not vendored under
/root/reference): restated from Pillow's published algorithm -- src/libImaging/Resample.c (precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal/Vertical_8bpc: bicubic a = -0.5, support scaled by the
down-scaling factor, 22-bit fixed-point coefficients, uint8 intermediate) and Geometry.c (affine_fixed: nearest
neighbour in 16.16 fixed point, zero fill) with Image.rotate's matrix set-up -- and PINNED against Pillow 12.2 itself
(installed in the build container: tests/test_pipeline_ops.py::test_image_oracle_is_pillow compares on random frames,
bit for bit) and against the reference's own ``img_transform_core`` / ``sample_augmentation`` imported through
tests/refshim (tests/golden/make_golden_image_pipeline.py -> tests/golden/image_pipeline.npz).
mmcv is not installed: ``normalize`` restates imnormalize_ (cv2.subtract / cv2.multiply on a float32 image with the
scalars converted to float32) -- that last-ulp behaviour is NOT pinned (tests allow 1e-6 relative)."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (bounds int32 [out, 2], kk int32 [out, ksize])"""
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = 2.0 * fs
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / fs
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize(img, out_w, out_h):
    """PIL Image.resize((out_w, out_h)) of a uint8 [H, W, C] frame (default resample: BICUBIC)"""
    H, W, C = img.shape
    cur = img
    if out_w != W:
        b, k = resample_coeffs(W, out_w)
        out = np.zeros((H, out_w, C), np.uint8)
        for xx in range(out_w):
            acc = np.full((H, C), 1 << (PRECISION_BITS - 1), np.int64)
            for x in range(b[xx, 1]):
                acc += cur[:, b[xx, 0] + x, :].astype(np.int64) * int(k[xx, x])
            out[:, xx, :] = _clip8(acc)
        cur = out
    if out_h != H:
        b, k = resample_coeffs(H, out_h)
        out = np.zeros((out_h, cur.shape[1], C), np.uint8)
        for yy in range(out_h):
            acc = np.full((cur.shape[1], C), 1 << (PRECISION_BITS - 1), np.int64)
            for y in range(b[yy, 1]):
                acc += cur[b[yy, 0] + y].astype(np.int64) * int(k[yy, y])
            out[yy] = _clip8(acc)
        cur = out
    return cur if cur is not img else img.copy()


def crop(img, box):
    """PIL Image.crop((x0, y0, x1, y1)): zero outside the frame"""
    x0, y0, x1, y1 = box
    H, W, C = img.shape
    out = np.zeros((y1 - y0, x1 - x0, C), img.dtype)
    sx0, sy0, sx1, sy1 = max(x0, 0), max(y0, 0), min(x1, W), min(y1, H)
    if sx1 > sx0 and sy1 > sy0:
        out[sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = img[sy0:sy1, sx0:sx1]
    return out


def rotate_affine(w, h, angle):
    """Image.rotate's set-up: (mode, 16.16 fixed-point a0..a5): 0 = identity, 1 = 180 degrees, 2 = affine_fixed"""
    angle = angle % 360.0
    if angle == 0:
        return 0, [0] * 6
    if angle == 180:
        return 1, [0] * 6
    rc = (w / 2.0, h / 2.0)
    ang = -math.radians(angle)
    m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0, round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
    m[2] = m[0] * -rc[0] + m[1] * -rc[1] + m[2]
    m[5] = m[3] * -rc[0] + m[4] * -rc[1] + m[5]
    m[2] += rc[0]
    m[5] += rc[1]
    fix = lambda v: int(math.floor(v * 65536.0 + 0.5))          # noqa: E731
    return 2, [fix(m[0]), fix(m[1]), fix(m[2] + m[0] * 0.5 + m[1] * 0.5), fix(m[3]), fix(m[4]),
               fix(m[5] + m[3] * 0.5 + m[4] * 0.5)]


def rotate(img, angle):
    """PIL Image.rotate(angle) (NEAREST, no expand, zero fill)"""
    H, W, C = img.shape
    mode, a = rotate_affine(W, H, angle)
    if mode == 0:
        return img.copy()
    if mode == 1:
        return img[::-1, ::-1].copy()
    ys, xs = np.mgrid[0:H, 0:W].astype(np.int64)
    xin = (a[2] + a[1] * ys + a[0] * xs) >> 16
    yin = (a[5] + a[4] * ys + a[3] * xs) >> 16
    ok = (xin >= 0) & (xin < W) & (yin >= 0) & (yin < H)
    out = np.zeros_like(img)
    out[ok] = img[yin[ok], xin[ok]]
    return out


def img_transform_core(img, resize_dims, crop_box, flip, rot):
    """loading_nusc_imgs.py:57-64 on a uint8 [H, W, 3] array"""
    img = resize(img, *resize_dims)
    img = crop(img, crop_box)
    if flip:
        img = img[:, ::-1].copy()
    return rotate(img, rot)


def normalize(img, mean=(123.675, 116.28, 103.53), std=(58.395, 57.12, 57.375), to_rgb=True):
    """mmlabNormalize (:179-193): uint8 [H, W, 3] -> float32 [3, H, W]"""
    x = img.astype(np.float32)
    if to_rgb:
        x = x[..., ::-1]
    mean32 = np.asarray(mean, np.float64).astype(np.float32)
    stdinv32 = (1.0 / np.asarray(std, np.float64)).astype(np.float32)
    x = (x - mean32) * stdinv32
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def sample_augmentation(H, W, data_config, is_train, flip=None, scale=None, rng=np.random):
    """:76-103, the numpy draws in the reference's order"""
    fH, fW = data_config["input_size"]
    if is_train:
        resize_ = float(fW) / float(W)
        resize_ += rng.uniform(*data_config["resize"])
        resize_dims = (int(W * resize_), int(H * resize_))
        newW, newH = resize_dims
        crop_h = int((1 - rng.uniform(*data_config["crop_h"])) * newH) - fH
        crop_w = int(rng.uniform(0, max(0, newW - fW)))
        crop_box = (crop_w, crop_h, crop_w + fW, crop_h + fH)
        flip = data_config["flip"] and rng.choice([0, 1])
        rot = rng.uniform(*data_config["rot"])
    else:
        resize_ = float(fW) / float(W)
        resize_ += data_config.get("resize_test", 0.0)
        if scale is not None:
            resize_ = scale
        resize_dims = (int(W * resize_), int(H * resize_))
        newW, newH = resize_dims
        crop_h = int((1 - np.mean(data_config["crop_h"])) * newH) - fH
        crop_w = int(max(0, newW - fW) / 2)
        crop_box = (crop_w, crop_h, crop_w + fW, crop_h + fH)
        flip = False if flip is None else flip
        rot = 0
    return resize_, resize_dims, crop_box, flip, rot
