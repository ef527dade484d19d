"""Comparison of a linear render with a picture the reference ships (fixtures of tools/make_golden_refimage.py): energy ratios per
region over the pixels whose 8-bit value was not clipped, and the dispersion-fringe statistic.  Pure numpy; used by the -m gpu tests."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = 8


def load(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name))
    h, w = (int(v) for v in z["shape"])
    valid = np.unpackbits(z["valid"])[:h * w].reshape(h, w).astype(bool)
    return z, valid, z["lin"].astype(np.float64)


def block_sums(img, valid):
    h, w, _ = img.shape
    v = img.astype(np.float64) * valid[..., None]
    return v[:h // B * B, :w // B * B].reshape(h // B, B, w // B, B, 3).sum((1, 3))


def region_ratios(mine_blocks, ref_blocks, regions):
    """regions: name -> boolean mask over BLOCKS.  Returns name -> per-channel (sum mine / sum ref)."""
    return {k: mine_blocks[m].sum(0) / ref_blocks[m].sum(0) for k, m in regions.items()}


def block_mask(shape_px, pred):
    """Boolean mask over the 8x8 blocks of a picture from a predicate on the block centres (x, y in pixels)."""
    h, w = shape_px
    yy, xx = np.mgrid[0:h // B, 0:w // B]
    return pred(xx * B + B / 2, yy * B + B / 2)


def fringe_map(lin):
    from scipy.ndimage import gaussian_filter
    img = gaussian_filter(lin.astype(np.float64), (1.5, 1.5, 0))
    c = (img[..., 0] - img[..., 2]) / (img.sum(2) + 1e-3)
    return c - gaussian_filter(c, 12)
