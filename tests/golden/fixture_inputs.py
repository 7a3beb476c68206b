"""Inputs of the committed golden fixtures, regenerated procedurally (no weights or images are shipped).

tests/golden/make_golden.py fills the REFERENCE model's state_dict and draws its batches with exactly these generators
(oracle/dbnet_oracle.py holds the same generators; tests/test_oracle_golden.py::test_fixture_inputs_equal_the_oracles_generators
holds the two copies together bit for bit for every architecture), so whoever wants to replay a fixture — the -m gpu tests, smoke() and the
parity gate of bench.py — can rebuild its inputs from (seed, shapes) alone.  This module is data generation only: it contains no
arithmetic of the path and imports nothing of oracle/ or of the package.
"""
import math

import torch


def key_seed(key, seed):
    hsh = 1469598103934665603
    for ch in key.encode():
        hsh = ((hsh ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (hsh ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def kind_of(key, state):
    """What a state_dict entry is, from its key and shapes alone (reference key set: src/models.py, modules/*.py)."""
    if key.endswith('num_batches_tracked'):
        return 'bn_nbt'
    if key.endswith('running_mean'):
        return 'bn_rm'
    if key.endswith('running_var'):
        return 'bn_rv'
    if key.startswith(('backbone.fc.', 'backbone.smooth.')):
        return 'dead'
    if 'conv2_offset' in key:
        return 'offset_w' if key.endswith('.weight') else 'offset_b'
    stem = key.rsplit('.', 1)[0]
    w = state[stem + '.weight']
    if w.dim() == 1:
        return 'bn_w' if key.endswith('.weight') else 'bn_b'
    if key.endswith('.bias'):
        return 'conv_b'
    # the head's two ConvTranspose2d layers per branch: segmentation_head.{binarize,thresh}.{3,6}
    if key.startswith('segmentation_head.') and stem.rsplit('.', 1)[1] in ('3', '6'):
        return 'convT_w'
    return 'conv_w'


def procedural_fill(state, seed=0):
    """Deterministic, per-key seeded fill (in place) of any dict keyed like the reference state_dict; trained-net-like scales."""
    for key, t in state.items():
        kind = kind_of(key, state)
        g = torch.Generator().manual_seed(key_seed(key, seed))
        if kind == 'bn_nbt':
            t.fill_(0)
            continue
        shape = tuple(t.shape)
        if kind in ('conv_w', 'convT_w', 'dead'):
            if len(shape) == 4:
                fan = shape[1] * shape[2] * shape[3] if kind != 'convT_w' else shape[0] * shape[2] * shape[3] // 4
            else:
                fan = shape[-1]
            v = torch.randn(shape, generator=g) * math.sqrt(2.0 / max(fan, 1))
        elif kind == 'conv_b':
            v = torch.randn(shape, generator=g) * 0.05
        elif kind == 'offset_w':  # offsets of a fraction of a pixel (the reference initialises them to 0, resnet.py:204-208)
            v = torch.randn(shape, generator=g) * (0.3 / math.sqrt(shape[1] * 9))
        elif kind == 'offset_b':
            v = torch.randn(shape, generator=g) * 0.3
        elif kind == 'bn_w':
            v = 0.5 + torch.rand(shape, generator=g)
        elif kind == 'bn_b':
            v = torch.randn(shape, generator=g) * 0.1
        elif kind == 'bn_rm':
            v = torch.randn(shape, generator=g) * 0.1
        elif kind == 'bn_rv':
            v = 0.5 + torch.rand(shape, generator=g)
        else:
            raise KeyError(kind)
        with torch.no_grad():
            t.copy_(v.to(t.dtype))
    return state


def synthetic_batch(n, size, seed=0, img_scale=1.0):
    """img ~ N(0,1) * scale; binary masks like the real loader (data_loaders.py:158-165); gts stacked in train.py:163-166 order."""
    g = torch.Generator().manual_seed(seed)
    h, w = (size, size) if isinstance(size, int) else size
    img = torch.randn(n, 3, h, w, generator=g) * img_scale
    u = torch.rand(4, n, h, w, generator=g)
    prob_gt = (u[0] > 0.9).float()
    sup_mask = (u[1] > 0.05).float()
    thresh_gt = 0.3 + 0.4 * u[2]
    text_area = (u[3] > 0.8).float()
    return img, torch.stack([prob_gt, sup_mask, thresh_gt, text_area])


def sample_idx(numel, k=256):
    """The strided sample positions the golden summaries keep of a large tensor (make_golden.summarize)."""
    import numpy as np
    if numel <= k:
        return np.arange(numel)
    return (np.arange(k, dtype=np.int64) * (numel // k)) + (numel // (2 * k))
