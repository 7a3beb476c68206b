"""CPU oracle for the DBNet hot path (TEST INFRASTRUCTURE — never on the product path).

A from-scratch functional restatement, in plain CPU PyTorch fp32 ops, of the
reference's model forward, loss stack and per-step update:

  * model assembly / forward ........ /root/reference/src/models.py:34-48
  * ResNet-18 stem + BasicBlock ..... /root/reference/src/modules/resnet.py:70-91,231-242
  * ConvBnRelu ...................... /root/reference/src/modules/basic.py:32-36
  * FPN neck ........................ /root/reference/src/modules/segmentation_body.py:64-87
  * DBHead + step function .......... /root/reference/src/modules/segmentation_head.py:35-45,106-108
  * OHEM-BCE / Dice / L1 / DBLoss ... /root/reference/src/losses.py:18-40,48-66,75-82,105-139
  * step order + Adam ............... /root/reference/src/train.py:110-117,160-172

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file, and only as the checker / reported baseline.  The product
package `db_text_minimal_amd` must never import it.

Pinning: the reference has no tests and no golden vectors, so the
oracle is pinned against outputs of the reference itself, imported in the build
container by `tests/golden/make_golden.py` (fixtures in `tests/golden/*.npz`,
checked by `tests/test_oracle_golden.py`).

The state is a flat dict keyed exactly like the reference's `state_dict()`
(211 entries).  Nothing here reads /root/reference.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

K_STEP = 50.0  # segmentation_head.py:21
BN_EPS = 1e-5
BN_MOMENTUM = 0.1

# ----------------------------------------------------------------------------
# state layout
# ----------------------------------------------------------------------------


def _bn_entries(prefix, c):
    return [
        (prefix + '.weight', (c, ), 'bn_w'),
        (prefix + '.bias', (c, ), 'bn_b'),
        (prefix + '.running_mean', (c, ), 'bn_rm'),
        (prefix + '.running_var', (c, ), 'bn_rv'),
        (prefix + '.num_batches_tracked', (), 'bn_nbt'),
    ]


ARCHS = {  # name -> (block, layers, dcn) — resnet.py:245-306; models.py:8 registers only resnet18 (SURVEY A4')
    'resnet18': ('basic', (2, 2, 2, 2), False),
    'deformable_resnet18': ('basic', (2, 2, 2, 2), True),
    'resnet50': ('bottleneck', (3, 4, 6, 3), False),
    'deformable_resnet50': ('bottleneck', (3, 4, 6, 3), True),
}


def backbone_out_channels(arch='resnet18'):
    e = 1 if ARCHS[arch][0] == 'basic' else 4
    return [64 * e, 128 * e, 256 * e, 512 * e]


def state_spec(arch='resnet18'):
    """[(key, shape, kind)] in the reference's state_dict order."""
    block, layers, dcn = ARCHS[arch]
    exp = 1 if block == 'basic' else 4
    s = []
    s.append(('backbone.conv1.weight', (64, 3, 7, 7), 'conv_w'))
    s += _bn_entries('backbone.bn1', 64)
    inpl = 64
    for li, planes in enumerate([64, 128, 256, 512], start=1):
        with_dcn = dcn and li > 1  # resnet.py:176-191: layer1 is built without dcn
        for bi in range(layers[li - 1]):
            p = 'backbone.layer%d.%d' % (li, bi)
            cin = inpl if bi == 0 else planes * exp
            if block == 'basic':
                s.append((p + '.conv1.weight', (planes, cin, 3, 3), 'conv_w'))
                s += _bn_entries(p + '.bn1', planes)
            else:
                s.append((p + '.conv1.weight', (planes, cin, 1, 1), 'conv_w'))
                s += _bn_entries(p + '.bn1', planes)
            if with_dcn:  # resnet.py:54-65 / 111-124: offsets conv (bias) then torchvision DeformConv2d (no bias)
                s.append((p + '.conv2_offset.weight', (18, planes, 3, 3), 'offset_w'))
                s.append((p + '.conv2_offset.bias', (18, ), 'offset_b'))
            s.append((p + '.conv2.weight', (planes, planes, 3, 3), 'conv_w'))
            s += _bn_entries(p + '.bn2', planes)
            if block == 'bottleneck':
                s.append((p + '.conv3.weight', (planes * 4, planes, 1, 1), 'conv_w'))
                s += _bn_entries(p + '.bn3', planes * 4)
            if bi == 0 and (li > 1 or cin != planes * exp):
                s.append((p + '.downsample.0.weight', (planes * exp, cin, 1, 1), 'conv_w'))
                s += _bn_entries(p + '.downsample.1', planes * exp)
        inpl = planes * exp
    # dead parameters the reference constructs but never uses (resnet.py:192-195)
    s.append(('backbone.fc.weight', (1000, 512 * exp), 'dead'))
    s.append(('backbone.fc.bias', (1000, ), 'dead'))
    s.append(('backbone.smooth.weight', (256, 2048, 1, 1), 'dead'))
    s.append(('backbone.smooth.bias', (256, ), 'dead'))
    b = 'segmentation_body.'
    for name, cin in zip(('reduce_conv_c2', 'reduce_conv_c3', 'reduce_conv_c4', 'reduce_conv_c5'), backbone_out_channels(arch)):
        s.append((b + name + '.conv.weight', (64, cin, 1, 1), 'conv_w'))
        s.append((b + name + '.conv.bias', (64, ), 'conv_b'))
        s += _bn_entries(b + name + '.bn', 64)
    for name in ('smooth_p4', 'smooth_p3', 'smooth_p2'):
        s.append((b + name + '.conv.weight', (64, 64, 3, 3), 'conv_w'))
        s.append((b + name + '.conv.bias', (64, ), 'conv_b'))
        s += _bn_entries(b + name + '.bn', 64)
    s.append((b + 'conv.0.weight', (256, 256, 3, 3), 'conv_w'))
    s.append((b + 'conv.0.bias', (256, ), 'conv_b'))
    s += _bn_entries(b + 'conv.1', 256)
    h = 'segmentation_head.'
    s.append((h + 'binarize.0.weight', (64, 256, 3, 3), 'conv_w'))
    s.append((h + 'binarize.0.bias', (64, ), 'conv_b'))
    s += _bn_entries(h + 'binarize.1', 64)
    s.append((h + 'binarize.3.weight', (64, 64, 2, 2), 'convT_w'))
    s.append((h + 'binarize.3.bias', (64, ), 'conv_b'))
    s += _bn_entries(h + 'binarize.4', 64)
    s.append((h + 'binarize.6.weight', (64, 1, 2, 2), 'convT_w'))
    s.append((h + 'binarize.6.bias', (1, ), 'conv_b'))
    s.append((h + 'thresh.0.weight', (64, 256, 3, 3), 'conv_w'))  # bias=False (segmentation_head.py:64-68)
    s += _bn_entries(h + 'thresh.1', 64)
    s.append((h + 'thresh.3.weight', (64, 64, 2, 2), 'convT_w'))
    s.append((h + 'thresh.3.bias', (64, ), 'conv_b'))
    s += _bn_entries(h + 'thresh.4', 64)
    s.append((h + 'thresh.6.weight', (64, 1, 2, 2), 'convT_w'))
    s.append((h + 'thresh.6.bias', (1, ), 'conv_b'))
    return s


def _key_seed(key, seed):
    hsh = 1469598103934665603
    for ch in key.encode():
        hsh = ((hsh ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (hsh ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def procedural_fill(state, seed=0):
    """Deterministic, per-key seeded fill (in place) of any dict keyed like the
    reference state_dict.  Applied identically to the imported reference model
    (when making goldens), to the oracle and to the HIP model, so no weights
    have to be shipped.  Values have trained-net-like scales."""
    kinds = {}
    for arch in ARCHS:
        kinds.update({k: kind for k, _, kind in state_spec(arch)})
    for key, t in state.items():
        kind = kinds[key]
        g = torch.Generator().manual_seed(_key_seed(key, seed))
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


def new_state(seed=0, arch='resnet18'):
    sd = OrderedDict()
    for key, shape, kind in state_spec(arch):
        sd[key] = torch.zeros(shape, dtype=torch.int64 if kind == 'bn_nbt' else torch.float32)
    return procedural_fill(sd, seed)


def arch_of(sd):
    """Architecture of a state dict, from its keys."""
    bott = 'backbone.layer1.0.conv3.weight' in sd
    dcn = 'backbone.layer2.0.conv2_offset.weight' in sd
    return ('deformable_' if dcn else '') + ('resnet50' if bott else 'resnet18')


def trainable_keys(include_dead=False, arch='resnet18'):
    out = []
    for key, _, kind in state_spec(arch):
        if kind in ('bn_rm', 'bn_rv', 'bn_nbt'):
            continue
        if kind == 'dead' and not include_dead:
            continue
        out.append(key)
    return out


def synthetic_batch(n, size, seed=0, img_scale=1.0):
    """Inputs of SURVEY.md §8c(2): img ~ N(0,1)·scale; binary masks like the real
    loader (data_loaders.py:158-165); gts stacked in train.py:163-166 order."""
    g = torch.Generator().manual_seed(seed)
    h, w = (size, size) if isinstance(size, int) else size  # (h, w): the inference CLIs resize without padding (utils.py:160-175)
    img = torch.randn(n, 3, h, w, generator=g) * img_scale
    u = torch.rand(4, n, h, w, generator=g)
    prob_gt = (u[0] > 0.9).float()
    sup_mask = (u[1] > 0.05).float()
    thresh_gt = 0.3 + 0.4 * u[2]
    text_area = (u[3] > 0.8).float()
    return img, torch.stack([prob_gt, sup_mask, thresh_gt, text_area])


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------


def _bn(sd, prefix, x, training, update_stats):
    w, b = sd[prefix + '.weight'], sd[prefix + '.bias']
    rm, rv = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    if not training:
        return F.batch_norm(x, rm, rv, w, b, False, BN_MOMENTUM, BN_EPS)
    if update_stats:
        y = F.batch_norm(x, rm, rv, w, b, True, BN_MOMENTUM, BN_EPS)
        sd[prefix + '.num_batches_tracked'] += 1
        return y
    return F.batch_norm(x, None, None, w, b, True, BN_MOMENTUM, BN_EPS)


def deform_conv2d(x, offset, weight, stride=1, pad=1):
    """torchvision.ops.deform_conv2d (v0.6.0, deformable_groups = 1, dilation 1, no bias) restated from DCNv1:
    out[n,co,ho,wo] = sum_{ci,k} w[co,ci,k] * bilinear(x[n,ci], ho*stride - pad + r_k + dy_k, wo*stride - pad + s_k + dx_k),
    offset channel 2k = dy_k, 2k+1 = dx_k over the R*S taps k = r*S + s; samples outside (-1, H) x (-1, W) are zero and
    each of the four corner reads is zero when its index is out of range (bilinear_interpolate of deform_conv2d_kernel).
    Built from gathers, so autograd yields the gradients w.r.t. x, offset and weight.
    PARITY UNPINNED: torchvision is not installed here and absent from /root/reference; pinned only by the
    zero-offset identity with F.conv2d and by finite differences (tests/test_oracle_golden.py)."""
    N, C, H, W = x.shape
    Co, _, R, S = weight.shape
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
    assert offset.shape == (N, 2 * R * S, Ho, Wo), offset.shape
    ho = torch.arange(Ho, dtype=x.dtype).view(1, Ho, 1) * stride - pad
    wo = torch.arange(Wo, dtype=x.dtype).view(1, 1, Wo) * stride - pad
    xf = x.reshape(N, C, H * W)
    cols = []
    for k in range(R * S):
        r, s_ = k // S, k % S
        y = ho + r + offset[:, 2 * k]  # [N,Ho,Wo]
        xx = wo + s_ + offset[:, 2 * k + 1]
        inside = (y > -1) & (y < H) & (xx > -1) & (xx < W)
        y0, x0 = torch.floor(y), torch.floor(xx)
        ly, lx = y - y0, xx - x0
        val = 0
        for (yy, xc, wgt) in ((y0, x0, (1 - ly) * (1 - lx)), (y0, x0 + 1, (1 - ly) * lx), (y0 + 1, x0, ly * (1 - lx)),
                              (y0 + 1, x0 + 1, ly * lx)):
            ok = inside & (yy >= 0) & (yy <= H - 1) & (xc >= 0) & (xc <= W - 1)
            idx = (yy.clamp(0, H - 1) * W + xc.clamp(0, W - 1)).long().view(N, 1, Ho * Wo).expand(N, C, Ho * Wo)
            g = torch.gather(xf, 2, idx).view(N, C, Ho, Wo)
            val = val + g * (wgt * ok.to(x.dtype)).unsqueeze(1)
        cols.append(val)
    col = torch.stack(cols, 2)  # [N, C, RS, Ho, Wo]
    return torch.einsum('ock,nckhw->nohw', weight.reshape(Co, C, R * S), col)


def _conv2(sd, p, x, stride):
    """3x3 conv2 of a block: plain (resnet.py:46-50 / 103-108) or conv2_offset + DeformConv2d (:54-65,81-82 / 111-124,145-146)."""
    if p + '.conv2_offset.weight' in sd:
        offset = F.conv2d(x, sd[p + '.conv2_offset.weight'], sd[p + '.conv2_offset.bias'], stride, 1)
        return deform_conv2d(x, offset, sd[p + '.conv2.weight'], stride, 1)
    return F.conv2d(x, sd[p + '.conv2.weight'], None, stride, 1)


def _bottleneck(sd, p, x, stride, has_down, training, upd):
    # resnet.py:135-159
    out = F.conv2d(x, sd[p + '.conv1.weight'], None, 1, 0)
    out = F.relu(_bn(sd, p + '.bn1', out, training, upd))
    out = _conv2(sd, p, out, stride)
    out = F.relu(_bn(sd, p + '.bn2', out, training, upd))
    out = F.conv2d(out, sd[p + '.conv3.weight'], None, 1, 0)
    out = _bn(sd, p + '.bn3', out, training, upd)
    if has_down:
        res = F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride, 0)
        res = _bn(sd, p + '.downsample.1', res, training, upd)
    else:
        res = x
    return F.relu(out + res)


def _basic_block(sd, p, x, stride, has_down, training, upd):
    # resnet.py:70-91
    out = F.conv2d(x, sd[p + '.conv1.weight'], None, stride, 1)
    out = F.relu(_bn(sd, p + '.bn1', out, training, upd))
    out = _conv2(sd, p, out, 1)
    out = _bn(sd, p + '.bn2', out, training, upd)
    if has_down:
        res = F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride, 0)
        res = _bn(sd, p + '.downsample.1', res, training, upd)
    else:
        res = x
    return F.relu(out + res)


def _cbr(sd, p, x, pad, training, upd):
    # basic.py:32-36
    y = F.conv2d(x, sd[p + '.conv.weight'], sd[p + '.conv.bias'], 1, pad)
    return F.relu(_bn(sd, p + '.bn', y, training, upd))


def _nearest(x, size):
    return F.interpolate(x, size=size)  # default mode='nearest' (segmentation_body.py:79-87)


def _head_branch(sd, p, x, training, upd, has_bias0, taps=None, tag=''):
    # segmentation_head.py:24-29 / 64-79
    t = taps if taps is not None else {}
    y0 = F.conv2d(x, sd[p + '.0.weight'], sd[p + '.0.bias'] if has_bias0 else None, 1, 1)
    z0 = F.relu(_bn(sd, p + '.1', y0, training, upd))
    y1 = F.conv_transpose2d(z0, sd[p + '.3.weight'], sd[p + '.3.bias'], 2)
    z1 = F.relu(_bn(sd, p + '.4', y1, training, upd))
    y2 = F.conv_transpose2d(z1, sd[p + '.6.weight'], sd[p + '.6.bias'], 2)
    t[tag + '/y0'], t[tag + '/z0'], t[tag + '/y1'], t[tag + '/z1'] = y0, z0, y1, z1
    return torch.sigmoid(y2)


def forward(sd, x, training=True, update_stats=True, taps=None):
    """DBTextModel.forward (models.py:34-48).  `taps`, if a dict, receives the
    intermediate feature maps (NCHW) by name."""
    H, W = x.shape[2], x.shape[3]
    upd = update_stats
    t = taps if taps is not None else {}
    # stem (resnet.py:231-235)
    y = F.conv2d(x, sd['backbone.conv1.weight'], None, 2, 3)
    y = F.relu(_bn(sd, 'backbone.bn1', y, training, upd))
    y = F.max_pool2d(y, 3, 2, 1)
    t['pool'] = y
    feats = []
    blockfn = _bottleneck if 'backbone.layer1.0.conv3.weight' in sd else _basic_block
    for li in range(1, 5):
        bi = 0
        while 'backbone.layer%d.%d.conv1.weight' % (li, bi) in sd:  # resnet.py:210-229 (_make_layer)
            p = 'backbone.layer%d.%d' % (li, bi)
            y = blockfn(sd, p, y, 2 if (bi == 0 and li > 1) else 1, p + '.downsample.0.weight' in sd, training, upd)
            bi += 1
        feats.append(y)
        t['c%d' % (li + 1)] = y
    c2, c3, c4, c5 = feats
    b = 'segmentation_body.'
    # FPN (segmentation_body.py:64-77)
    p5 = _cbr(sd, b + 'reduce_conv_c5', c5, 0, training, upd)
    p4 = _nearest(p5, c4.shape[2:]) + _cbr(sd, b + 'reduce_conv_c4', c4, 0, training, upd)
    p4 = _cbr(sd, b + 'smooth_p4', p4, 1, training, upd)
    p3 = _nearest(p4, c3.shape[2:]) + _cbr(sd, b + 'reduce_conv_c3', c3, 0, training, upd)
    p3 = _cbr(sd, b + 'smooth_p3', p3, 1, training, upd)
    p2 = _nearest(p3, c2.shape[2:]) + _cbr(sd, b + 'reduce_conv_c2', c2, 0, training, upd)
    p2 = _cbr(sd, b + 'smooth_p2', p2, 1, training, upd)
    hw = p2.shape[2:]
    cat = torch.cat([p2, _nearest(p3, hw), _nearest(p4, hw), _nearest(p5, hw)], dim=1)
    f = F.conv2d(cat, sd[b + 'conv.0.weight'], sd[b + 'conv.0.bias'], 1, 1)
    f = F.relu(_bn(sd, b + 'conv.1', f, training, upd))
    t['p5'], t['p4'], t['p3'], t['p2'], t['fpn'] = p5, p4, p3, p2, f
    # DB head (segmentation_head.py:35-45)
    P = _head_branch(sd, 'segmentation_head.binarize', f, training, upd, True, t, 'binarize')
    T = _head_branch(sd, 'segmentation_head.thresh', f, training, upd, False, t, 'thresh')
    if training:
        B = torch.reciprocal(1 + torch.exp(-K_STEP * (P - T)))  # :106-108
        y = torch.cat((P, T, B), dim=1)
    else:
        y = torch.cat((P, T), dim=1)
    # models.py:43-46 — identity when H, W are multiples of 32
    return F.interpolate(y, size=(H, W), mode='bilinear', align_corners=True)


# ----------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------


def ohem_bce(pred, gt, mask, negative_ratio=3, eps=1e-6, reduction='mean'):
    """losses.py:18-40, statement by statement (including the scalar-BCE quirk
    under reduction='mean')."""
    positive = gt * mask
    negative = (1 - gt) * mask
    n_pos = int(positive.sum())
    n_neg = min(int(n_pos * negative_ratio), int(negative.sum()))
    loss = F.binary_cross_entropy(pred, gt, reduction=reduction)
    pos_loss = loss * positive
    neg_loss = loss * negative
    neg_loss, _ = torch.topk(neg_loss.reshape(-1), n_neg)
    return (pos_loss.sum() + neg_loss.sum()) / (n_pos + n_neg + eps)


def dice(pred, gt, mask, eps=1e-6):
    # losses.py:62-64
    inter = (pred * gt * mask).sum()
    union = (pred * mask).sum() + (gt * mask).sum() + eps
    return 1 - 2.0 * inter / union


def masked_l1(pred, gt, mask, eps=1e-6):
    # losses.py:77-78
    return (torch.abs(pred - gt) * mask).sum() / (mask.sum() + eps)


def db_loss(preds, gts, alpha=1.0, beta=10.0, reduction='mean', negative_ratio=3, eps=1e-6):
    """DBLoss.forward (losses.py:105-139).  Returns the 5-tuple for 3-channel
    preds, the single prob+beta*thresh value for 2-channel preds."""
    assert preds.dim() == 4 and gts.dim() == 4
    P, T = preds[:, 0], preds[:, 1]
    G, M, Tg, A = gts[0], gts[1], gts[2], gts[3]
    prob = ohem_bce(P, G, M, negative_ratio, eps, reduction)
    thr = masked_l1(T, Tg, A, eps)
    pt = prob + beta * thr
    if preds.size(1) == 3:
        binl = dice(preds[:, 2], G, M, eps)
        return prob, thr, binl, pt, alpha * binl + pt
    return pt


def db_loss_closed_form(preds, gts, alpha=1.0, beta=10.0, negative_ratio=3, eps=1e-6, reduction='mean'):
    """What the HIP loss kernel evaluates for reduction='mean' / 'sum' with binary
    gt/mask: bce_scalar * (sum_pos + n_neg) / (n_pos+n_neg+eps).
    Float64 sums; used to cross-check the literal restatement above."""
    P, T = preds[:, 0].double(), preds[:, 1].double()
    G, M, Tg, A = (g.double() for g in gts)
    s_pos = (G * M).sum()
    s_neg = ((1 - G) * M).sum()
    n_pos = int(s_pos)
    n_neg = min(int(n_pos * negative_ratio), int(s_neg))
    bce = -(G * torch.clamp(torch.log(P), min=-100) + (1 - G) * torch.clamp(torch.log1p(-P), min=-100))
    bce = bce.sum() if reduction == 'sum' else bce.mean()
    prob = bce * (s_pos + n_neg) / (n_pos + n_neg + eps)
    thr = ((T - Tg).abs() * A).sum() / (A.sum() + eps)
    out = [prob, thr]
    if preds.size(1) == 3:
        B = preds[:, 2].double()
        binl = 1 - 2.0 * (B * G * M).sum() / ((B * M).sum() + s_pos + eps)
        out += [binl, prob + beta * thr, alpha * binl + prob + beta * thr]
    else:
        out += [prob + beta * thr]
    return [float(v) for v in out]


# ----------------------------------------------------------------------------
# per-step loop (train.py:160-172) with an explicit Adam restatement
# ----------------------------------------------------------------------------


class AdamState:
    """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=0,
    amsgrad=False) restated (train.py:114-117)."""

    def __init__(self, lr=0.005, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, sd, grads):
        self.t += 1
        bc1 = 1 - self.b1**self.t
        bc2 = 1 - self.b2**self.t
        for k, g in grads.items():
            if g is None:
                continue
            if k not in self.m:
                self.m[k] = torch.zeros_like(g)
                self.v[k] = torch.zeros_like(g)
            m, v = self.m[k], self.v[k]
            m.mul_(self.b1).add_(g, alpha=1 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (v.sqrt() / math.sqrt(bc2)).add_(self.eps)
            sd[k].addcdiv_(m, denom, value=-self.lr / bc1)


def loss_and_grads(sd, img, gts, update_stats=True, world_scale=1.0, **loss_kw):
    """forward + DBLoss + backward; returns (preds, losses(5 floats), grads dict)."""
    keys = trainable_keys(arch=arch_of(sd))
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in keys}
    work = OrderedDict(sd)
    work.update(leaves)
    preds = forward(work, img, training=True, update_stats=update_stats)
    for k in sd:  # carry running-stat updates back
        if k not in leaves:
            sd[k] = work[k]
    losses = db_loss(preds, gts, **loss_kw)
    total = losses[4]
    grads = torch.autograd.grad(total, [leaves[k] for k in keys], allow_unused=True)
    grads = {k: (g * world_scale if g is not None else None) for k, g in zip(keys, grads)}
    return preds.detach(), [float(v.detach()) for v in losses], grads


def to_dtype(sd, dtype):
    """Copy of a state dict with the floating tensors in `dtype` (fp64 evaluation of the same weights: the reference in
    `.double()`, tests/golden/fp64_2x128.npz)."""
    return OrderedDict((k, v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in sd.items())


def train_step(sd, opt, img, gts, **loss_kw):
    """One iteration of train.py:160-172 on the flat state dict."""
    preds, losses, grads = loss_and_grads(sd, img, gts, **loss_kw)
    with torch.no_grad():
        opt.step(sd, grads)
    return preds, losses


# ----------------------------------------------------------------------------
# per-step pixel metric (text_metrics.py:9-82), numpy restatement
# ----------------------------------------------------------------------------


def pixel_confusion(texts, gt_texts, training_masks, thresh=0.5, n_classes=2):
    """cal_text_score's thresholding (text_metrics.py:73-80) + RunningScore._fast_hist (:14-24),
    summed over the batch.  Returns the n x n confusion matrix (rows: gt, cols: pred)."""
    import numpy as np
    masks = training_masks.detach().cpu().numpy()
    pred = texts.detach().cpu().numpy() * masks
    pred[pred <= thresh] = 0
    pred[pred > thresh] = 1
    pred = pred.astype(np.int32)
    gt = (gt_texts.detach().cpu().numpy() * masks).astype(np.int32)
    hist = np.zeros((n_classes, n_classes))
    for lt, lp in zip(gt, pred):
        lt, lp = lt.flatten(), lp.flatten()
        m = (lt >= 0) & (lt < n_classes)
        hist += np.bincount(n_classes * lt[m].astype(int) + lp[m], minlength=n_classes**2).reshape(n_classes, n_classes)
    return hist


def scores_from_confusion(hist):
    """RunningScore.get_scores (text_metrics.py:36-58)."""
    import numpy as np
    acc = np.diag(hist).sum() / (hist.sum() + 0.0001)
    acc_cls = np.nanmean(np.diag(hist) / (hist.sum(axis=1) + 0.0001))
    iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist) + 0.0001)
    freq = hist.sum(axis=1) / (hist.sum() + 0.0001)
    return {'Overall Acc': acc, 'Mean Acc': acc_cls, 'FreqW Acc': (freq[freq > 0] * iu[freq > 0]).sum(), 'Mean IoU': np.nanmean(iu)}
