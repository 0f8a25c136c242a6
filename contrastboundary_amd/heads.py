"""Host mirror of the reference's CBL criterion, /root/reference/pytorch/model/heads.py:63-253 (ContrastHead).

Same constructor arguments (head_cfg, config), same forward(output, target, stage_list) -> list of scalar losses (one per
stage of head_cfg.stage), same numbers.  The whole of point_contrast (:185-246) after the two knnquery calls is ONE fused HIP
kernel forward and one backward (csrc/cbl.hip) instead of ~15 torch ops, 4 materialised (m,K-1,.) tensors and a host sync.
Supported head options = the shipped config (config/s3dis/origin_multi-...-contrast-Ua-softnn-latent-label-l2-w.1.yaml:60-68):
pos='cnt', dist='l2', contrast='softnn', sample='label', no projection MLP; anything else raises NotImplementedError.
"""
import ctypes
import re

import torch
from torch.autograd import Function

from . import _lib, pointops
from .basic_operators import get_subscene_label

_c_int = ctypes.c_int
_c_float = ctypes.c_float


def parse_stage(stage, num_layers):
    """model/utils.py:27-35: 'Ua' -> [('up',0),...,('up',4)], 'D012_U34' -> ..."""
    stage = stage.replace("a", "".join(f"{i}" for i in range(num_layers)))
    parts = [i.strip("_") for i in re.split(r"(\d+)", stage) if i and i.strip("_")]
    assert len(parts) % 2 == 0, f"invalid stage compound: {parts} from {stage}"
    names = {"D": "down", "down": "down", "U": "up", "up": "up"}
    out = []
    for n, digits in zip(parts[0::2], parts[1::2]):
        out += [(names[n], int(d)) for d in digits]
    return out


class _PointContrast(Function):
    """Forward: mining + loss terms (+, when the features need a gradient, the scalar coefficient of every pair and the centre half of
    the gradient) in one pass — cbl_contrast_pairs_forward.  Backward: the neighbour half as a gather over the transposed neighbour
    table — cbl_contrast_pairs_backward: no atomics, no zero fill, deterministic (SURVEY.md 7 hard part 6)."""

    @staticmethod
    def forward(ctx, features, amax, neighbor_idx, temperature, weight, transposed, nce=False):
        m, d = features.shape
        nsample = neighbor_idx.shape[1]
        dev = features.device
        flags = (2 if amax.dtype == torch.int64 else 0) | (4 if nce else 0)   # int64: the reference's hard targets as they are, no int32 copy
        per_point = torch.empty(m, dtype=torch.float32, device=dev)
        mask = torch.empty(m, dtype=torch.int32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        L = _lib.lib()
        grad = ctx.needs_input_grad[0]
        order = pointops.spatial_order(neighbor_idx)                     # processing order only: the values do not depend on it
        coef = torch.empty((m, nsample), dtype=torch.float32, device=dev) if grad else None
        own = torch.empty((m, d), dtype=torch.float32, device=dev) if grad else None
        _lib.check(L.cbl_contrast_pairs_forward(_c_int(m), _c_int(0x7fffffff), _c_int(flags), _c_int(nsample), _c_int(d), _lib.ptr(features), _lib.ptr(amax),
                                                _c_int(0), _c_float(0.0), _lib.ptr(neighbor_idx), _lib.ptr(order), _c_float(temperature), _c_float(weight),
                                                _lib.ptr(per_point), _lib.ptr(mask), _lib.ptr(stats), _lib.ptr(loss), _lib.ptr(coef), _lib.ptr(own),
                                                _lib.stream_of(features)), "cbl_contrast_pairs_forward")
        if grad:
            # the transposed neighbour table is only needed by the backward pass: handed in, or found in / built into the registry then
            ctx.save_for_backward(features, coef, own, stats, neighbor_idx, *(() if transposed is None else transposed[1:] + ((transposed[0],) if transposed[0] is not None else ())))
        ctx.weight, ctx.nsample = weight, nsample
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)        # no zero tensor for the (integer) mask output in backward: that was one fill launch per step
        return loss.view(()), mask

    @staticmethod
    def backward(ctx, grad_loss, _grad_mask):
        if grad_loss is None:                                            # the loss took no part in what was differentiated
            return None, None, None, None, None, None, None
        features, coef, own, stats, neighbor_idx, *rest = ctx.saved_tensors
        if rest:
            inv_start, inv_src = rest[0], rest[1]
            order = rest[2] if len(rest) > 2 else None
        else:
            tr = pointops.neighbor_transpose(neighbor_idx, features.shape[0])
            if tr is None:                                           # beyond the table's size limit: the reference's own scatter, with atomics
                return _pairs_backward_atomic(features, coef, own, stats, neighbor_idx, grad_loss, ctx.weight, ctx.nsample), None, None, None, None, None, None
            order, inv_start, inv_src = tr
        m, d = features.shape
        g = torch.empty_like(features)
        gl = grad_loss.reshape(1).to(torch.float32).contiguous()
        _lib.check(_lib.lib().cbl_contrast_pairs_backward(_c_int(m), _c_int(ctx.nsample), _c_int(d), _lib.ptr(features), _lib.ptr(coef), _lib.ptr(own),
                                                          _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(stats), _lib.ptr(gl),
                                                          _c_float(ctx.weight), _lib.ptr(g), _lib.stream_of(features)), "cbl_contrast_pairs_backward")
        return g, None, None, None, None, None, None


def _pairs_backward_atomic(features, coef, own, stats, neighbor_idx, grad_loss, weight, nsample):
    """cbl_contrast_pairs_backward_atomic: the neighbour half of the CBL gradient scattered with float atomics (no transposed table: > 1 M rows)"""
    m, d = features.shape
    g = torch.empty_like(features)
    gl = grad_loss.reshape(1).to(torch.float32).contiguous()
    _lib.check(_lib.lib().cbl_contrast_pairs_backward_atomic(_c_int(m), _c_int(m), _c_int(nsample), _c_int(d), _lib.ptr(features), _lib.ptr(coef), _lib.ptr(own),
                                                             _lib.ptr(neighbor_idx), _lib.ptr(stats), _lib.ptr(gl), _c_float(weight), _lib.ptr(g),
                                                             _lib.stream_of(features)), "cbl_contrast_pairs_backward_atomic")
    return g


def point_contrast(features, labels, neighbor_idx, temperature=1.0, weight=0.1, return_mask=False, transposed=None, contrast="softnn"):
    """features (m,d) f32, labels (m,ncls) f32 soft/one-hot OR (m,) int class ids, neighbor_idx (m,nsample) i32 incl. the self column
    -> scalar loss (device tensor, differentiable w.r.t. features).  transposed: pointops.neighbor_transpose(neighbor_idx, m) if the
    caller already has it (it is looked up in / built into the active neighbour cache otherwise).
    contrast: 'softnn' (heads.py:151-165) or 'nce' (:167-183: one term per positive pair, mean over all of them)."""
    if contrast not in ("softnn", "nce"):
        raise NotImplementedError(f"point_contrast: contrast={contrast!r}")
    if features.shape[1] not in (4, 8, 16, 32, 64):
        raise NotImplementedError(f"point_contrast: feature width {features.shape[1]} (the fused head covers 4, 8, 16, 32, 64)")
    m = features.shape[0]
    if labels.dim() == 2:
        amax = torch.empty(m, dtype=torch.int32, device=features.device)
        labels = labels.contiguous()
        _lib.check(_lib.lib().cbl_label_argmax(_c_int(m), _c_int(labels.shape[1]), _lib.ptr(labels), _lib.ptr(amax), _lib.stream_of(features)),
                   "cbl_label_argmax")
    else:
        amax = labels.contiguous() if labels.dtype in (torch.int64, torch.int32) else labels.to(torch.int32).contiguous()
    loss, mask = _PointContrast.apply(features.contiguous(), amax, neighbor_idx.contiguous(), float(temperature), float(weight), transposed,
                                      contrast == "nce")
    return (loss, (mask > 0).to(torch.int32) if contrast == "nce" else mask) if return_mask else loss   # ('nce' keeps the number of positives)


class ContrastHead(torch.nn.Module):
    """heads.py:63-253.  Used as a criterion: forward(output, target, stage_list) -> [loss per stage]."""

    def __init__(self, head_cfg, config):
        super().__init__()
        self.nsample = [int(v) for v in config.nsample]
        self.nstride = [int(v) for v in config.nstride]
        self.num_classes = int(config.num_classes)
        self.head_cfg, self.config = head_cfg, config
        self.stages = parse_stage(head_cfg.stage, config.num_layers)
        self.ftype = head_cfg.ftype if head_cfg.ftype not in ("out", "fout") else "f_out"
        # every option the REFERENCE can run: dist 'l2' (dist_kl is called with two of its four arguments, :235, and no other dist_* exists),
        # pos 'cnt' (the only posmask_* defined, :145), contrast 'softnn' | 'nce' (:151-183), an optional projection MLP (:88-92)
        for key, allowed in (("dist", ("l2",)), ("pos", ("cnt",)), ("contrast", ("softnn", "nce"))):
            if getattr(head_cfg, key) not in allowed:
                raise NotImplementedError(f"ContrastHead {key}={getattr(head_cfg, key)!r}: the reference itself only runs {allowed} "
                                          "(heads.py:116-183: dist_kl's call passes two of its four arguments, no other posmask_* / dist_* exists)")
        assert head_cfg.sample in ["cnt", "glb", "sub", "subspatial", "pts", "label", "vote"], f"not support sample = {head_cfg.sample}"
        self.project = None
        if "project" in head_cfg and head_cfg.project:
            from .blocks import MLPbyOps
            self.project = torch.nn.ModuleDict({f"{n}{i}": MLPbyOps(head_cfg.project, config.base_fdim * 2 ** i, d_out=config.base_fdim)
                                                for n, i in self.stages})            # heads.py:88-92
        if self.ftype != "latent" and self.project is None:
            # 'f_out' / 'out' hand the head the stage widths (pointtransformer_seg.py: planes 32 ... 512); the fused kernels cover rows of 4 ... 64 floats
            planes = [int(v) for v in config.planes] if "planes" in config else [32, 64, 128, 256, 512]
            bad = [(n, i) for n, i in self.stages if planes[i] not in (4, 8, 16, 32, 64)]
            if bad:
                raise NotImplementedError(f"ContrastHead ftype={head_cfg.ftype!r}: stages {bad} are wider than the 64 floats per row the fused HIP "
                                          "path covers (the shipped config contrasts the 32-d 'latent' features)")
        self.temperature = float(head_cfg.temperature) if "temperature" in head_cfg and head_cfg.temperature is not None else 1.0
        self.weight = float(head_cfg.weight[1:])                         # 'w.1' -> 0.1, heads.py:241-243

    def point_contrast(self, n, i, stage_list, target):
        stage = stage_list[n][i]
        p, features, o = stage["p_out"], stage[self.ftype], stage["offset"]
        if self.project is not None:
            features = self.project[f"{n}{i}"](features)                  # :187-188
        if i == 0:
            labels = target                                               # one-hot's argmax is the label itself
        else:
            labels = get_subscene_label(n, i, stage_list, target, self.nstride, self.num_classes)   # :189
        neighbor_idx, _ = pointops.knnquery_raw(self.nsample[i], p, p, o, o, algo="set")           # :192; the mining is order-invariant
        return point_contrast(features, labels, neighbor_idx, self.temperature, self.weight, contrast=self.head_cfg.contrast)

    def forward(self, output, target, stage_list):
        return [self.point_contrast(n, i, stage_list, target) for n, i in self.stages]              # :248-253


# ---------------------------------------------------------------------------------------------------------------------------
# TF flavour: /root/reference/tensorflow/models/heads/head.py:462-807 (contrast_head), scene labels :25-49 / :117-131
# ---------------------------------------------------------------------------------------------------------------------------
class _TFContrast(Function):
    @staticmethod
    def forward(ctx, features, labels, neighbors, temperature, weight, kl_threshold=None):
        m, d = features.shape
        n_valid = labels.shape[0]
        dev = features.device
        per_point = torch.empty(m, dtype=torch.float32, device=dev)
        mask = torch.empty(m, dtype=torch.int32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        L = _lib.lib()
        if kl_threshold is None:
            args = (_c_int(m), _c_int(n_valid), _c_int(neighbors.shape[1]), _c_int(d), _lib.ptr(features), _lib.ptr(labels), _lib.ptr(neighbors),
                    _c_float(temperature), _c_float(weight), _lib.ptr(per_point), _lib.ptr(mask), _lib.ptr(stats), _lib.ptr(loss))
            fwd, fwd_grad = L.cbl_tf_contrast_forward, L.cbl_tf_contrast_forward_grad
        else:                                                            # sample 'labelkl<thr>': labels are (N, ncls) distributions
            args = (_c_int(m), _c_int(n_valid), _c_int(neighbors.shape[1]), _c_int(d), _lib.ptr(features), _lib.ptr(labels), _c_int(labels.shape[1]),
                    _c_float(kl_threshold), _lib.ptr(neighbors), _c_float(temperature), _c_float(weight), _lib.ptr(per_point), _lib.ptr(mask),
                    _lib.ptr(stats), _lib.ptr(loss))
            fwd, fwd_grad = L.cbl_tf_contrast_forward_kl, L.cbl_tf_contrast_forward_grad_kl
        if ctx.needs_input_grad[0]:
            unit = torch.zeros_like(features)
            _lib.check(fwd_grad(*args, _lib.ptr(unit), _lib.stream_of(features)), "cbl_tf_contrast_forward_grad")
            ctx.save_for_backward(unit, stats)
        else:
            _lib.check(fwd(*args, _lib.stream_of(features)), "cbl_tf_contrast_forward")
        ctx.weight = weight
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)        # no zero tensor for the (integer) mask output in backward: that was one fill launch per step
        return loss.view(()), mask

    @staticmethod
    def backward(ctx, grad_loss, _gm):
        unit, stats = ctx.saved_tensors
        if grad_loss is None:                                            # the loss took no part in what was differentiated
            return None, None, None, None, None, None
        g = torch.empty_like(unit)
        gl = grad_loss.reshape(1).to(torch.float32).contiguous()
        _lib.check(_lib.lib().cbl_contrast_grad_scale(ctypes.c_longlong(unit.numel()), _lib.ptr(unit), _lib.ptr(stats), _lib.ptr(gl), _c_float(ctx.weight),
                                                      _lib.ptr(g), _lib.stream_of(unit)), "cbl_contrast_grad_scale")
        return g, None, None, None, None, None


class _TFContrastPairs(Function):
    """TF flavour through the atomic-free kernels (cbl_contrast_pairs_*, flags bit 0): contrast 'nce' (head.py:773-795), the sample strings beyond
    'label' (head.py:560-625) and the 'S' margin (:759-760, :783-785)"""

    @staticmethod
    def forward(ctx, features, labels, samples, temperature, weight, nce, separate=False, roles=None, sample_valid=None, kl_threshold=None):
        m, d = features.shape
        n_valid, nsample = labels.shape[0], samples.shape[1]
        if n_valid != m:
            raise NotImplementedError("tf_contrast through the pair kernels: labels must cover exactly the stage's own points")
        if nsample > 65:
            raise NotImplementedError(f"tf_contrast: {nsample - 1} sample columns (at most 64)")
        dev = features.device
        per_point = torch.empty(m, dtype=torch.float32, device=dev); mask = torch.empty(m, dtype=torch.int32, device=dev)
        stats = torch.empty(2, dtype=torch.float32, device=dev); loss = torch.empty(1, dtype=torch.float32, device=dev)
        grad = ctx.needs_input_grad[0]
        coef = torch.empty((m, nsample), dtype=torch.float32, device=dev) if grad else None
        own = torch.empty((m, d), dtype=torch.float32, device=dev) if grad else None
        order = pointops.spatial_order(samples)
        ncls = 0 if kl_threshold is None else labels.shape[1]
        flags = 1 | (4 if nce else 0) | (8 if separate else 0)
        _lib.check(_lib.lib().cbl_contrast_pairs_forward_samples(
            _c_int(m), _c_int(n_valid), _c_int(flags), _c_int(nsample), _c_int(d), _lib.ptr(features), _lib.ptr(labels), _c_int(ncls),
            _c_float(0.0 if kl_threshold is None else kl_threshold), _lib.ptr(samples), _lib.ptr(roles), _lib.ptr(sample_valid), _lib.ptr(order),
            _c_float(temperature), _c_float(weight), _lib.ptr(per_point), _lib.ptr(mask), _lib.ptr(stats), _lib.ptr(loss), _lib.ptr(coef), _lib.ptr(own),
            _lib.stream_of(features)), "cbl_contrast_pairs_forward_samples")
        if grad:
            ctx.save_for_backward(features, coef, own, stats, samples)
        ctx.weight, ctx.nsample = weight, nsample
        ctx.mark_non_differentiable(mask)
        ctx.set_materialize_grads(False)
        return loss.view(()), mask

    @staticmethod
    def backward(ctx, grad_loss, _gm):
        none = (None,) * 9
        if grad_loss is None:
            return (None,) + none
        features, coef, own, stats, samples = ctx.saved_tensors
        m, d = features.shape
        tr = pointops.neighbor_transpose(samples, m)                     # shadow neighbours (index m) are left out of the table
        if tr is None:
            return (_pairs_backward_atomic(features, coef, own, stats, samples, grad_loss, ctx.weight, ctx.nsample),) + none
        order, inv_start, inv_src = tr
        g = torch.empty_like(features)
        gl = grad_loss.reshape(1).to(torch.float32).contiguous()
        _lib.check(_lib.lib().cbl_contrast_pairs_backward(_c_int(m), _c_int(ctx.nsample), _c_int(d), _lib.ptr(features), _lib.ptr(coef), _lib.ptr(own),
                                                          _lib.ptr(order), _lib.ptr(inv_start), _lib.ptr(inv_src), _lib.ptr(stats), _lib.ptr(gl),
                                                          _c_float(ctx.weight), _lib.ptr(g), _lib.stream_of(features)), "cbl_contrast_pairs_backward")
        return (g,) + none


ROLE_LABEL, ROLE_POS, ROLE_NEG, ROLE_NEG_REJECT = 0, 1, 2, 3               # include/cbl_amd.h CBL_ROLE_*


def tf_sample_columns(neighbors, sample, rand_idx=None, batches_len=None, generator=None):
    """contrast_head.sample_labels (head.py:551-625) as arrays for cbl_contrast_pairs_forward_samples: neighbors (m,k) i32 incl. the self column
    -> samples (m, 1 + S) i32 (self column, then the '-'-joined segments), roles (S,) u8, sample_valid (m,S) u8 or None.
    'label*' = the neighbour columns, 'nn<k>' = the first k of them, 'rand<n>[R]' = n uniform draws per point from its own cloud (:568-596;
    batches_len (B,) = points per cloud, one cloud if None) — or the caller's rand_idx (list of (m,n) tensors, one per rand segment), since no
    generator here replays tf.random.uniform; 'R' marks the draws that hit one of the point's neighbours (:611-615)."""
    m = neighbors.shape[0]
    nbr = neighbors[:, 1:]
    rand_idx = list(rand_idx) if rand_idx is not None else []
    cols, roles, reject = [neighbors[:, :1]], [], []
    for seg in sample.split("-"):
        if seg.startswith("label"):
            cur, role = nbr, ROLE_LABEL
        elif seg.startswith("nn"):
            k = int(seg[2:])
            if k > nbr.shape[1]:
                raise ValueError(f"sample {seg!r}: only {nbr.shape[1]} neighbour columns")            # 'assume enough neighbor', :565
            cur, role = nbr[:, :k], ROLE_POS
        elif seg.startswith("rand"):
            n_neg = int("".join(ch for ch in seg[4:] if ch.isdigit()))
            if rand_idx:
                cur = rand_idx.pop(0).to(device=neighbors.device, dtype=torch.int32)
                if tuple(cur.shape) != (m, n_neg):
                    raise ValueError(f"sample {seg!r}: rand_idx of shape {tuple(cur.shape)}, expected {(m, n_neg)}")
            else:
                lens = [m] if batches_len is None else [int(v) for v in batches_len.tolist()]
                if sum(lens) != m:
                    raise ValueError("batches_len does not add up to the number of points")
                parts, start = [], 0
                for nb_ in lens:                                                                        # per-cloud draw, :574-589
                    parts.append(torch.randint(0, max(nb_, 1), (nb_, n_neg), device=neighbors.device, generator=generator, dtype=torch.int32) + start)
                    start += nb_
                cur = torch.cat(parts)
            role = ROLE_NEG_REJECT if "R" in seg else ROLE_NEG
        else:
            raise NotImplementedError(f"not supported sample = {seg} in {sample}")                      # :598-599
        cols.append(cur); roles += [role] * cur.shape[1]
        reject.append((cur[:, :, None] != nbr[:, None, :]).all(-1) if role == ROLE_NEG_REJECT else None)
    samples = torch.cat(cols, 1).contiguous()
    roles_t = torch.tensor(roles, dtype=torch.uint8, device=neighbors.device)
    valid = None
    if any(r is not None for r in reject):
        valid = torch.cat([torch.ones(c.shape, dtype=torch.uint8, device=neighbors.device) if r is None else r.to(torch.uint8)
                           for c, r in zip(cols[1:], reject)], 1).contiguous()
    return samples, roles_t, valid


def tf_contrast(features, labels, neighbors, temperature=1.0, weight=0.1, return_mask=False, kl_threshold=None, contrast="softnn", sample="label",
                margin=None, rand_idx=None, batches_len=None, generator=None, atomic_scatter=False):
    """TF contrast_head.contrast ('softnn' | 'nce', dist 'l2') for one stage: features (m,d) f32, neighbors (m,k) i32 radius neighbours incl. the self
    column, padded with N.  sample 'label': labels (N,) hard labels of the N support points of that stage (negative = ignored);
    sample 'labelkl<thr>' (kl_threshold=thr): labels (N,ncls) f32 label distributions (tf_scene_label(..., 'soft'); one-hot at stage 0);
    sample with 'nn<k>' / 'rand<n>[R]' segments: tf_sample_columns; margin: a string holding 'S' (pos / neg kept separate, head.py:759-760 /
    :783-785) and / or 'T<float>' (temperature, :740-743)."""
    if contrast not in ("softnn", "nce"):
        raise NotImplementedError(f"tf_contrast: contrast={contrast!r}")
    separate = False
    if margin:
        separate = "S" in margin
        if "T" in margin:
            temperature = float(margin[margin.index("T") + 1:])
    plain_sample = all(seg.startswith("label") for seg in sample.split("-")) and "-" not in sample
    if kl_threshold is None:
        lab = labels.to(torch.int32).contiguous()
    else:
        lab = labels.to(torch.float32).contiguous()
        if lab.dim() != 2 or lab.shape[1] > 255:
            raise ValueError("labelkl: labels must be (N, ncls <= 255) distributions")
    # Every configuration takes the pair kernels: mining + loss + per-pair coefficients in one pass, the neighbour half of the gradient as a gather
    # over the transposed table of `neighbors` (the table AdaptiveWeight's backward builds for the same tensor: cached, not rebuilt) — no float
    # atomics, run-to-run deterministic.  `atomic_scatter=True` keeps round 1's kernels reachable (cbl_tf_contrast_*: gradient by row atomics).
    if not atomic_scatter or contrast == "nce" or separate or not plain_sample:
        if plain_sample:
            samples, roles, valid = neighbors.contiguous(), None, None
        else:
            samples, roles, valid = tf_sample_columns(neighbors, sample, rand_idx, batches_len, generator)
        loss, mask = _TFContrastPairs.apply(features.contiguous(), lab, samples, float(temperature), float(weight), contrast == "nce", separate, roles,
                                            valid, None if kl_threshold is None else float(kl_threshold))
        return (loss, mask) if return_mask else loss
    loss, mask = _TFContrast.apply(features.contiguous(), lab, neighbors.contiguous(), float(temperature), float(weight),
                                   None if kl_threshold is None else float(kl_threshold))
    return (loss, mask) if return_mask else loss


def tf_scene_label(point_labels, scene_neighbor, num_classes, reduction="max"):
    """get_scene_label_infer (head.py:25-49): labels of sub-sampled points from `scene_neighbor` (m,k) i32 into the stage-0 points
    (pad = len(point_labels)); 'max' -> (m,) int64 hard labels, 'soft' -> (m,ncls) distribution over the VALID neighbours"""
    m, k = scene_neighbor.shape
    pl = point_labels.to(torch.int64).contiguous()
    out = torch.empty((m, num_classes), dtype=torch.float32, device=pl.device)
    nb = scene_neighbor.contiguous()
    _lib.check(_lib.lib().cbl_tf_scene_label(_c_int(m), _c_int(pl.shape[0]), _c_int(k), _c_int(num_classes), _lib.ptr(pl), _lib.ptr(nb),
                                             _c_int(0 if reduction in ("max", "cnt") else 1), _lib.ptr(out), _lib.stream_of(pl)), "cbl_tf_scene_label")
    if reduction in ("max", "cnt"):
        amax = torch.empty(m, dtype=torch.int32, device=pl.device)
        _lib.check(_lib.lib().cbl_label_argmax(_c_int(m), _c_int(num_classes), _lib.ptr(out), _lib.ptr(amax), _lib.stream_of(pl)), "cbl_label_argmax")
        return amax.long()
    return out
