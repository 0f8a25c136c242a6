"""cbl_oracle.py — TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's Contrastive Boundary Learning head (pytorch side) and boundary masks:
    get_subscene_label / get_subscene_features   /root/reference/pytorch/model/basic_operators.py:9-50
    ContrastHead.point_contrast                  /root/reference/pytorch/model/heads.py:185-246
        posmask_cnt :145-149, dist_l2 :116-119, contrast_softnn :151-165
    get_boundary_mask                            /root/reference/pytorch/model/basic_operators.py:69-97
Neighbour indices are inputs (the KNN itself is oracle/pointops_oracle.c).  Pinned by tests/golden/cbl_pytorch.npz and
boundary_mask.npz, which were produced by importing and running the reference's own Python on CPU
(tests/golden/gen_cbl_goldens.py).  float32 forward like the reference; summation ORDER differs from torch's
vectorised reductions, so the pin is 1e-5-relative, not bitwise.  Only tests/, smoke() and bench.py's cpu_baseline
may import this module.
"""
import numpy as np

_EPS = np.float32(1e-12)       # basic_operators.py:7


def subscene_label(target, neighbor_idx, num_classes):
    """soft label of a coarse point = mean one-hot label of its kr nearest stage-0 points (basic_operators.py:13,40-41).
    target (N,) int, neighbor_idx (m, kr) rows into target -> (m, num_classes) float32"""
    m, kr = neighbor_idx.shape
    lab = np.asarray(target)[neighbor_idx.reshape(-1)].reshape(m, kr)
    out = np.zeros((m, num_classes), np.float32)
    for c in range(num_classes):
        out[:, c] = (lab == c).sum(1).astype(np.float32) / np.float32(kr)      # x.float().mean(-2)
    return out


def one_hot_label(target, num_classes):
    return np.eye(num_classes, dtype=np.float32)[np.asarray(target)]           # stage 0: F.one_hot(...).float(), :13-17


def point_contrast(features, labels, neighbor_idx_full, temperature=None, weight=0.1, grad=True):
    """heads.py:185-246 for pos='cnt', dist='l2', contrast='softnn'.
    features (m,d) f32; labels (m,ncls) soft/one-hot; neighbor_idx_full (m,nsample) from knnquery (column 0 = self, dropped :196).
    -> loss (float32 scalar, 0 if no point has both a positive and a negative neighbour), d loss / d features (m,d), point_mask (m,)"""
    f = np.asarray(features, np.float32)
    nbr = np.asarray(neighbor_idx_full)[:, 1:]                                  # exclude self-loop, :195-196
    m, ns = nbr.shape
    amax = np.argmax(labels, axis=-1)                                           # first maximal index, like torch.argmax
    posmask = amax[:, None] == amax[nbr]                                        # posmask_cnt :145-149
    cnt = posmask.sum(1)
    point_mask = (cnt > 0) & (cnt < ns)                                         # :212-213
    g = np.zeros_like(f)
    if not point_mask.any():
        return np.float32(0.0), g, point_mask                                   # :233
    rows = np.nonzero(point_mask)[0]
    fi = f[rows]                                                                # (r,d)
    fj = f[nbr[rows]]                                                           # (r,ns,d)
    diff = fi[:, None, :] - fj
    dist = np.sqrt((diff * diff).sum(-1, dtype=np.float32) + _EPS).astype(np.float32)   # dist_l2 :116-119
    neg = -dist
    neg = neg - neg.max(-1, keepdims=True)                                      # :153
    if temperature is not None:
        neg = (neg / np.float32(temperature)).astype(np.float32)                # :154-155
    e = np.exp(neg).astype(np.float32)
    pm = posmask[rows].astype(np.float32)
    pos = (e * pm).sum(-1, dtype=np.float32)
    alls = e.sum(-1, dtype=np.float32)
    ratio = pos / alls
    per_point = -np.log(ratio + _EPS)                                           # :163
    r = np.float32(len(rows))
    loss = np.float32(per_point.mean(dtype=np.float32) * np.float32(weight))    # :241-243
    if not grad:
        return loss, g, point_mask
    # analytic gradient (float64 for the checker): d loss/d dist_j = w/r * e_j (pos_j*A - P) / (T A^2 (ratio+eps)); the max-shift
    # cancels in P/A.  d dist_j/d f_i = (f_i - f_j)/dist_j, d dist_j/d f_j = -(f_i - f_j)/dist_j.
    T = 1.0 if temperature is None else float(temperature)
    e64, pm64, A, P = e.astype(np.float64), pm.astype(np.float64), alls.astype(np.float64), pos.astype(np.float64)
    dl_dd = e64 * (pm64 * A[:, None] - P[:, None]) / (T * (A * A)[:, None] * (P / A + 1e-12)[:, None])
    dl_dd *= float(weight) / float(r)
    coef = (dl_dd / dist.astype(np.float64))[:, :, None] * diff.astype(np.float64)       # (r,ns,d)
    g64 = np.zeros(f.shape, np.float64)
    np.add.at(g64, rows, coef.sum(1))
    np.add.at(g64, nbr[rows].reshape(-1), -coef.reshape(-1, f.shape[1]))
    return loss, g64.astype(np.float32), point_mask


def boundary_mask(labels, neighbor_label, valid_mask=None, get_plain=False, get_cnt=False):
    """basic_operators.py:69-97: a point is a boundary point if any VALID (>= 0) neighbour label differs from its own."""
    labels = np.asarray(labels)[:, None]
    valid_nb = neighbor_label >= 0
    neq = (labels != neighbor_label) & valid_nb
    if get_cnt:
        bound = neq.sum(-1)
        bound = bound * valid_mask if valid_mask is not None else bound
    else:
        bound = neq.any(-1)
        bound = bound & valid_mask if valid_mask is not None else bound
    if get_plain:
        eq = (labels == neighbor_label) | ~valid_nb
        plain = eq.all(-1)
        plain = plain & valid_mask if valid_mask is not None else plain
        return bound, plain
    return bound
