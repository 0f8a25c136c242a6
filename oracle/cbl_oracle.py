"""cbl_oracle.py — TEST INFRASTRUCTURE ONLY.

numpy restatement of the reference's Contrastive Boundary Learning head (pytorch side) and boundary masks:
    get_subscene_label / get_subscene_features   /root/reference/pytorch/model/basic_operators.py:9-50
    ContrastHead.point_contrast                  /root/reference/pytorch/model/heads.py:185-246
        posmask_cnt :145-149, dist_l2 :116-119, contrast_softnn :151-165
    get_boundary_mask                            /root/reference/pytorch/model/basic_operators.py:69-97
    boundary-IoU evaluation                      /root/reference/pytorch/tool/test.py:392-417 + util/common_util.py:25-37
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


def point_contrast(features, labels, neighbor_idx_full, temperature=None, weight=0.1, grad=True, contrast="softnn"):
    """heads.py:185-246 for pos='cnt', dist='l2', contrast='softnn' (:151-165) or 'nce' (:167-183).
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
    if contrast == "nce":
        # :176-182: neg = sum of the negatives' exps; one term -log(exp_j / (exp_j + neg)) per POSITIVE pair, mean over all of them
        negs = (e * (1 - pm)).sum(-1, dtype=np.float32)
        terms = -np.log(e / (e + negs[:, None]))
        npos = pm.sum()
        loss = np.float32(terms[pm > 0].mean(dtype=np.float32) * np.float32(weight))
        if not grad:
            return loss, g, point_mask
        T = 1.0 if temperature is None else float(temperature)
        e64, pm64, N = e.astype(np.float64), pm.astype(np.float64), negs.astype(np.float64)[:, None]
        Q = (pm64 / (e64 + N)).sum(-1, keepdims=True)
        dl_dd = np.where(pm64 > 0, N / (e64 + N), -e64 * Q) / T * float(weight) / float(npos)
        coef = (dl_dd / dist.astype(np.float64))[:, :, None] * diff.astype(np.float64)
        g64 = np.zeros(f.shape, np.float64)
        np.add.at(g64, rows, coef.sum(1))
        np.add.at(g64, nbr[rows].reshape(-1), -coef.reshape(-1, f.shape[1]))
        return loss, g64.astype(np.float32), point_mask
    if contrast != "softnn":
        raise ValueError(contrast)
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


def intersection_and_union(output, target, K, ignore_index=255):
    """util/common_util.py:25-37"""
    output = np.asarray(output).reshape(-1).copy(); target = np.asarray(target).reshape(-1)
    output[target == ignore_index] = ignore_index
    inter = output[output == target]
    ai = np.histogram(inter, bins=np.arange(K + 1))[0]
    ao = np.histogram(output, bins=np.arange(K + 1))[0]
    at = np.histogram(target, bins=np.arange(K + 1))[0]
    return ai, ao + at - ai, at


def boundary_iou(pred, labels, neighbor_idx, num_classes, ignore_label=255):
    """tool/test.py:392-417: (i, u, t) of the boundary points and of the plain points"""
    labels = np.asarray(labels)
    bound, plain = boundary_mask(labels, labels[neighbor_idx], get_plain=True)
    return {"bound": intersection_and_union(np.asarray(pred)[bound], labels[bound], num_classes, ignore_label),
            "plain": intersection_and_union(np.asarray(pred)[plain], labels[plain], num_classes, ignore_label)}


# ---------------------------------------------------------------------------------------------------------------------------
# TF flavour — /root/reference/tensorflow/models/heads/head.py:462-807 (contrast_head with sample 'label', contrast 'softnn',
# dist 'l2'), get_scene_label_infer :25-49, get_neighbor_summary :117-131, calc_dist :180-195.
# PARITY UNPINNED BY EXECUTION: TF1 graph code, TensorFlow absent from the build container; restated from the source.
# ---------------------------------------------------------------------------------------------------------------------------
def tf_scene_label(point_labels, scene_neighbor, num_classes, reduction="max"):
    """labels of sub-sampled points from their stage-0 neighbours: tf_gather with shadow -1 -> one-hot (invalid -> zeros) -> sum;
    'max' -> argmax (first maximum) (n,), 'soft' -> sum / (#valid + 1e-12) (n, ncls)"""
    pl = np.concatenate([np.asarray(point_labels), [-1]])
    lab = pl[scene_neighbor]                                         # shadow index == len(point_labels) -> -1
    valid = lab >= 0
    onehot = np.zeros(lab.shape + (num_classes,), np.float32)
    r, c = np.nonzero(valid)
    onehot[r, c, lab[r, c]] = 1
    s = onehot.sum(1, dtype=np.float32)
    if reduction in ("max", "cnt"):
        return np.argmax(s, -1)
    return (s / (valid.sum(-1, keepdims=True).astype(np.float32) + _EPS)).astype(np.float32)


def tf_label_kl(soft_labels, neighbors):
    """KL(p_i || p_j) of the soft labels of every (centre, neighbour) pair, calc_dist(..., dist='kl') heads/head.py:189-191 as used by
    collect_labels :498-511: sum_c xlogy(p_i[c], p_i[c] / max(p_j[c], 1e-12)); a shadow neighbour gathers the zero row (shadow_fn=0, :505).
    soft_labels (N,ncls) f32, neighbors (m,ns) (self column already dropped) -> (m,ns) f32"""
    p = np.asarray(soft_labels, np.float32)
    N = len(p)
    ppad = np.concatenate([p, np.zeros((1, p.shape[1]), np.float32)])
    pi = p[:len(neighbors), None, :]
    pj = ppad[np.minimum(neighbors, N)]
    ratio = (pi / np.maximum(pj, np.float32(_EPS))).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        term = np.where(pi > 0, pi * np.log(ratio, where=ratio > 0, out=np.zeros_like(ratio)), np.float32(0.0)).astype(np.float32)   # xlogy(0, .) = 0
    return term.sum(-1, dtype=np.float32)


def tf_samples(labels, neighbors, m, sample="label", rand_idx=None, kl_threshold=None):
    """sample_labels (head.py:551-625) for radius neighbourhoods: -> sample_idx (m,S), pos_mask, neg_mask (m,S) bool BEFORE the point mask.
    sample = '-'-joined segments: 'label' | 'labelkl<thr>' (kl_threshold) -> the neighbour columns, positives mined from the labels (collect_labels
    :485-547) and the valid mask of :540-545; 'nn<k>' -> the first k neighbour columns, all positives (:564-566, :603-604); 'rand<n>[R]' -> the
    caller's draws rand_idx (one (m,n) array per rand segment, in order: the reference draws tf.random.uniform per cloud, :568-596, which nothing
    here can replay), all negatives (:605-606), with 'R' invalid where the draw is one of the point's neighbours (:611-615).  Segments without a mask
    of their own are valid everywhere — also where an 'nn' column is a shadow neighbour and where the centre's label is ignored (:616-617)."""
    N = len(labels)
    nbr = np.asarray(neighbors)[:, 1:]                               # exclude self-loop, :560
    rand_idx = list(rand_idx) if rand_idx is not None else []
    idxs, posnegs, valids = [], [], []
    for seg in sample.split("-"):
        if seg.startswith("label"):
            cur = nbr
            if kl_threshold is not None:
                posneg = tf_label_kl(labels, nbr) < np.float32(kl_threshold)      # :511
                valid = nbr < N                                      # mask_n of tf_gather(get_mask=bool), :505-509 (no ignored labels: mask_c is None)
            else:
                lab = np.concatenate([np.asarray(labels), [-1]])     # shadow label -1, :537
                nl = lab[np.minimum(nbr, N)]
                me = np.asarray(labels)[:m]
                posneg = me[:, None] == nl                           # :538
                valid = (nl >= 0) & (me[:, None] >= 0)               # :540-545
        elif seg.startswith("nn"):
            cur = nbr[:, :int(seg[2:])]
            posneg = np.ones(cur.shape, bool); valid = np.ones(cur.shape, bool)
        elif seg.startswith("rand"):
            n_neg = int("".join(ch for ch in seg[4:] if ch.isdigit()))
            cur = np.asarray(rand_idx.pop(0))
            assert cur.shape == (m, n_neg), (cur.shape, seg)
            posneg = np.zeros(cur.shape, bool)
            valid = (cur[:, :, None] != nbr[:, None, :]).all(-1) if "R" in seg else np.ones(cur.shape, bool)
        else:
            raise NotImplementedError(seg)                           # :598-599
        idxs.append(cur); posnegs.append(posneg); valids.append(valid)
    idx = np.concatenate(idxs, 1); posneg = np.concatenate(posnegs, 1); valid = np.concatenate(valids, 1)
    return idx, posneg & valid, ~posneg & valid                      # :621-627


def tf_contrast_terms64(f64, idx, pos_mask, neg_mask, rows, n_rows_pad, temperature, contrast, separate):
    """per-point loss terms of calc_loss_from_dist (head.py:729-795) in float64 — the function whose gradient tf_contrast states analytically
    (tests differentiate it numerically)"""
    fpad = np.concatenate([f64, np.zeros((n_rows_pad, f64.shape[1]))])
    diff = f64[rows][:, None, :] - fpad[np.minimum(idx[rows], len(fpad) - 1)]
    dist = np.sqrt(np.maximum((diff * diff).sum(-1), 1e-12))
    d = -dist / (1.0 if temperature is None else float(temperature))
    e = np.exp(d - d.max(-1, keepdims=True))
    pm, nm = pos_mask[rows].astype(np.float64), neg_mask[rows].astype(np.float64)
    P, Nn = (e * pm).sum(-1), (e * nm).sum(-1)
    if contrast == "nce":
        under = e + Nn[:, None] if separate else (P + Nn)[:, None]
        return -(np.log(e / under + 1e-12) * pm).sum(-1)
    return -np.log((P / np.maximum(Nn, 1e-12) if separate else P / (P + Nn)) + 1e-12)


def tf_contrast(features, labels, neighbors, temperature=None, weight=0.1, grad=True, kl_threshold=None, contrast="softnn", sample="label",
                rand_idx=None, separate=False):
    """features (m,d); labels (N,) hard labels of the support points (N >= m; negative = ignored) — or, with kl_threshold (sample
    'labelkl<thr>', :492-511), (N,ncls) soft labels: a neighbour is a positive if KL(p_centre || p_neighbour) < thr; neighbors (m,k)
    radius neighbours incl. self column, padded with N; sample / rand_idx: tf_samples; separate = margin 'S' (:759-760, :783-785).
    -> loss, d loss/d features (m,d), point_mask"""
    f = np.asarray(features, np.float32)
    N = len(labels)
    m = len(f)
    idx, pos_mask, neg_mask = tf_samples(labels, neighbors, m, sample, rand_idx, kl_threshold)
    point_mask = pos_mask.any(1) & neg_mask.any(1)                   # :629-640
    g = np.zeros_like(f)
    if not point_mask.any():
        return np.float32(0.0), g, point_mask                        # false_fn :662-665
    rows = np.nonzero(point_mask)[0]
    fpad = np.concatenate([f, np.zeros((max(N + 1 - len(f), 1), f.shape[1]), np.float32)])   # tf_gather shadow row = zeros, :703
    nb = np.minimum(idx[rows], len(fpad) - 1)
    fi = f[rows]; fj = fpad[nb]
    diff = fi[:, None, :] - fj
    dist = np.sqrt(np.maximum((diff * diff).sum(-1, dtype=np.float32), _EPS)).astype(np.float32)   # :184-185
    d = -dist
    if temperature is not None:
        d = (d / np.float32(temperature)).astype(np.float32)         # :750-751
    d = d - d.max(-1, keepdims=True)                                 # over ALL columns, :752
    e = np.exp(d).astype(np.float32)
    pm, nm = pos_mask[rows].astype(np.float32), neg_mask[rows].astype(np.float32)
    T = 1.0 if temperature is None else float(temperature)
    e64, pm64, nm64 = e.astype(np.float64), pm.astype(np.float64), nm.astype(np.float64)
    pos = (e * pm).sum(-1, dtype=np.float32); neg = (e * nm).sum(-1, dtype=np.float32)
    if contrast == "nce":
        # :773-795 without masking: -sum over positives of log(exp_j / under + eps); under = the sum of the valid exps, or with 'S' exp_j + the negatives
        under = e + neg[:, None] if separate else np.broadcast_to((e * (pm + nm)).sum(-1, dtype=np.float32)[:, None], e.shape)
        rr = e / under
        per_point = -(np.log(rr + _EPS) * pm).sum(-1, dtype=np.float32)
        r64, u64 = rr.astype(np.float64), under.astype(np.float64)
        if separate:                                                 # d term / d e_j of a positive, d term / d (sum of negatives)
            dl_de = -pm64 * (u64 - e64) / ((r64 + 1e-12) * u64 * u64)
            dl_de = dl_de + nm64 * (pm64 * e64 / ((r64 + 1e-12) * u64 * u64)).sum(-1, keepdims=True)
            dl_dd = -dl_de * e64 / T
        else:
            G = (pm64 * r64 / (r64 + 1e-12)).sum(-1, keepdims=True)
            dl_dd = (pm64 + nm64) * (pm64 * r64 / (r64 + 1e-12) - r64 * G) / T
    elif contrast == "softnn":
        ratio = pos / np.maximum(neg, _EPS) if separate else pos / (pos + neg)      # :759-762
        per_point = -np.log(ratio + _EPS)                            # :766-767
        P, Nn = (e64 * pm64).sum(-1), (e64 * nm64).sum(-1)
        if separate:
            Nc = np.maximum(Nn, 1e-12)
            R = P / Nc
            dr_de = pm64 / Nc[:, None] - nm64 * (np.where(Nn > 1e-12, P / (Nc * Nc), 0.0))[:, None]
        else:
            A = P + Nn
            R = P / A
            dr_de = (pm64 * A[:, None] - (pm64 + nm64) * P[:, None]) / (A * A)[:, None]
        dl_dd = e64 * dr_de / (T * (R + 1e-12)[:, None])
    else:
        raise NotImplementedError(contrast)
    loss = np.float32(per_point.mean(dtype=np.float32) * np.float32(weight))   # :805-806
    if not grad:
        return loss, g, point_mask
    dl_dd = dl_dd * (float(weight) / float(len(rows)))
    coef = (dl_dd / dist.astype(np.float64))[:, :, None] * diff.astype(np.float64)
    coef[dist <= 1e-6] = 0.0                                         # sqrt(max(s, 1e-12)) is flat below the clamp
    g64 = np.zeros((len(fpad), f.shape[1]), np.float64)
    np.add.at(g64, rows, coef.sum(1))
    np.add.at(g64, nb.reshape(-1), -coef.reshape(-1, f.shape[1]))
    return loss, g64[:len(f)].astype(np.float32), point_mask
