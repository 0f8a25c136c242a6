"""local_aggregation_oracle.py — TEST INFRASTRUCTURE ONLY.

numpy restatement of the TF-side local aggregation operators of the reference (value semantics of the TF1 graph ops):
    AdaptiveWeight   /root/reference/tensorflow/models/local_aggregation_operators.py:316-500  (shipped config
                     config/s3dis/adapt.yaml:19-26: local_input_feature='dp', fc_num=1, shared_channels=1,
                     weight_softmax=False, reduction='mean' incl. the padding-count quirk :466-470)
    PseudoGrid       ...:620-746  KPConv, depthwise: influence 'linear' / 'constant', mode 'sum' / 'closest'
    PosPool          ...:15-250   all runnable position embeddings, reductions sum / mean / max (before pool_bn / activation)
    ind_max_pool / ind_closest_pool   /root/reference/tensorflow/models/basic_operators.py:155-192
    tf_gather (shadow row)            ...:381-410

PARITY UNPINNED BY EXECUTION: TensorFlow is not installed in the build container (and the TF1.14 graph code cannot
run on any TF this image could hold), so these are pinned only by reading the reference, not by running it.  For
PseudoGrid the kernel point generator `create_kernel_points` (:669) and `radius_gaussian` (:702) are not defined
anywhere in the reference: kernel points are an INPUT here, and the gaussian influence is not restated.
Stage activations after these ops (batch norm, relu, 1x1 convs) are dense layers outside this path.
"""
import numpy as np


def gather_shadow(x, idx, shadow=0.0):
    """tf.gather(concat([x, shadow_row]), idx): index == len(x) selects the shadow row (basic_operators.py:381-410)"""
    x = np.asarray(x)
    if np.isscalar(shadow):
        row = np.full((1,) + x.shape[1:], shadow, x.dtype)
    else:
        row = np.asarray(shadow, x.dtype).reshape((1,) + x.shape[1:])
    return np.concatenate([x, row], 0)[idx]


def adaptive_weight(query_points, support_points, neighbors_indices, features, radius, fc_weight, fc_bias, reduction="mean"):
    """-> aggregation_feature (n, C) BEFORE batch norm / activation (:484).  fc_weight (3, C), fc_bias (C): the single
    batch_conv1d_1x1 of fc_num=1 (:426-430, with_bias, no activation, no bn)."""
    q = np.asarray(query_points, np.float32); s = np.asarray(support_points, np.float32)
    f = np.asarray(features, np.float32); idx = np.asarray(neighbors_indices)
    nf = gather_shadow(f, idx, 0.0)                                   # :360-362  (n,K,C)
    npnt = gather_shadow(s, idx, 0.0)                                 # :369-370
    rel = (npnt - q[:, None, :]) / np.float32(radius)                 # :371-373
    w = rel @ np.asarray(fc_weight, np.float32) + np.asarray(fc_bias, np.float32)      # (n,K,C)
    agg = (w * nf).sum(1, dtype=np.float32)                           # :457-464 with shared_channels=1
    if reduction in ("mean", "avg"):
        padding_num = idx.max()                                       # :466 — max over the WHOLE index tensor
        nn = (idx < padding_num).sum(-1, keepdims=True).astype(np.float32) + np.float32(1e-5)   # :467-470
        agg = agg / nn
    elif reduction != "sum":
        raise NotImplementedError(reduction)
    return agg.astype(np.float32)


def adaptive_weight_grads(query_points, support_points, neighbors_indices, features, radius, fc_weight, fc_bias, grad_out, reduction="mean"):
    """analytic gradients (float64) of sum(adaptive_weight * grad_out) w.r.t. features, fc_weight, fc_bias"""
    q = np.asarray(query_points, np.float64); s = np.asarray(support_points, np.float64)
    f = np.asarray(features, np.float64); idx = np.asarray(neighbors_indices); go = np.asarray(grad_out, np.float64)
    n0 = f.shape[0]
    nf = gather_shadow(f, idx, 0.0); npnt = gather_shadow(s, idx, 0.0)
    rel = (npnt - q[:, None, :]) / float(radius)
    w = rel @ np.asarray(fc_weight, np.float64) + np.asarray(fc_bias, np.float64)
    if reduction in ("mean", "avg"):
        nn = (idx < idx.max()).sum(-1, keepdims=True).astype(np.float64) + 1e-5
        go = go / nn
    g_nf = w * go[:, None, :]                                         # d/d nf
    g_w = nf * go[:, None, :]                                         # d/d w
    g_f = np.zeros((n0 + 1, f.shape[1])); np.add.at(g_f, idx.reshape(-1), g_nf.reshape(-1, f.shape[1]))
    g_fcw = np.einsum("nka,nkc->ac", rel, g_w)
    g_fcb = g_w.sum((0, 1))
    return g_f[:n0].astype(np.float32), g_fcw.astype(np.float32), g_fcb.astype(np.float32)


def kpconv(query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights, extent,
           influence="linear", mode="sum"):
    """PseudoGrid :681-728 -> output_features (n, C) before bn/activation.
    kernel_points (KP,3), kernel_weights (KP,C) [depthwise], extent = KP_extent * radius / density_parameter (:664)."""
    q = np.asarray(query_points, np.float32); s = np.asarray(support_points, np.float32)
    f = np.asarray(features, np.float32); idx = np.asarray(neighbors_indices)
    kp = np.asarray(kernel_points, np.float32)
    nb = gather_shadow(s, idx, 1e6) - q[:, None, :]                   # :681-684 (shadow point at 1e6)
    diff = nb[:, :, None, :] - kp[None, None, :, :]                   # :685-687  (n,K,KP,3)
    sq = (diff * diff).sum(-1, dtype=np.float32)                      # :688
    if influence == "constant":
        w = np.ones_like(sq)                                          # :693
    elif influence == "linear":
        w = np.maximum(np.float32(1) - np.sqrt(sq) / np.float32(extent), np.float32(0))     # :697
    else:
        raise NotImplementedError("gaussian influence: radius_gaussian is not defined in the reference")
    w = np.transpose(w, (0, 2, 1))                                    # (n,KP,K)
    if mode == "closest":
        nn1 = sq.argmin(2)                                            # :707  (n,K) closest kernel point per neighbour
        w = w * np.transpose(np.eye(kp.shape[0], dtype=np.float32)[nn1], (0, 2, 1))          # :708
    elif mode != "sum":
        raise ValueError(mode)
    nf = gather_shadow(f, idx, 0.0)                                   # :713-715
    wf = np.matmul(w, nf)                                             # :716  (n,KP,C)
    out = (np.asarray(kernel_weights, np.float32)[None] * wf).sum(1, dtype=np.float32)       # :723-727
    return out.astype(np.float32)


def kpconv_grads(query_points, support_points, neighbors_indices, features, kernel_points, kernel_weights, extent, grad_out,
                 influence="linear", mode="sum"):
    """analytic gradients (float64) w.r.t. features and kernel_weights"""
    q = np.asarray(query_points, np.float64); s = np.asarray(support_points, np.float64)
    f = np.asarray(features, np.float64); idx = np.asarray(neighbors_indices); go = np.asarray(grad_out, np.float64)
    kp = np.asarray(kernel_points, np.float64); kw = np.asarray(kernel_weights, np.float64)
    n0 = f.shape[0]
    nb = gather_shadow(s, idx, 1e6) - q[:, None, :]
    diff = nb[:, :, None, :] - kp[None, None]
    sq = (diff * diff).sum(-1)
    w = np.ones_like(sq) if influence == "constant" else np.maximum(1 - np.sqrt(sq) / float(extent), 0)
    w = np.transpose(w, (0, 2, 1))
    if mode == "closest":
        w = w * np.transpose(np.eye(kp.shape[0])[sq.argmin(2)], (0, 2, 1))
    nf = gather_shadow(f, idx, 0.0)
    wf = np.matmul(w, nf)
    g_kw = np.einsum("nc,npc->pc", go, wf)
    g_nf = np.einsum("npk,pc,nc->nkc", w, kw, go)
    g_f = np.zeros((n0 + 1, f.shape[1])); np.add.at(g_f, idx.reshape(-1), g_nf.reshape(-1, f.shape[1]))
    return g_f[:n0].astype(np.float32), g_kw.astype(np.float32)


def ind_max_pool(x, inds):
    """basic_operators.py:155-172: shadow row = column-wise min of x, max over each row of inds"""
    x = np.asarray(x, np.float32)
    return gather_shadow(x, inds, x.min(0)).max(1)


def ind_closest_pool(x, inds):
    """basic_operators.py:175-192: first column only, shadow row = zeros"""
    return gather_shadow(np.asarray(x, np.float32), np.asarray(inds)[:, 0], 0.0)


def pospool(query_points, support_points, neighbors_indices, features, radius, position_embedding="sin_cos", reduction="mean"):
    """PosPool aggregation_feature (n, C) BEFORE pool_bn / activation / output_conv (local_aggregation_operators.py:55-249)."""
    q = np.asarray(query_points, np.float32); s = np.asarray(support_points, np.float32)
    f = np.asarray(features, np.float32); idx = np.asarray(neighbors_indices)
    n, K = idx.shape
    n0, fdim = f.shape
    nf = gather_shadow(f, idx, 0.0)                                   # :60-62
    rel = (gather_shadow(s, idx, 0.0) - q[:, None, :]) / np.float32(radius)       # :65-70
    dist = np.sqrt((rel * rel).sum(2, keepdims=True, dtype=np.float32)).astype(np.float32)     # :71
    direction = (rel / (dist + np.float32(1e-6))).astype(np.float32)  # :72
    x, y, z = rel[:, :, :1], rel[:, :, 1:2], rel[:, :, 2:3]
    pe = position_embedding
    if pe == "one":
        geo, mid = np.ones_like(dist), 1
    elif pe == "xyz":
        geo, mid = rel, 3
    elif pe == "distance":
        geo, mid = dist, 1
    elif pe == "exp_-d":
        geo, mid = np.exp(-dist).astype(np.float32), 1
    elif pe in ("direction_exp_-d", "direction_d"):
        d = np.exp(-dist).astype(np.float32) if pe == "direction_exp_-d" else dist
        if fdim <= 18:
            geo, mid = np.concatenate([direction, d, direction, d, d], -1), 9
        else:
            geo, mid = np.concatenate([direction, d], -1), 4
    elif pe == "sin_cos":
        fd = 1 if fdim == 9 else fdim // 6
        dim_mat = np.power(np.float32(1000.0), np.float32(1.0 / fd) * np.arange(fd, dtype=np.float32)).astype(np.float32)
        div = (np.float32(100.0) * rel)[..., None] / dim_mat          # (n,K,3,fd)
        emb = np.concatenate([np.sin(div), np.cos(div)], -1).astype(np.float32).reshape(n, K, 6 * fd)
        geo = np.concatenate([emb, rel], -1) if fdim == 9 else emb
        mid = fdim
    elif pe in ("two_order", "three_order"):
        second = [rel, x * y, x * z, y * z, x * x, y * y, z * z]
        if pe == "two_order" or fdim == 9:
            geo, mid = np.concatenate(second, -1), 9
        else:
            xx, yy, zz = x * x, y * y, z * z
            third = [np.power(x, 3), np.power(y, 3), np.power(z, 3), xx * y, xx * z, yy * x, yy * z, zz * x, zz * y]
            geo, mid = np.concatenate(second + third, -1), 18
    else:
        raise NotImplementedError(pe)
    shared = fdim // mid
    if mid * shared != fdim or geo.shape[-1] != mid:
        raise ValueError("feature dim %d does not fit position embedding %s" % (fdim, pe))
    agg = (geo.astype(np.float32)[..., None] * nf.reshape(n, K, mid, shared)).reshape(n, K, fdim)    # :227-231
    if reduction == "sum":
        out = agg.sum(1, dtype=np.float32)
    elif reduction in ("mean", "avg"):
        nn = (idx < idx.max()).sum(-1, keepdims=True).astype(np.float32) + np.float32(1e-5)         # :236-241
        out = agg.sum(1, dtype=np.float32) / nn
    elif reduction == "max":
        out = (agg + np.where(idx == n0, np.float32(-65535.0), np.float32(0.0))[..., None]).max(1)   # :243-249
    else:
        raise NotImplementedError(reduction)
    return out.astype(np.float32), geo.astype(np.float32), agg


def pospool_grad_features(query_points, support_points, neighbors_indices, features, radius, grad_out, position_embedding="sin_cos", reduction="mean"):
    """d sum(pospool * grad_out) / d features, float64 accumulation; 'max' as tf.reduce_max (equal maxima share the gradient)"""
    f = np.asarray(features, np.float32); idx = np.asarray(neighbors_indices); go = np.asarray(grad_out, np.float64)
    n, K = idx.shape
    n0, fdim = f.shape
    out, geo, agg = pospool(query_points, support_points, neighbors_indices, features, radius, position_embedding, reduction)
    mid = geo.shape[-1]
    gfull = np.repeat(geo.astype(np.float64), fdim // mid, axis=-1)    # (n,K,fdim): the factor of features[nbr, c]
    if reduction in ("mean", "avg"):
        go = go / ((idx < idx.max()).sum(-1, keepdims=True).astype(np.float64) + 1e-5)
    if reduction == "max":
        vals = agg + np.where(idx == n0, np.float32(-65535.0), np.float32(0.0))[..., None]
        sel = (vals == out[:, None, :]).astype(np.float64)
        coef = sel / sel.sum(1, keepdims=True) * go[:, None, :] * gfull
    else:
        coef = gfull * go[:, None, :]
    g = np.zeros((n0 + 1, fdim), np.float64)
    np.add.at(g, idx.reshape(-1), coef.reshape(-1, fdim))
    return g[:n0].astype(np.float32)
