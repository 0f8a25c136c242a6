"""Host mirror of the Point-Transformer blocks of the reference, /root/reference/pytorch/model/blocks.py:
    PointTransformerLayer :14-44 (vector attention over K neighbours), TransitionDown :47-77, TransitionUp :80-109,
    PointTransformerBlock :112-136.
Same class names, constructor arguments, parameter / sub-module names (state_dicts are interchangeable) and numbers.  What
changes is how the K-neighbour part runs:
  * ONE knnquery per layer (the reference runs two identical ones, blocks.py:34-35) — pass `idx=` to share it per stage;
  * relative coordinates come from the fused gather (cbl_queryandgroup with c = 0), `k_j - q_i` from the subtraction kernel
    (K7) and the final `sum_k (v_j + p_r) * w` from the aggregation kernel (K9, share_planes = c % w_c): the gathered
    (n,K,C) copies of x_k / x_v and the 4-d view/sum of blocks.py:43 are never materialised;
  * BatchNorm over (n*K) rows is applied on the flattened view (same statistics as the reference's transpose → BN1d →
    transpose, blocks.py:38,40, without the two transposed copies), in train mode by the two-pass kernels of csrc/bn_rows.hip with
    the following ReLU folded in (`dense.batch_norm`);
  * the four Linear layers that act on (n*K) rows with widths 3 / C/8 (linear_p, linear_w) run as streaming kernels
    (`dense.linear`, csrc/skinny_linear.hip): as library GEMMs they were half of a layer's time.
The per-point dense layers (q/k/v Linear, BatchNorm, ReLU, softmax over K) stay torch (rocBLAS / elementwise).
"""
import torch
import torch.nn as nn

from . import attention, dense, pointops, pt_layer


class PointTransformerLayer(nn.Module):
    def __init__(self, in_planes, out_planes, share_planes=8, nsample=16):
        super().__init__()
        self.mid_planes = mid_planes = out_planes // 1
        self.out_planes = out_planes
        self.share_planes = share_planes
        self.nsample = nsample
        self.linear_q = nn.Linear(in_planes, mid_planes)
        self.linear_k = nn.Linear(in_planes, mid_planes)
        self.linear_v = nn.Linear(in_planes, out_planes)
        self.linear_p = nn.Sequential(nn.Linear(3, 3), nn.BatchNorm1d(3), nn.ReLU(inplace=True), nn.Linear(3, out_planes))
        self.linear_w = nn.Sequential(nn.BatchNorm1d(mid_planes), nn.ReLU(inplace=True),
                                      nn.Linear(mid_planes, mid_planes // share_planes),
                                      nn.BatchNorm1d(mid_planes // share_planes), nn.ReLU(inplace=True),
                                      nn.Linear(out_planes // share_planes, out_planes // share_planes))
        self.softmax = nn.Softmax(dim=1)
        self.fused = True                                             # True: csrc/pt_layer.hip (C = 32 / 64, K = 8 / 16) else csrc/attention.hip where it applies; "split": attention.hip only; False: separate kernels

    def forward(self, pxo, idx=None) -> torch.Tensor:
        p, x, o = pxo                                                        # (n,3), (n,c), (b)
        if idx is None:
            idx = pointops.knn_indices(self.nsample, p, p, o, o)              # once, not twice (:34-35); the distances are not used here
        else:
            # a caller-supplied table (shared per stage) reaches the kernels as a raw pointer with K = idx.shape[1]: anything but (n, nsample) int32 rows
            # would be read as garbage row ids — fail here, loudly (a sliced table is made contiguous: values, not layout, are the contract)
            idx = pointops._req(idx.contiguous(), torch.int32, "idx", 2)
            if tuple(idx.shape) != (x.shape[0], int(self.nsample)) or idx.device != x.device:
                raise ValueError(f"idx: expected a ({x.shape[0]}, {int(self.nsample)}) table on {x.device}, got {tuple(idx.shape)} on {idx.device}")
        wide = self.fused and self.fused not in ("split", "ops") and pt_layer.supported_wide(self, x, idx, p)
        if wide and self.fused != "qkv3" and all(l.bias is not None and l.weight.shape == (self.out_planes, self.out_planes) for l in (self.linear_q, self.linear_k, self.linear_v)):
            # the wide stages: projections and everything behind them as one node (one batched product for q / k / v each way, then cbl_pt_layer_wide_*)
            return pt_layer.attention_wide_projected(self, p, x, idx)
        x_q, x_k, x_v = dense.triple_linear(x, self.linear_q, self.linear_k, self.linear_v)                        # :33, one launch per direction
        if self.fused and self.fused != "split" and pt_layer.supported(self, x, idx, p):
            # the two full-resolution shapes: everything behind the three projections as one pass structure (csrc/pt_layer.hip)
            return pt_layer.attention(self, p, x_q, x_k, x_v, idx)
        if wide:
            # fused = "qkv3": the three projections as separate Linear layers in front of the one call each way (cbl_pt_layer_wide_*: ~23 launches per layer
            # and pass pair instead of ~42); fused = "ops" keeps round 3's op-by-op issue of the same kernels reachable for A/B runs
            return pt_layer.attention_wide(self, p, x_q, x_k, x_v, idx)
        p_r = pointops.queryandgroup(self.nsample, p, p, p.new_zeros((p.shape[0], 0)), idx, o, o, use_xyz=True)   # (n,K,3) relative xyz
        if self.fused and attention.supported(self, x):
            # the C-wide part without its (n,K,C) tensors: only p1 (n,K,3) in and w2 (n,K,C/8) out exist (csrc/attention.hip)
            lin_pc, bn_c, lin_a = self.linear_p[3], self.linear_w[0], self.linear_w[2]
            p1 = dense.sequential(self.linear_p[:3], p_r).contiguous()        # Linear(3,3) -> BN -> ReLU, narrow
            w2 = attention.AttnW2.apply(x_q.contiguous(), x_k.contiguous(), p1, lin_pc.weight, lin_pc.bias, bn_c.weight, bn_c.bias, lin_a.weight, lin_a.bias,
                                        idx, bn_c, self.training)
            w = dense.sequential(self.linear_w[3:], w2)                       # BN -> ReLU -> Linear(C/8, C/8), narrow
            # softmax over K (:41) inside the aggregation kernels, forward and backward
            return attention.AttnAgg.apply(x_v.contiguous(), p1, lin_pc.weight, lin_pc.bias, w.contiguous(), idx, True)
        p_r = dense.sequential(self.linear_p, p_r)                            # :38  Linear(3,3) -> BN -> ReLU -> Linear(3,C) over (n*K) rows
        q_minus_k = pointops.subtraction(x_q.contiguous(), x_k.contiguous(), idx)      # x_q - x_k[idx]  (n,K,c)
        n, K, c = p_r.shape
        ratio = self.out_planes // self.mid_planes                            # 1 in every shipped configuration: the fold below is a no-op then
        pe = p_r if ratio == 1 else p_r.view(n, K, ratio, self.mid_planes).sum(2)
        w = pe - q_minus_k                                                    # x_k[idx] - x_q + p_r  (:39), one pass instead of negate + add
        w = dense.sequential(self.linear_w, w)                                # :40  BN -> ReLU -> Linear -> BN -> ReLU -> Linear
        w = self.softmax(w)                                                   # over K, :41
        return pointops.aggregation(x_v.contiguous(), p_r.contiguous(), w.contiguous(), idx)   # :42-43


class TransitionDown(nn.Module):
    def __init__(self, in_planes, out_planes, stride=1, nsample=16):
        super().__init__()
        self.stride, self.nsample = stride, nsample
        if stride != 1:
            self.linear = nn.Linear(3 + in_planes, out_planes, bias=False)
            self.pool = nn.MaxPool1d(nsample)
        else:
            self.linear = nn.Linear(in_planes, out_planes, bias=False)
        self.bn = nn.BatchNorm1d(out_planes)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, pxo):
        p, x, o = pxo
        if self.stride != 1:
            n_p, n_o, _ = pointops.fps_downsample(p, o, self.stride)                        # :61-68
            x = pointops.queryandgroup(self.nsample, p, n_p, x, None, o, n_o, use_xyz=True)   # (m,K,3+c) :69
            x = dense.batch_norm(dense.apply(self.linear, x), self.bn, relu=True)         # :70
            x = x.max(1)[0]                                                                 # MaxPool1d(nsample) over K, :71
            p, o = n_p, n_o
        else:
            x = dense.batch_norm(dense.apply(self.linear, x), self.bn, relu=True)         # :74
        return [p, x, o]


class TransitionUp(nn.Module):
    def __init__(self, in_planes, out_planes=None):
        super().__init__()
        if out_planes is None:
            self.linear1 = nn.Sequential(nn.Linear(2 * in_planes, in_planes), nn.BatchNorm1d(in_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, in_planes), nn.ReLU(inplace=True))
        else:
            self.linear1 = nn.Sequential(nn.Linear(out_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))
            self.linear2 = nn.Sequential(nn.Linear(in_planes, out_planes), nn.BatchNorm1d(out_planes), nn.ReLU(inplace=True))

    def forward(self, pxo1, pxo2=None):
        if pxo2 is None:
            _, x, o = pxo1                                                    # :91-103
            ends = pointops.host_offsets(o)
            x_tmp, s_i = [], 0
            for e_i in ends:
                cnt = e_i - s_i
                x_b = x[s_i:e_i, :]
                x_b = torch.cat((x_b, self.linear2(x_b.sum(0, True) / cnt).repeat(cnt, 1)), 1)
                x_tmp.append(x_b)
                s_i = e_i
            x = dense.sequential(self.linear1, torch.cat(x_tmp, 0))
        else:
            p1, x1, o1 = pxo1; p2, x2, o2 = pxo2                              # :105-108
            x = dense.sequential(self.linear1, x1) + pointops.interpolation(p2, p1, dense.sequential(self.linear2, x2).contiguous(), o2, o1)
        return x


class PointTransformerBlock(nn.Module):
    expansion = 1

    def __init__(self, in_planes, planes, share_planes=8, nsample=16):
        super().__init__()
        self.linear1 = nn.Linear(in_planes, planes, bias=False)
        self.bn1 = nn.BatchNorm1d(planes)
        self.transformer2 = PointTransformerLayer(planes, planes, share_planes, nsample)
        self.bn2 = nn.BatchNorm1d(planes)
        self.linear3 = nn.Linear(planes, planes * self.expansion, bias=False)
        self.bn3 = nn.BatchNorm1d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, pxo, idx=None):
        p, x, o = pxo
        identity = x
        x = dense.batch_norm(dense.apply(self.linear1, x), self.bn1, relu=True)
        x = dense.batch_norm(self.transformer2([p, x, o], idx), self.bn2, relu=True)
        x = dense.batch_norm(dense.apply(self.linear3, x), self.bn3, relu=True, residual=identity)     # :130-133 bn3, += identity, relu: one call each way
        return [p, x, o]


class MLPbyOps(nn.Module):
    """String-configured MLP of the reference, blocks.py:194-247 ('linear', 'linearbn', 'mlp', 'mlp2', ... joined by '-'): the projection
    in front of the contrast (heads.py:90-92) and the *_ops heads.  Same construction order and module names (`ops_func.<k>`), so a state_dict
    of the reference loads and a model built under the same seed has the same initial parameters; the dense layers run through dense.sequential."""
    mlp_kwargs = {"activation": "relu", "bias": True, "bn": True, "linear_bn": False}

    def __init__(self, ops, fdim, d_mid=None, d_out=None, **kwargs):
        super().__init__()
        import re
        ops_seq = ops.split("-") if "-" in ops else [ops]
        d_mid = d_mid if d_mid else fdim
        d_out = d_out if d_out else d_mid
        layers = []
        for op in ops_seq:
            assert "mlp" in op or op in ["linear", "linearbn"], f"invalid ops = {op}"
            kw = dict(self.mlp_kwargs); kw.update(kwargs)
            num = re.search(r"\d+", op)
            num = int(num.group()) if num else 1
            linear = "linear" in op or not op.endswith("mlp")        # linear / linearbn / mlp2: ends with a plain Linear (:217)

            def add(din, dout, kw):
                layers.append(nn.Linear(din, dout, bias=kw["bias"]))
                if kw["bn"]:
                    layers.append(nn.BatchNorm1d(dout))
                if kw["activation"] == "relu":
                    layers.append(nn.ReLU(inplace=True))
                elif kw["activation"] != "":
                    raise ValueError("not support activation = " + kw["activation"])
            for _ in range(num - 1):
                add(fdim, d_mid, kw)
                fdim = d_mid
            if linear:
                kw["activation"] = ""; kw["bn"] = False
            cur_out = d_out if op == ops_seq[-1] else d_mid
            add(fdim, cur_out, kw)
            fdim = cur_out
            if kw["linear_bn"] or "linearbn" in op:
                layers.append(nn.BatchNorm1d(fdim))
        self.ops_func = nn.Sequential(*layers)

    def forward(self, features):
        return dense.sequential(self.ops_func, features)
