"""Host mirror of the reference's segmentation network and criterion, /root/reference/pytorch/model/pointtransformer_seg.py:
    Loss :15-25, PointTransformerSeg :27-143, pointtransformer_seg_repro :146-150, with MultiHead (model/heads.py:13-60) and
    MLP (model/blocks.py:157-189).
Same class names, constructor arguments, sub-module names and construction ORDER (state_dicts are interchangeable, and a model
built under the same torch.manual_seed has the same initial parameters as the reference's), same `forward(inputs) -> (logits,
stage_list)` contract, so the reference's tool/train.py can use it as `model` / `criterion`.

What changes is the neighbourhood work: every block is the fused mirror of blocks.py, and one forward + criterion runs inside
`pointops.neighbor_cache()` when `forward_and_loss` is used — the 5 self-KNNs (one per stage, shared by all encoder and decoder
blocks of the stage), 4 down-sampling KNNs, 4+4 interpolation KNNs and the CBL head's 5+4 searches are each computed once
(57+ launches in the reference).
"""
import torch
import torch.nn as nn

from . import pointops
from .blocks import PointTransformerBlock, TransitionDown, TransitionUp
from .heads import ContrastHead, parse_stage


class Config(dict):
    """attribute-style dict like the reference's util.config.CfgNode (config.py:9-34): `cfg.key`, `'key' in cfg`, nested dicts"""

    def __init__(self, init=None, **kw):
        super().__init__()
        for k, v in dict(init or {}, **kw).items():
            self[k] = Config(v) if type(v) is dict else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value


def get_ftype(ftype):
    """model/utils.py:59-69"""
    if ftype in ["out", "fout", "f_out", "latent", "logits", "probs"]:
        return ("f_out" if ftype in ["out", "fout"] else ftype), "p_out"
    if ftype in ["sample", "fsample", "f_sample"]:
        return ("f_sample" if ftype in ["sample", "fsample"] else ftype), "p_sample"
    raise KeyError(f"not supported ftype = {ftype}")


def fetch_pxo(stage_n, stage_i, stage_list, ftype):
    stage = stage_list[stage_n][stage_i]
    return stage["p_out"], stage[ftype], stage["offset"]


class MLP(nn.Module):
    """f_out -> latent / logits (blocks.py:157-189; the `*_ops` string-configured variants are not part of the shipped configs)"""
    fkey_to_dims = None

    def __init__(self, fdim, head_cfg, config, fkey, drop=None):
        super().__init__()
        fkey = get_ftype(fkey)[0]
        valid_fkey = {"latent": config.base_fdim, "logits": config.num_classes}
        assert fkey in valid_fkey
        if MLP.fkey_to_dims is None:
            MLP.fkey_to_dims = valid_fkey
        for key in ("latent_ops", "logits_ops"):
            if key in head_cfg and head_cfg[key]:
                raise NotImplementedError(f"{key}: string-configured MLPs (blocks.py:191-240) are not mirrored")
        d_out = valid_fkey["latent"]
        infer_list = [nn.Linear(fdim, d_out), nn.BatchNorm1d(d_out), nn.ReLU(inplace=True)]
        if fkey == "logits":
            infer_list += [nn.Linear(d_out, valid_fkey["logits"])]
        self.infer = nn.Sequential(*infer_list)

    def forward(self, stage, k):
        return self.infer(stage[k])


class MultiHead(nn.Module):
    """heads.py:13-60: per-stage MLP to the latent, nearest-neighbour upsampling to stage 0, concat, classifier"""

    def __init__(self, fdims, head_cfg, config):
        super().__init__()
        self.head_cfg = head_cfg
        self.ftype = get_ftype(head_cfg.ftype)[0]
        infer_list, ni_list = nn.ModuleList(), []
        for n, i in parse_stage(head_cfg.stage, config.num_layers):
            infer_list.append(MLP(fdims[i], head_cfg, config, self.ftype))
            ni_list.append((n, i))
        self.infer_list, self.ni_list = infer_list, ni_list
        if not head_cfg.combine.startswith("concat"):
            raise ValueError(f"not supported {head_cfg.combine}")
        fdim = MLP.fkey_to_dims[head_cfg.ftype] * len(ni_list)
        k = config.num_classes
        if head_cfg.combine.endswith("mlp"):
            d = config.base_fdim
            self.cls = nn.Sequential(nn.Linear(fdim, d), nn.BatchNorm1d(d), nn.ReLU(inplace=True), nn.Linear(d, k))
        else:
            self.cls = nn.Linear(fdim, k)

    def upsample(self, stage_n, stage_i, stage_list):
        p, x, o = fetch_pxo(stage_n, stage_i, stage_list, self.ftype)
        if stage_i == 0:
            return x
        p0, _, o0 = fetch_pxo("up", 0, stage_list, self.ftype)
        return pointops.interpolation(p, p0, x.contiguous(), o, o0, k=1)           # :50

    def forward(self, stage_list):
        collect_list = []
        for (n, i), func in zip(self.ni_list, self.infer_list):
            stage_list[n][i][self.ftype] = func(stage_list[n][i], "f_out")         # :56-57
            collect_list.append(self.upsample(n, i, stage_list))
        return self.cls(torch.cat(collect_list, 1)), stage_list


class Loss(nn.Module):
    """pointtransformer_seg.py:15-25: cross entropy + the CBL losses, stacked (1 + stages,)"""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.contrast_head = ContrastHead(config.contrast, config) if "contrast" in config else None
        self.xen = nn.CrossEntropyLoss(ignore_index=config.ignore_label)

    def forward(self, output, target, stage_list):
        loss_list = [self.xen(output, target)]
        if self.contrast_head is not None:
            loss_list += self.contrast_head(output, target, stage_list)
        return torch.stack(loss_list)


class PointTransformerSeg(nn.Module):
    def __init__(self, block, blocks, c=6, k=13, config=None):
        super().__init__()
        self.c = c
        self.in_planes = c
        config = config if config is not None else Config()
        if "planes" not in config:
            config.planes = [32, 64, 128, 256, 512]
        planes = config.planes
        if "share_planes" not in config:
            config.share_planes = 8
        share_planes = config.share_planes
        stride, nsample = [1, 4, 4, 4, 4], [8, 16, 16, 16, 16]
        if "stride" not in config:
            config.stride = stride
        if "nsample" not in config:
            config.nsample = nsample
        self.enc1 = self._make_enc(block, planes[0], blocks[0], share_planes, stride=stride[0], nsample=nsample[0])
        self.enc2 = self._make_enc(block, planes[1], blocks[1], share_planes, stride=stride[1], nsample=nsample[1])
        self.enc3 = self._make_enc(block, planes[2], blocks[2], share_planes, stride=stride[2], nsample=nsample[2])
        self.enc4 = self._make_enc(block, planes[3], blocks[3], share_planes, stride=stride[3], nsample=nsample[3])
        self.enc5 = self._make_enc(block, planes[4], blocks[4], share_planes, stride=stride[4], nsample=nsample[4])
        self.dec5 = self._make_dec(block, planes[4], 2, share_planes, nsample=nsample[4], is_head=True)
        self.dec4 = self._make_dec(block, planes[3], 2, share_planes, nsample=nsample[3])
        self.dec3 = self._make_dec(block, planes[2], 2, share_planes, nsample=nsample[2])
        self.dec2 = self._make_dec(block, planes[1], 2, share_planes, nsample=nsample[1])
        self.dec1 = self._make_dec(block, planes[0], 2, share_planes, nsample=nsample[0])
        self.head = self.cls = None
        self.config = config
        config.num_layers = 5
        config.num_classes = k
        if "multi" in config:
            self.head = MultiHead(planes, config.multi, config)
        else:
            self.cls = nn.Sequential(nn.Linear(planes[0], planes[0]), nn.BatchNorm1d(planes[0]), nn.ReLU(inplace=True), nn.Linear(planes[0], k))

    def _make_enc(self, block, planes, blocks, share_planes=8, stride=1, nsample=16):
        layers = [TransitionDown(self.in_planes, planes * block.expansion, stride, nsample)]
        self.in_planes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.in_planes, self.in_planes, share_planes, nsample=nsample))
        return nn.Sequential(*layers)

    def _make_dec(self, block, planes, blocks, share_planes=8, nsample=16, is_head=False):
        layers = [TransitionUp(self.in_planes, None if is_head else planes * block.expansion)]
        self.in_planes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.in_planes, self.in_planes, share_planes, nsample=nsample))
        return nn.Sequential(*layers)

    def forward(self, inputs):
        p0, x0, o0 = inputs["points"], inputs["features"], inputs["offset"]
        if self.c == 3:
            x0 = p0
        elif self.c == 6:
            x0 = torch.cat((p0, x0), 1)
        elif self.c == 7:
            x0 = torch.cat((torch.ones_like(p0[..., :1]), p0, x0), 1)
        else:
            raise ValueError(f"in_feature_dims c={self.c}")
        stage_list = {"inputs": inputs}
        p1, x1, o1 = self.enc1([p0, x0, o0])
        p2, x2, o2 = self.enc2([p1, x1, o1])
        p3, x3, o3 = self.enc3([p2, x2, o2])
        p4, x4, o4 = self.enc4([p3, x3, o3])
        p5, x5, o5 = self.enc5([p4, x4, o4])
        stage_list["down"] = [{"p_out": p, "f_out": x, "offset": o} for p, x, o in ((p1, x1, o1), (p2, x2, o2), (p3, x3, o3), (p4, x4, o4), (p5, x5, o5))]
        x5 = self.dec5[1:]([p5, self.dec5[0]([p5, x5, o5]), o5])[1]
        x4 = self.dec4[1:]([p4, self.dec4[0]([p4, x4, o4], [p5, x5, o5]), o4])[1]
        x3 = self.dec3[1:]([p3, self.dec3[0]([p3, x3, o3], [p4, x4, o4]), o3])[1]
        x2 = self.dec2[1:]([p2, self.dec2[0]([p2, x2, o2], [p3, x3, o3]), o2])[1]
        x1 = self.dec1[1:]([p1, self.dec1[0]([p1, x1, o1], [p2, x2, o2]), o1])[1]
        stage_list["up"] = [{"p_out": p, "f_out": x, "offset": o} for p, x, o in ((p1, x1, o1), (p2, x2, o2), (p3, x3, o3), (p4, x4, o4), (p5, x5, o5))]
        if self.head is not None:
            x, stage_list = self.head(stage_list)
        else:
            x = self.cls(x1)
        return x, stage_list


def pointtransformer_seg_repro(**kwargs):
    """pointtransformer_seg.py:146-150"""
    return PointTransformerSeg(PointTransformerBlock, [2, 3, 4, 6, 3], **kwargs)


def forward_and_loss(model, criterion, inputs, target):
    """one training-step forward: network + criterion under ONE neighbour cache -> (logits, stage_list, loss vector, cache)"""
    with pointops.neighbor_cache() as nc:
        output, stage_list = model(inputs)
        loss = criterion(output, target, stage_list)
    return output, stage_list, loss, nc
