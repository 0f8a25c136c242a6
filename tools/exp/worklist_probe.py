"""how many queries of a search go to the exact replay (reads the worklist counter out of the search workspace): python worklist_probe.py"""
import torch
from contrastboundary_amd import pointops, synthetic as S
n = 40960
xyz = torch.from_numpy(S.s_room(n, seed=0)[0]).cuda()
off = torch.tensor([n], dtype=torch.int32, device="cuda")
for K, algo in ((16, "reference"), (16, "set"), (16, "anytie"), (36, "reference"), (36, "set"), (8, "reference")):
    idx, d2 = pointops.knnquery_raw(K, xyz, xyz, off, off, algo=algo)
    torch.cuda.synchronize()
    ws = pointops._workspace(1, xyz.device)
    cnt = ws[256:260].view(torch.int32).item()
    dup = (d2[:, 1:] == d2[:, :-1]).any(1).sum().item()
    print(K, algo, "worklist", cnt, "queries with equal distances inside the list", dup)
