"""one configuration of knnquery, for rocprofv3 --kernel-trace --stats: python knn_one.py <room|uniform> <K> <algo> [n]"""
import sys, torch
from contrastboundary_amd import pointops, hotpath, synthetic as S
name, K, algo = sys.argv[1], int(sys.argv[2]), sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 40960
xyz = torch.from_numpy(S.s_room(n, seed=0)[0] if name == "room" else S.s_uniform(n, seed=1)).cuda().float()
off = torch.tensor([n], dtype=torch.int32, device="cuda")
for _ in range(10):
    pointops.knnquery_raw(K, xyz, xyz, off, off, algo=algo)
torch.cuda.synchronize()
