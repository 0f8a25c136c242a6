"""does the bench's graph capture work while an RCCL process group (with its watchdog thread) is alive?  (multi-rank runs)"""
import os, sys, json, io, contextlib, torch
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
sys.argv = ["bench.py", "--no-cpu-baseline"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    bench.main()
d = json.loads(buf.getvalue().strip().splitlines()[-1])
print(round(d["ms_per_step"], 4), d["config"]["issue"][:70])
