"""The gather (cbl_queryandgroup, row a3) at N = 200 000, K = 16, C = 64 — 857 MB per launch, past the Infinity Cache — alone in a process:
    python tools/gather_200k.py        -> the roofline entry bench.py embeds (bench.gather_200k), one JSON line
Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` its query_group_lds_pipe<16, 16> launches are ONLY the 200 000-point ones."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(bench.gather_200k()))
