#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for g in 512 768 1024; do CBL_KB_GRID=$g timeout 120 python tools/exp/kb_probe.py 2>&1 | grep -v amdgpu.ids | head -1; done
timeout 120 python tools/exp/kb_probe.py 2>&1 | grep -v amdgpu.ids | head -1
