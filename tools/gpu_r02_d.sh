#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for w in ifg ift ofg oft ibt obt; do timeout 120 python tools/exp/capture_probe.py $w 2>&1 | grep -v "^  File \"/usr/lib\|amdgpu.ids\|Extension modules" | head -14 | cut -c1-200; done
