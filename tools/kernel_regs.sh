#!/bin/bash
# register / occupancy summary of one HIP source of the library: tools/kernel_regs.sh contrastboundary_amd/csrc/pt_layer.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -Iinclude -Icontrastboundary_amd/csrc -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
name=None; row={}
for l in sys.stdin:
    if ' error' in l or 'warning:' in l: print(l.rstrip())
    m=re.search(r'Function Name: (\S+)',l)
    if m:
        name=subprocess.run(['c++filt',m.group(1)],capture_output=True,text=True).stdout.strip().replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        row={}
    for key,tag in (('    VGPRs: ','vgpr'),('VGPRs Spill: ','spill'),('LDS Size [bytes/block]: ','lds'),('Occupancy [waves/SIMD]: ','occ'),('AGPRs: ','agpr'),('ScratchSize [bytes/lane]: ','scratch')):
        if key in l: row[tag]=int(l.split(key)[1].split()[0])
    if 'LDS Size' in l: print('%-52s vgpr %3d agpr %3d spill %3d occ %d lds %6d scratch %d' % (name[:52],row.get('vgpr',0),row.get('agpr',0),row.get('spill',0),row.get('occ',0),row.get('lds',0),row.get('scratch',0)))
"
