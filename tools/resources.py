#!/usr/bin/env python3
"""kernel-resource-usage report (hipcc -Rpass-analysis=kernel-resource-usage 2> file) as one line per kernel:
   tools/resources.py FILE [name-substring]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
blocks = re.split(r'remark: Function Name: ', txt)[1:]
names = [b.split()[0] for b in blocks]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
for b, d in zip(blocks, dem):
    if pat not in d:
        continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    d = re.sub(r'fz::|\(anonymous namespace\)::', '', d)
    d = d.split('(')[0] if len(d) > 170 else d
    print('%-120s V %3s A %3s S %3s scr %4s occ %s LDS %6s' % (d[:120], g('VGPRs'), g('AGPRs'), g('TotalSGPRs'), g(r'ScratchSize \[bytes/lane\]'),
                                                 g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
