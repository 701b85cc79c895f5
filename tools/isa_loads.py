#!/usr/bin/env python3
"""Loads vs wait groups per kernel in a gfx950 assembly listing: how many memory round trips does a kernel's straight
line pay for?  A kernel with ~as many ``s_waitcnt vmcnt`` wait points as ``global_load``s waits for each load on its own
(the compiler sank the loads into the branches that consume them); one whose loads are requested together shows a
handful of wait groups.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -sink-common-insts=false -Iinclude -S --cuda-device-only \
          osrl_amd/csrc/glue.hip -o /tmp/glue.s && python tools/isa_loads.py /tmp/glue.s
"""
import re
import sys

txt = open(sys.argv[1]).read().split('\n')
kern, res = None, {}
for ln in txt:
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        kern = m.group(1)
        res[kern] = dict(loads=0, waits=0, pend=0)
        continue
    if kern is None:
        continue
    r = res[kern]
    if re.search(r'\b(global_load|buffer_load|flat_load)', ln):
        r['loads'] += 1
        r['pend'] += 1
    elif 's_waitcnt' in ln and 'vmcnt' in ln:
        if r['pend'] > 0:
            r['waits'] += 1
        r['pend'] = 0
for k, r in res.items():
    print(f"{r['loads']:4d} loads {r['waits']:4d} wait-groups  {k[:100]}")
