#!/usr/bin/env python3
"""Loads vs wait groups per kernel in a gfx950 assembly listing: how many memory round trips does a kernel's straight
line pay for?  A kernel with ~as many ``s_waitcnt vmcnt`` wait points as ``global_load``s waits for each load on its own
(the compiler sank the loads into the branches that consume them); one whose loads are requested together shows a
handful of wait groups.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -sink-common-insts=false -Iinclude -S --cuda-device-only \
          osrl_amd/csrc/glue.hip -o /tmp/glue.s && python tools/isa_loads.py /tmp/glue.s

tests/test_isa_cpu.py holds the step's latency-chain kernels to their wait-group counts (a compiler that turns the
selects back into branches shows up there, not as a slower bench three rounds later).
"""
import re
import sys
from typing import Dict


def scan(path: str) -> Dict[str, Dict[str, int]]:
    """{mangled kernel name: {loads, waits, scratch, vgprs, b128_writes}} of an assembly listing."""
    kern, res = None, {}
    for ln in open(path):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            kern = m.group(1)
            res[kern] = dict(loads=0, waits=0, pend=0, scratch=0, vgprs=0, b128_writes=0)
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
        elif 'ds_write_b128' in ln:
            r['b128_writes'] += 1
        else:
            m = re.search(r';\s*ScratchSize:\s*(\d+)', ln)
            if m:
                r['scratch'] = int(m.group(1))
            m = re.search(r';\s*TotalNumVgprs:\s*(\d+)', ln)
            if m:
                r['vgprs'] = int(m.group(1))
    return res


if __name__ == "__main__":
    for k, r in scan(sys.argv[1]).items():
        print(f"{r['loads']:4d} loads {r['waits']:4d} wait-groups  {k[:100]}")
