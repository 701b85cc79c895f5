"""osrl_amd -- the OSRL offline safe-RL training hot path, native to AMD Instinct MI355X (gfx950).

Same Trainer / model API surface as liuzuxin/OSRL (``osrl.algorithms``); the per-step arithmetic is
hand-written HIP in ``libosrl_amd.so`` (C ABI: include/osrl_amd.h).  See DESIGN.md.
"""
__version__ = "0.1.0"

import os as _os

# Kernel arguments in DEVICE memory.  The HIP runtime decides per box where launches' argument blocks live; in host
# memory every wave of every launch reads them over PCIe and the CPQ step measured 1690 instead of 2150 steps/s
# (profiles/r3_kernarg_ab.txt -- the driver's round-2 number to 0.4 %).  The runtime reads the switch when it
# initialises (the process's first HIP call), so it is set at import, before any device work; an explicit setting by the
# operator wins.  Independently of it, the fused-MLP launches of a captured step keep their descriptors in HBM
# themselves (engine/core.py ArgArena), which is what still helps where the runtime cannot honour the switch.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
