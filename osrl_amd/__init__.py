"""osrl_amd -- the OSRL offline safe-RL training hot path, native to AMD Instinct MI355X (gfx950).

Same Trainer / model API surface as liuzuxin/OSRL (``osrl.algorithms``); the per-step arithmetic is
hand-written HIP in ``libosrl_amd.so`` (C ABI: include/osrl_amd.h).  See DESIGN.md.
"""
__version__ = "0.1.0"
