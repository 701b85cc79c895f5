import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def nccl_world1():
    """ONE 1-rank NCCL (= RCCL) process group for the whole session.  The data-parallel GPU tests used to create and
    destroy their own; on some MI355X boxes a hipGraph replay later in the same process then crashes inside the runtime
    (round 3: reproduced with round 2's tree as well, 5 runs of 6 on one lease, never under a debugger) -- a training
    process initialises its group once and keeps it, and so does the test session now."""
    import torch
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
        created = True
    yield dist
    if created:
        torch.cuda.synchronize()
        dist.destroy_process_group()
