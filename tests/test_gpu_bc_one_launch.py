"""The BC step as ONE launch (include/osrl_amd.h osrl_mlp_regress_step, csrc/mlp.hip mlp_step_kernel) against the six
launches it replaces (osrl_step_begin + osrl_mlp_forward + osrl_mse_loss + osrl_mlp_backward_dz +
osrl_mlp_backward_dw_tiles + osrl_adam_step_packed): the same parameter, moment and packed-weight BITS after every step
(same MFMA chains, same summation orders, same element-wise Adam), the same minibatch rows drawn inside the launch, the
same step count and statistics ring; the loss statistic (summed per row tile, then in tile order) to 1e-6.
Oracle parity of the step itself (bc.py:45-55,103-109) is test_gpu_train_step.py's bc_c1 / bc_small golden cases, which
run through the one-launch path wherever the shape allows it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pair(od, ad, hidden, seed=3):
    from osrl_amd.algorithms import BC
    ms = []
    for _ in range(2):
        torch.manual_seed(seed)
        ms.append(BC(od, ad, 1.0, hidden, 50, device=DEV))
        ms[-1].setup_optimizers(1e-3)
    for (ka, va), (kb, vb) in zip(ms[0].state_dict().items(), ms[1].state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    return ms


def _store(od, ad, n=5000, seed=7):
    from osrl_amd.common.replay import ReplayStore, synthetic_transitions
    return ReplayStore(synthetic_transitions(n, od, ad, seed=seed), DEV, seed=11)


def _same_state(ea, eb, what):
    ga, gb = ea.model.groups["actor"], eb.model.groups["actor"]
    for name in ("p", "m", "v", "pf", "pb"):
        a, b = getattr(ga, name), getattr(gb, name)
        assert torch.equal(a, b), f"{what}: {name} differs in {(a != b).sum().item()} of {a.numel()} floats, " \
                                  f"max |d| {(a - b).abs().max().item():.3e}"
    assert torch.equal(ea.obs, eb.obs) and torch.equal(ea.act, eb.act), f"{what}: the minibatches differ"
    assert ea.st.device_step() == eb.st.device_step() == ea.st.host_step == eb.st.host_step
    assert torch.equal(ea.st.state[8:20], eb.st.state[8:20]), f"{what}: bias corrections / lr scale differ"
    la, lb = ea.st.read_stats()["loss/actor_loss"], eb.st.read_stats()["loss/actor_loss"]
    assert abs(la - lb) <= 1e-6 * max(abs(lb), 1e-3), (what, la, lb)
    assert not ea.one_launch_failed()


@pytest.mark.parametrize("B,hidden,od,ad,graph", [
    (256, [256, 256], 8, 2, False),    # BASELINE.json configs[0] (use_graph=False: the by-value kernel)
    (256, [256, 256], 8, 2, True),     # ... the default: direct launches, the descriptor lives in the argument arena
    (250, [256, 256], 17, 6, True),    # ragged last row tile, wider in / out layers
    (48, [400, 300], 11, 3, True),     # 25 / 19 column blocks: the <1, 4, 8> tile, ragged dW tiles
    (500, [256, 256, 256], 8, 2, True),  # three hidden layers, 32 row tiles vs 44 dW tiles
])
def test_one_launch_equals_the_six_launch_plan_with_replay(B, hidden, od, ad, graph):
    ma, mb = _pair(od, ad, hidden)
    ea, eb = ma.engine(B), mb.engine(B)
    assert ea.one_launch, "the shape should take the one-launch step"
    eb.one_launch = False
    store = _store(od, ad)
    ea.attach_replay(store)
    eb.attach_replay(store)
    for s in range(7):
        ea.step_replay(use_graph=graph)
        eb.step_replay(use_graph=graph)
        torch.cuda.synchronize()
        _same_state(ea, eb, f"step {s + 1}")
    assert ea.one_launch and ea.graph is None  # (a one-kernel step is launched directly, never captured)
    assert (eb.graph is not None) == graph
    if graph:  # ... with its descriptor in HBM: recorded by the first step, found by every later one
        assert ea._arena_direct.misses == 0 and ea._arena_direct.hits >= 1
    # statistics ring: every earlier step's loss was committed by the NEXT launch
    ra, rb = ea.st.read_stats_many(range(1, 8)), eb.st.read_stats_many(range(1, 8))
    for s in range(1, 8):
        assert abs(ra[s][0] - rb[s][0]) <= 1e-6 * max(abs(rb[s][0]), 1e-3), (s, ra[s], rb[s])
    # and the policy the actor reads afterwards is the same one
    o = torch.randn(5, od, device=DEV)
    assert torch.equal(ma.actor(o), mb.actor(o))


def test_one_launch_with_caller_batches_and_the_trainer_api():
    """train_one_step(observations, actions) (no replay store: n_fields = 0) through the trainer, eager and captured."""
    from osrl_amd.algorithms import BCTrainer
    from osrl_amd.common.logger import DummyLogger
    od, ad, B = 8, 2, 256
    for graph in (False, True):
        ma, mb = _pair(od, ad, [256, 256], seed=5)
        ta = BCTrainer(ma, None, DummyLogger(), actor_lr=1e-3, device=DEV, use_graph=graph, stats_mode="sync")
        tb = BCTrainer(mb, None, DummyLogger(), actor_lr=1e-3, device=DEV, use_graph=graph, stats_mode="sync")
        mb.engine(B).one_launch = False
        g = torch.Generator(device="cpu").manual_seed(1)
        for s in range(5):
            obs = torch.randn(B, od, generator=g).to(DEV)
            act = torch.rand(B, ad, generator=g).to(DEV) * 2 - 1
            ta.train_one_step(obs, act)
            tb.train_one_step(obs, act)
            torch.cuda.synchronize()
            _same_state(ma._engine, mb._engine, f"graph={graph} step {s + 1}")
        assert ma._engine.one_launch and not mb._engine.one_launch


def test_shapes_outside_the_fused_launch_keep_the_plan():
    """Narrow nets (4-wave tiles) and batches with several row splits per dW tile are refused by the library
    (OSRL_E_UNSUPPORTED) or by the engine, and train through the six launches as before."""
    from osrl_amd.algorithms import BC
    torch.manual_seed(0)
    m = BC(8, 2, 1.0, [32, 32], 50, device=DEV)   # cpw = 1: the 4-wave tile
    m.setup_optimizers(1e-3)
    e = m.engine(64)
    e.step(torch.randn(64, 8, device=DEV), torch.rand(64, 2, device=DEV), use_graph=False)
    torch.cuda.synchronize()
    assert not e.one_launch and e.st.device_step() == 1
    m2 = BC(8, 2, 1.0, [256, 256], 50, device=DEV)
    m2.setup_optimizers(1e-3)
    e2 = m2.engine(1024)                          # 4 row splits per tile: gradient slabs
    assert not e2.one_launch
    e2.step(torch.randn(1024, 8, device=DEV), torch.rand(1024, 2, device=DEV), use_graph=True)
    torch.cuda.synchronize()
    assert e2.st.device_step() == 1


def test_the_c_abi_refuses_bad_descriptors():
    import ctypes as C
    from osrl_amd import _lib as L
    k = L.MlpStepT()
    assert L.load().osrl_mlp_regress_step(C.byref(k), None) == -1
    assert L.load().osrl_mlp_regress_step(None, None) == -1


def test_one_launch_captured_in_a_graph_too(monkeypatch):
    """OSRL_BC_DIRECT=0: the one-kernel step inside a hipGraph (what an outer capture of several steps would do)."""
    monkeypatch.setenv("OSRL_LAB", "1")  # (lab switches are read only under OSRL_LAB=1: engine/plan.py)
    monkeypatch.setenv("OSRL_BC_DIRECT", "0")
    ma, mb = _pair(8, 2, [256, 256])
    ea, eb = ma.engine(256), mb.engine(256)
    assert ea.one_launch and not ea.direct
    eb.one_launch = False
    store = _store(8, 2)
    ea.attach_replay(store)
    eb.attach_replay(store)
    for s in range(4):
        ea.step_replay()
        eb.step_replay()
        torch.cuda.synchronize()
        _same_state(ea, eb, f"step {s + 1}")
    assert ea.graph is not None and ea._arena.misses == 0 and ea._arena.hits >= 1


def test_a_flagged_one_launch_step_is_reported_at_the_next_synchronisation_and_the_engine_falls_back():
    """the one-launch step's dW workgroups wait for the row tiles with a bounded poll; when it expires the
    kernel sets an error word and continues.  The word is now read wherever the host synchronises anyway (statistics
    read, device_step, evaluate, checkpoint): it raises once, clears the word, and the engine runs the six-launch plan
    from then on."""
    from osrl_amd import _lib as L
    from osrl_amd.common.checkpoint import checkpoint_state
    ma, _ = _pair(8, 2, [256, 256])
    ea = ma.engine(256)
    assert ea.one_launch
    ea.attach_replay(_store(8, 2))
    ea.step_replay()
    ea.st.read_stats()  # healthy: no exception
    ea.step_ws[L.STEP_MAX_WG + 2] = 1.0  # what the kernel writes when a workgroup gives up waiting
    with pytest.raises(RuntimeError, match="one-launch BC step"):
        ea.st.read_stats()
    assert not ea.one_launch and float(ea.step_ws.abs().sum()) == 0.0
    p0 = ma.groups["actor"].p.clone()
    ea.step_replay()  # the six-launch plan
    torch.cuda.synchronize()
    assert not torch.equal(p0, ma.groups["actor"].p)
    ea.st.read_stats()
    checkpoint_state(ma)  # (also a check point; healthy now)
