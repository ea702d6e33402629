"""Round-3 GPU tests (through the C ABI): HIP-graph replay safety of the bez_walk / bez_orient goal draw, ...
All tests need a GPU: `pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task", ["bez_walk", "bez_orient"])
def test_walk_goal_draw_advances_under_graph_replay(task):
    """the goal of bez_walk / bez_orient is drawn per reset call (walk_env.py:570-575).  A captured HIP
    graph replays kernel arguments verbatim, so the draw must come from device state: capture ONE fused step, replay it with all
    envs flagged for reset, and the goal must follow the oracle's call-counter sequence step by step."""
    import torch
    from bez_isaacgym_amd import abi
    from tests.sim_adapter import SimAdapter
    from tests.test_tasks import make_cfg, oracle
    n = 96
    g = SimAdapter(make_cfg(n, task=task, seed=5))
    o = oracle(n, task=task, seed=5)
    act = torch.zeros(n * 18, device=g.dev)
    reset = g.sim.tensor(abi.TENSOR_RESET)
    zero = np.zeros((n, 18), np.float32)
    # two eager steps (as the PPO loop's warm-up), then capture
    for _ in range(2):
        reset.fill_(1); o.set_reset(np.ones(n, np.int64))
        g.sim.step(act); o.step(zero)
        np.testing.assert_array_equal(g.goal, o.goal)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        reset.fill_(1)
        g.sim.step(act)  # warm the stream
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            g.sim.step(act)
    o.set_reset(np.ones(n, np.int64)); o.step(zero)   # the warm-up step above
    # NOTE: capture itself does not execute the step
    seen = []
    for k in range(6):
        reset.fill_(1); o.set_reset(np.ones(n, np.int64))
        graph.replay(); torch.cuda.synchronize()
        o.step(zero)
        gg = g.goal
        np.testing.assert_array_equal(gg, o.goal)      # bit-exact: same Philox draw, same counter
        assert (gg == gg[0]).all()                       # one draw per call, shared by every env it resets
        seen.append(tuple(gg[0]))
    assert len(set(seen)) == 6, seen                     # a fresh goal on every replay
