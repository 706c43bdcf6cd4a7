"""Product-side host topology compilation (sip_lqr_compile_topology, C ABI)
against the reference's integer KATs (tests/lqr_test.cpp:931-980, 452-464) and
against the oracle on random trees.  Host integer work: no GPU needed."""
import numpy as np
import pytest



@pytest.fixture(scope="module")
def topo():
    import __graft_entry__ as entry
    entry.build_hip()
    from sip_optimal_control_amd.tree import compile_topology
    return compile_topology


def test_five_node_kat(topo):
    """LQRTopology.CompilesMultiChildPreorderAndPostorder: exact arrays."""
    st, arr = topo(4, 0, [0, 0, 1, 1], [1, 2, 3, 4])
    assert st == 0
    assert arr["child_offsets"] == [0, 2, 4, 4, 4, 4]
    assert arr["child_edges"] == [0, 1, 2, 3]
    assert arr["preorder_nodes"] == [0, 1, 3, 4, 2]
    assert arr["postorder_nodes"] == [2, 4, 3, 1, 0]


def test_invalid_topologies(topo):
    assert topo(2, 0, [0, 0], [1, 1])[0] == 4            # two edges into node 1 (:452-464)
    assert topo(4, 0, [0, 0, 1, 4], [1, 2, 3, 3])[0] == 4  # disconnected (:955-967)
    assert topo(4, 0, [4, 0, 1, 1], [1, 2, 3, 4])[0] == 4  # cycle (:969-980)
    assert topo(2, 3, [0, 0], [1, 2])[0] == 4            # root out of range (lqr.cpp:572)
    assert topo(2, 0, [0, 1], [1, 1])[0] == 4            # self loop (lqr.cpp:581)
    assert topo(2, 0, [0, 5], [1, 2])[0] == 4            # id out of range
    assert topo(2, 0, None, None)[0] == 4                # null arrays (lqr.cpp:567)


def test_chain_is_identity_order(topo):
    T = 9
    st, arr = topo(T, 0, list(range(T)), list(range(1, T + 1)))
    assert st == 0
    assert arr["preorder_nodes"] == list(range(T + 1))
    assert arr["postorder_nodes"] == list(range(T, -1, -1))


@pytest.mark.parametrize("seed", range(5))
def test_random_trees_match_oracle(topo, oracle_lib, seed):
    rng = np.random.default_rng(seed)
    N = int(rng.integers(2, 40))
    perm = rng.permutation(N)  # random node labels, random root
    parents, children = [], []
    for k in range(1, N):
        parents.append(int(perm[rng.integers(0, k)]))
        children.append(int(perm[k]))
    order = rng.permutation(N - 1)
    parents = [parents[i] for i in order]
    children = [children[i] for i in order]
    root = int(perm[0])
    st, arr = topo(N - 1, root, parents, children)
    assert st == 0
    blocks = {k: [np.zeros((1, 1))] * (N if k in ("Q",) else N - 1) for k in ("Q", "M", "R", "A", "B")}
    blocks.update({k: [np.zeros(1)] * N for k in ("q", "c", "delta")})
    blocks["r"] = [np.zeros(1)] * (N - 1)
    lqr = oracle_lib.TreeLQR(parents, children, [1] * N, [1] * (N - 1), blocks, root=root)
    assert lqr.topology_status == 0
    assert lqr.topology_arrays() == arr
