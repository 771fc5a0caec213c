"""Host-side state of the aggregator modules (hipops.ModuleState): what keeps the drop-in safe under the reference's own
nn.DataParallel wrapper (/root/reference/main_msnet.py:174 -- one Python THREAD per replica, replicas are shallow copies that
share every non-tensor attribute) -- a replica refuses loudly, the activation arena in use is a per-thread notion, and the
per-device slots are shared by reference so no two copies can own the same buffers.  CPU only; the concurrent-forward twin
on the GPU is tests/test_gpu_aggregators.py::test_concurrent_forwards_from_two_threads."""
import threading

import pytest
import torch

import msnets_amd  # noqa: F401
from msnets_amd import hipops
from msnets_amd.gcnet_3dcnn import GCNet_CostVolumeAggre
from msnets_amd.psmnet_3dcnn import PSMNet_CostVolumeAggre


@pytest.mark.parametrize("make,shape", [(lambda: GCNet_CostVolumeAggre(32), (1, 8, 16, 16, 32)),
                                        (lambda: PSMNet_CostVolumeAggre(32), (1, 64, 8, 16, 16))])
def test_data_parallel_replica_is_refused_loudly(make, shape):
    m = make().eval()
    replica = m._replicate_for_data_parallel()            # what nn.parallel.replicate creates for every device
    replica._is_replica = True                            # (replicate() sets it on each copy)
    assert replica.__dict__["_state"] is m.__dict__["_state"]          # shared by reference, never duplicated
    with pytest.raises(RuntimeError, match="one process per GPU"):
        replica(torch.zeros(shape))


def test_arena_in_use_is_per_thread():
    a, b = hipops.Arena(), hipops.Arena()
    seen, gate = {}, threading.Barrier(2)

    def worker(name, arena):
        with hipops.use_arena(arena):
            gate.wait(10)                                 # both threads are inside their context at the same time
            seen[name] = hipops._tls.arena
            gate.wait(10)
        seen[name + "_after"] = hipops._tls.arena
    ts = [threading.Thread(target=worker, args=("a", a)), threading.Thread(target=worker, args=("b", b))]
    [t.start() for t in ts]
    [t.join(30) for t in ts]
    assert seen["a"] is a and seen["b"] is b and seen["a_after"] is None and seen["b_after"] is None
    assert hipops._tls.arena is None


def test_slots_are_per_device_and_reset_by_invalidate_plans():
    m = GCNet_CostVolumeAggre(32).eval()
    st = m.__dict__["_state"]
    s_cpu = st.slot(torch.device("cpu"))
    assert st.slot(torch.device("cpu")) is s_cpu and st.slot(torch.device("cuda", 0)) is not s_cpu
    assert st.slot(torch.device("cuda", 0)) is not st.slot(torch.device("cuda", 1))
    assert m._arena is s_cpu.arena and m._forced_precision is None and m._graphs == {}
    m._forced_precision = "fp32"
    assert s_cpu.forced_precision == "fp32"
    m.invalidate_plans()
    assert m._forced_precision is None and m._arena is not s_cpu.arena


def test_graph_cache_drops_stale_parameter_states_and_is_bounded_in_bytes():
    """hipops._graphed_forward's eviction rule, exercised on the dict it manages (no GPU needed for the rule itself)."""
    class FakeArena:
        def __init__(self, n):
            self.n = n

        def nbytes(self):
            return self.n
    graphs = {("in", 1, "split-fp16", "old"): {"graph": object(), "arena": FakeArena(6 << 30)},
              ("in", 2, "split-fp16", "new"): {"graph": object(), "arena": FakeArena(20 << 30)},
              ("in", 3, "split-fp16", "new"): {"graph": object(), "arena": FakeArena(20 << 30)}}
    hipops._evict_graphs(graphs, "new")
    assert list(graphs) == [("in", 3, "split-fp16", "new")]           # the stale state first, then oldest until under the byte bound


def test_module_survives_deepcopy_and_pickle():
    """copy.deepcopy(model) and torch.save(model) (whole-module pickling) are common in training scripts: the device state holds
    locks and device buffers, so a copy gets fresh, empty state of its own instead of failing on the lock."""
    import copy
    import io
    m = GCNet_CostVolumeAggre(32).eval()
    m._forced_precision = "fp32"
    c = copy.deepcopy(m)
    assert c.__dict__["_state"] is not m.__dict__["_state"] and c._forced_precision is None and c._graphs == {}
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), c.state_dict().values()))
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    r = torch.load(buf, weights_only=False)
    assert r._forced_precision is None and isinstance(r.__dict__["_state"], hipops.ModuleState)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), r.state_dict().values()))
