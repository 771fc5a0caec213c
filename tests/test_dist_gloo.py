"""CPU, gloo, world_size 2 and 8: the rank-sharding + all-gather contract of ms-nets_amd/dist.py.  The world-size-8 cases run
BASELINE.json configs #4 / #5's index arithmetic (n_total 32 -> 4 per rank, 16 -> 2 per rank) with the map shape scaled down."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q, hw=(3, 5)):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    import msnets_amd  # noqa: F401
    from msnets_amd import dist as msdist
    r, w, _ = msdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    mine = msdist.shard_indices(n_total, r, w)
    H, W = hw
    # map i carries i in every pixel plus a per-pixel ramp, so a mis-ordered or mis-strided gather cannot pass on corners alone
    ramp = torch.arange(H * W, dtype=torch.float32).view(H, W) / 1024.0
    local = torch.stack([torch.full((H, W), float(i)) + ramp for i in mine]) if mine else torch.zeros((0, H, W))
    out = msdist.gather_disparities(local, n_total)
    msdist.barrier()
    whole = bool(torch.equal(out, torch.arange(n_total, dtype=torch.float32).view(-1, 1, 1) + ramp))
    q.put((rank, mine, out[:, 0, 0].tolist(), tuple(out.shape), whole))
    torch.distributed.destroy_process_group()


def _run(n_total, world=2, hw=(3, 5)):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, n_total, q, hw)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return sorted(res)


def test_even_batch_round_robin_and_order():
    res = _run(4)
    assert res[0][1] == [0, 2] and res[1][1] == [1, 3]           # sample i -> rank i mod world
    for _, _, vals, shape, whole in res:
        assert shape == (4, 3, 5) and vals == [0.0, 1.0, 2.0, 3.0] and whole   # original order on every rank


def test_uneven_batch_is_padded_for_the_collective():
    res = _run(3)
    assert res[0][1] == [0, 2] and res[1][1] == [1]
    for _, _, vals, shape, whole in res:
        assert shape == (3, 3, 5) and vals == [0.0, 1.0, 2.0] and whole


@pytest.mark.parametrize("n_total,per_rank,hw", [(32, 4, (17, 30)),     # config #4: 32 pairs of 544x960 over 8 GPUs, maps /32
                                                 (16, 2, (12, 39)),     # config #5: 16 pairs of 384x1248 over 8 GPUs, maps /32
                                                 (13, None, (5, 7))])   # ragged: ranks 0-4 hold 2, ranks 5-7 hold 1 (padded)
def test_world_size_8_index_arithmetic(n_total, per_rank, hw):
    """The exact 8-rank sharding / padding / re-ordering the driver's 8-GPU bench would exercise (main_msnet.py:174's
    DataParallel share), over gloo on CPU.  No scaling curve is implied by this -- it is index arithmetic."""
    world = 8
    res = _run(n_total, world=world, hw=hw)
    assert [r[0] for r in res] == list(range(world))
    owned = []
    for rank, mine, vals, shape, whole in res:
        assert mine == list(range(rank, n_total, world))
        if per_rank is not None:
            assert len(mine) == per_rank
        assert shape == (n_total,) + hw and vals == [float(i) for i in range(n_total)] and whole
        owned += mine
    assert sorted(owned) == list(range(n_total))                 # every pair computed exactly once


def test_single_process_is_identity():
    import msnets_amd  # noqa: F401
    from msnets_amd import dist as msdist
    x = torch.rand(2, 3, 4)
    assert msdist.gather_disparities(x, 2) is x
    assert msdist.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]


def test_oversubscription_rule_per_launcher_env():
    """a multi-node launch that exports RANK / WORLD_SIZE / LOCAL_RANK but no LOCAL_WORLD_SIZE (srun, mpirun
    wrappers) must not be refused because the GLOBAL world size exceeds one node's device count."""
    from msnets_amd.dist import oversubscribed
    # torchrun on one 8-GPU node, and two ranks forced onto one device
    assert oversubscribed({"LOCAL_WORLD_SIZE": "8"}, 7, 8) == (False, 8)
    assert oversubscribed({"LOCAL_WORLD_SIZE": "2"}, 1, 1) == (True, 2)
    # srun over 4 nodes x 8 GPUs: WORLD_SIZE = 32, no LOCAL_WORLD_SIZE; every local rank 0..7 has its own device
    for lr in range(8):
        assert oversubscribed({"WORLD_SIZE": "32", "RANK": str(8 + lr), "LOCAL_RANK": str(lr)}, lr, 8)[0] is False
    # the same launcher with more local ranks than devices is still caught, by the rank that has no device
    assert oversubscribed({"WORLD_SIZE": "32"}, 8, 8)[0] is True


def test_devices_remapped_rule():
    from msnets_amd.dist import devices_remapped
    assert devices_remapped({}) is False
    assert devices_remapped({"ROCR_VISIBLE_DEVICES": "0", "HIP_VISIBLE_DEVICES": "0"}) is False       # this pool's boxes
    assert devices_remapped({"HIP_VISIBLE_DEVICES": "0,1,2,3"}) is False
    assert devices_remapped({"HIP_VISIBLE_DEVICES": "3"}) is True
    assert devices_remapped({"ROCR_VISIBLE_DEVICES": "1,0"}) is True
    assert devices_remapped({"CUDA_VISIBLE_DEVICES": "GPU-deadbeef"}) is True
