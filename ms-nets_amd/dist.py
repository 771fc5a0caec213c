"""Multi-GPU data parallelism for the forward path: one process per GPU, the batch of stereo pairs sharded
by rank, weights replicated per rank, and ONE collective -- an all-gather of the [n_local, H, W] fp32
disparity maps over RCCL/xGMI.  Replaces the reference's single-process nn.DataParallel
(/root/reference/main_msnet.py:174), whose scatter/replicate/gather it makes unnecessary: eval-mode forward
has no cross-sample state (BN uses running statistics, main_msnet.py:534).

Sharding rule: sample i -> rank i mod world_size.
On CPU-only hosts the same code runs over gloo (tests/test_dist_gloo.py, world_size 2)."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  A plain single-process run (no RANK) returns (0, 1, 0)."""
    if "RANK" not in os.environ:
        return 0, 1, 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        os.environ.setdefault("NCCL_DEBUG", "WARN")      # no RCCL version banner on stdout next to a caller's own output
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_total, rank, world):
    """Global sample indices owned by `rank`: i with i mod world == rank."""
    return list(range(rank, n_total, world))


def gather_disparities(local_disp, n_total):
    """local_disp [n_local, H, W] on every rank -> [n_total, H, W] in original sample order on every rank.
    Ranks with one sample fewer (n_total not divisible by world) are padded for the collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_disp                                      # plain single-process run: nothing to gather
    # (an initialised group of ONE rank still goes through the collective: that is how a 1-GPU box exercises the RCCL path)
    world = dist.get_world_size()
    n_max = (n_total + world - 1) // world
    _, H, W = local_disp.shape
    send = local_disp
    if local_disp.shape[0] < n_max:
        send = torch.zeros((n_max, H, W), dtype=local_disp.dtype, device=local_disp.device)
        send[:local_disp.shape[0]] = local_disp
    recv = torch.empty((world * n_max, H, W), dtype=local_disp.dtype, device=local_disp.device)
    dist.all_gather_into_tensor(recv, send.contiguous())      # rank-major concatenation along dim 0
    # recv.view(world, n_max)[r, j] is sample j*world + r
    return recv.view(world, n_max, H, W).permute(1, 0, 2, 3).reshape(world * n_max, H, W)[:n_total].contiguous()


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
