"""Multi-GPU data parallelism for the forward path: one process per GPU, the batch of stereo pairs sharded
by rank, weights replicated per rank, and ONE collective -- an all-gather of the [n_local, H, W] fp32
disparity maps over RCCL/xGMI.  Replaces the reference's single-process nn.DataParallel
(/root/reference/main_msnet.py:174), whose scatter/replicate/gather it makes unnecessary: eval-mode forward
has no cross-sample state (BN uses running statistics, main_msnet.py:534).

Sharding rule: sample i -> rank i mod world_size.
On CPU-only hosts the same code runs over gloo (tests/test_dist_gloo.py, world_size 2).

Backend: RCCL ("nccl") on GPUs by default.  `MSNET_DIST_BACKEND=gloo` (bench.py --dist-backend gloo) keeps the ranks' tensors
on the GPU and runs the collective through host memory -- RCCL refuses two ranks on one device, gloo does not, so a box with
ONE GPU can run world_size 2 with both ranks on cuda:0 (LOCAL_RANK modulo the device count) and execute every world > 1
branch on hardware (tests/test_gpu_bench_contract.py::test_bench_world2_on_one_gpu).  Diagnostic only: never the headline."""
import glob
import os

import torch
import torch.distributed as dist

_ranks_per_device = 1


def backend():
    """'nccl' | 'gloo' | None (no process group)."""
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def ranks_per_device():
    return _ranks_per_device


_affinity_before = []


def _gpu_cards():
    """amdgpu devices in /sys/class/drm, in card order (the order HIP enumerates them in on a stock node)."""
    out = []
    for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device"), key=lambda p: int(p.split("card")[-1].split("/")[0])):
        try:
            if open(os.path.join(c, "vendor")).read().strip() == "0x1002" and os.path.exists(os.path.join(c, "mem_info_vram_total")):
                out.append(c)
        except OSError:
            pass
    return out


def devices_remapped(env=None):
    """True if ROCR_/HIP_/CUDA_VISIBLE_DEVICES or GPU_DEVICE_ORDINAL re-number the node's GPUs, i.e. HIP ordinal i may not be the
    node's device i (sysfs card order, rocm_smi index).  A variable that lists the identity prefix "0,1,...,k-1" (what a
    one-GPU-per-container pool exports: "0") hides devices but re-numbers none."""
    env = os.environ if env is None else env
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL"):
        v = env.get(k)
        if v and [t.strip() for t in v.split(",")] != [str(i) for i in range(len(v.split(",")))]:
            return True
    return False


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa_node(local_rank):
    """Pin this process to the CPUs of the NUMA node its GPU hangs off (/sys/class/drm/card*/device/numa_node): on a
    two-socket 8-GPU node the launch thread of a rank otherwise migrates across sockets, and eight ranks' launch loops and
    pinned read-backs then cross the socket link.  Reads sysfs only -- call it BEFORE anything touches the GPU.  Returns the
    node bound to, or None (single-node host, no such file, visible-device remapping that cannot be resolved: nothing done)."""
    try:
        cards = _gpu_cards()
        # a visible-device REMAPPING (the variables compose) makes "HIP ordinal -> drm card" a guess: do nothing then
        if devices_remapped():
            return None
        if not cards or local_rank >= len(cards):
            return None
        node = int(open(os.path.join(cards[local_rank], "numa_node")).read().strip())
        if node < 0:
            return None
        cpus = _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
        before = os.sched_getaffinity(0)
        cpus &= before
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        _affinity_before.append(before)
        return node
    except (OSError, ValueError, IndexError):
        return None


def oversubscribed(env, local_rank, ndev):
    """(ranks of this node outnumber its devices?, ranks on this node as far as the launcher says).  LOCAL_WORLD_SIZE (torchrun
    exports it) counts the ranks of THIS node.  Launchers that export only RANK / WORLD_SIZE / LOCAL_RANK (srun, mpirun
    wrappers) say nothing about it -- WORLD_SIZE counts every node -- so there only LOCAL_RANK itself can show that a rank has
    no device of its own."""
    if "LOCAL_WORLD_SIZE" in env:
        lw = int(env["LOCAL_WORLD_SIZE"])
        return lw > ndev, lw
    return local_rank >= ndev, local_rank + 1


def restore_affinity():
    """Undo bind_to_gpu_numa_node (threads started afterwards -- a CPU baseline's OpenMP pool -- see every core again)."""
    if _affinity_before:
        try:
            os.sched_setaffinity(0, _affinity_before.pop())
        except OSError:
            pass


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank) -- local_rank already reduced modulo the number of visible GPUs, i.e. the device
    index to use.  A plain single-process run (no RANK) returns (0, 1, 0)."""
    global _ranks_per_device
    if "RANK" not in os.environ:
        return 0, 1, 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    backend = backend or os.environ.get("MSNET_DIST_BACKEND") or None
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
    if torch.cuda.is_available():
        ndev = torch.cuda.device_count()
        over, local_world = oversubscribed(os.environ, local, ndev)
        if over:
            if backend == "nccl":
                raise RuntimeError("%d ranks on %d GPU(s): RCCL needs one device per rank (use one process per GPU, or "
                                   "MSNET_DIST_BACKEND=gloo for a functional run that shares devices)" % (local_world, ndev))
            _ranks_per_device = (local_world + ndev - 1) // ndev
        local = local % ndev
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
    if dist.get_backend() == "gloo" and send.is_cuda:
        # functional multi-rank runs on shared devices: the collective goes through host memory, everything around it stays
        # on the device
        host = torch.empty((world * n_max, H, W), dtype=local_disp.dtype)
        dist.all_gather_into_tensor(host, send.contiguous().cpu())
        recv = host.to(local_disp.device)
    else:
        recv = torch.empty((world * n_max, H, W), dtype=local_disp.dtype, device=local_disp.device)
        dist.all_gather_into_tensor(recv, send.contiguous())      # rank-major concatenation along dim 0
    # recv.view(world, n_max)[r, j] is sample j*world + r
    return recv.view(world, n_max, H, W).permute(1, 0, 2, 3).reshape(world * n_max, H, W)[:n_total].contiguous()


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
