"""Landmark-sharded bundle adjustment across ranks.

Every rank holds all camera poses and a contiguous range of landmarks with all their
observations.  Per LM iteration the ranks exchange one sum-all-reduce of the packed reduced
camera system [S blocks | g_c | W V^-1 g_p | diag U | scalars] and one of five step scalars; the
dense solve is replicated.  The exchange is injected into the solver through the
``mpsfm_allreduce_fn`` hook of include/mpsfm_hip.h, so the C library itself stays free of torch.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .problem import ALLREDUCE_FN, BAProblem


def landmark_ranges(prob: BAProblem, world: int) -> list[tuple[int, int]]:
    """Contiguous landmark ranges balanced by residual-block count."""
    w = np.bincount(prob.obs_pt, minlength=prob.n_pts).astype(np.int64)
    if prob.n_dobs:
        w += np.bincount(prob.dobs_pt, minlength=prob.n_pts)
    csum = np.concatenate([[0], np.cumsum(w)])
    total = csum[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(csum, total * r / world, side="left")))
    cuts.append(prob.n_pts)
    cuts = np.maximum.accumulate(np.clip(cuts, 0, prob.n_pts))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def shard_problem(prob: BAProblem, rank: int, world: int) -> tuple[BAProblem, tuple[int, int]]:
    """The shard of `prob` owned by `rank`: landmarks [lo, hi) re-indexed from 0, all cameras."""
    lo, hi = landmark_ranges(prob, world)[rank]
    mo = (prob.obs_pt >= lo) & (prob.obs_pt < hi)
    md = (prob.dobs_pt >= lo) & (prob.dobs_pt < hi)
    shard = BAProblem(
        cam_quat=prob.cam_quat.copy(), cam_t=prob.cam_t.copy(), pts=prob.pts[lo:hi].copy(),
        cam_intr=prob.cam_intr, cam_intr_idx=prob.cam_intr_idx, pose_const=prob.pose_const,
        pt_const=prob.pt_const[lo:hi], obs_cam=prob.obs_cam[mo], obs_pt=prob.obs_pt[mo] - lo,
        obs_xy=prob.obs_xy[mo], gauge_axis_cam=prob.gauge_axis_cam,
        reproj_loss_type=prob.reproj_loss_type, reproj_loss_scale=prob.reproj_loss_scale,
        reproj_loss_magnitude=prob.reproj_loss_magnitude,
        dobs_cam=prob.dobs_cam[md], dobs_pt=prob.dobs_pt[md] - lo, dobs_depth=prob.dobs_depth[md],
        dobs_magnitude=prob.dobs_magnitude[md], dobs_param=prob.dobs_param[md],
        depth_loss_type=prob.depth_loss_type, shift_logscale=prob.shift_logscale,
    )
    return shard, (lo, hi)


class _DevView:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {
            "shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2, "strides": None,
        }


def make_torch_allreduce(group=None):
    """Returns (ALLREDUCE_FN, keepalive).  Sums host buffers (gloo or nccl via staging) and device
    buffers (RCCL, in place) over the ranks of `group` with torch.distributed."""
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    paranoid = os.environ.get("MPSFM_DIST_SYNC") == "1"  # debugging aid: device-wide sync around every collective

    def _cb(user, buf, count, on_device, stream):
        try:
            count = int(count)
            if count <= 0:
                return 0
            ptr = C.addressof(buf.contents)
            if on_device:
                # The buffer is produced and consumed on the solver's HIP stream, which is NOT torch's
                # current stream in general (the library creates its own when options.stream is 0).  Make
                # it current for the collective: RCCL's stream then waits for the kernels queued before
                # the call, and work.wait() makes the solver's stream wait for the collective.
                s = torch.cuda.ExternalStream(int(stream)) if stream else torch.cuda.current_stream()
                if paranoid:
                    torch.cuda.synchronize()
                with torch.cuda.stream(s):
                    t = torch.as_tensor(_DevView(ptr, count), device="cuda")
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                if paranoid:
                    torch.cuda.synchronize()
            else:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                t = torch.from_numpy(a)
                if backend == "nccl":
                    g = t.cuda()
                    dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
                    t.copy_(g.cpu())
                else:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return 0
        except Exception as e:  # noqa: BLE001 - never unwind through C
            import sys

            print(f"[mpsfm_amd.dist] all-reduce failed: {e!r}", file=sys.stderr)
            return -1

    fn = ALLREDUCE_FN(_cb)
    return fn, (_cb, fn)


def hook_options(opts, fn, group=None):
    """`opts` for a sharded solve through the hook: the hook itself plus this rank's position (per-rank slots of the
    summed scalar tail keep the landmark-gradient maximum exact)."""
    import torch.distributed as dist

    opts.allreduce = fn
    opts.world_size, opts.rank = dist.get_world_size(group), dist.get_rank(group)
    return opts


def use_native_rccl(opts, group=None):
    """Fills `opts` (capi.COptions) so that the library creates its OWN RCCL communicator over the ranks of `group`
    (ncclCommInitRank at mpsfm_ba_create): rank 0 draws the unique id, torch.distributed only carries its 128 bytes to
    the other ranks.  Afterwards no Python runs inside the LM loop: every exchange is an ncclAllReduce on the solver's
    stream."""
    import torch
    import torch.distributed as dist

    from . import capi

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        t.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
    dist.broadcast(t, src=0, group=group)
    raw = bytes(t.cpu().numpy().tobytes())
    for i in range(128):
        opts.comm_id[i] = raw[i]
    opts.world_size, opts.rank, opts.use_rccl = world, rank, 1
    opts.allreduce = ALLREDUCE_FN()  # NULL: the hook is not used
    return opts
