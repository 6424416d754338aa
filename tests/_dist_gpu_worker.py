"""Worker of tests/test_gpu_more.py::test_two_process_sharded_solve_on_one_gpu: one rank of a 2-rank
landmark-sharded HIP solve; both ranks share cuda:0 and exchange the device-resident reduced system with
gloo (RCCL refuses two ranks on one device), through the same make_torch_allreduce hook bench.py uses."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    from mpsfm_amd import capi
    from mpsfm_amd.dist import make_torch_allreduce, shard_problem
    from mpsfm_amd.synthetic import make_scene

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = sys.argv[1]
    torch.cuda.set_device(0)
    import datetime

    # a collective that does not complete raises after two minutes instead of holding the test for gloo's default half hour
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        probe = torch.ones(4, dtype=torch.float64, device="cuda")
        dist.all_reduce(probe)
        supported = bool(torch.allclose(probe.cpu(), torch.full((4,), float(world), dtype=torch.float64)))
    except Exception as e:  # noqa: BLE001
        supported = False
        print(f"gloo on device tensors unavailable: {e!r}", file=sys.stderr)
    res = {"supported": supported}
    if supported:
        prob, _ = make_scene(12, 6000, True, seed=int(sys.argv[2]))
        if os.environ.get("MPSFM_TEST_FIXED"):  # constant cameras x constant landmarks: fixed blocks, whose cost is a sum over the ranks
            prob.pose_const[2:5] = 1
            prob.pt_const[::4] = 1
        shard, (lo, hi) = shard_problem(prob, rank, world)
        fn, keep = make_torch_allreduce()
        opts = capi.default_options(device=0, verbose=int(os.environ.get("MPSFM_VERBOSE", "0")))  # stream 0: the library creates its own, the hook must follow it
        opts.allreduce = fn
        opts.world_size, opts.rank = world, rank  # exact landmark-gradient maximum through the per-rank slots
        if os.environ.get("MPSFM_TEST_GTOL"):
            opts.gradient_tolerance = float(os.environ["MPSFM_TEST_GTOL"])
        s = capi.ba_solve(shard, opts)
        res.update(lo=lo, hi=hi, final_cost=s["final_cost"], initial_cost=s["initial_cost"], iters=s["num_iterations"], termination=s["termination"],
                   nblocks=s["num_residual_blocks"], trace=list(s["trace_cost"]))
        np.savez(os.path.join(out, f"state{rank}.npz"), pts=shard.pts, cam_quat=shard.cam_quat, cam_t=shard.cam_t)
    with open(os.path.join(out, f"r{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
