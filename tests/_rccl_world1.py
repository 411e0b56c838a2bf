"""Child process of test_config4_through_rccl_at_world_size_1 (not collected by pytest).

One rank, one GPU: a torch.distributed process group of size 1 on the "nccl" backend (= RCCL on ROCm), created BESIDE this
library's HIP runtime, and the BASELINE config-4 batch (8 frames, 100 000 records) taken through DeviceShardedRelocalizer with
both exchanges routed through dist.all_gather_into_tensor on the groups' side streams (force_collective) -- compared with the
unsharded fused tick frame by frame.  Not several ranks on one GPU: RCCL is never asked for that."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
    records = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    order = sys.argv[2] if len(sys.argv) > 2 else "torch_first"
    if order == "library_first":                       # communicator creation after the library has made its context
        from nclt_slam_project_amd.engine import Engine
        e = Engine(0, 640, 480, 2048)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if order != "library_first":
        from nclt_slam_project_amd.engine import Engine
        e = Engine(0, 640, 480, 2048)
    from nclt_slam_project_amd.sharded import DeviceShardedRelocalizer, HipShard
    import bench
    frames, db, base_poses = bench.build_workload(e, records, "fixed64", 8)
    e.db_upload(*db)
    ref = [e.tick(img, bp, global_reloc=True, seed=100 + f) for f, (img, bp) in enumerate(zip(frames, base_poses))]
    # a first collective on the default stream, checked: the communicator works at all
    t = torch.arange(16, dtype=torch.int32, device=dev)
    out = torch.empty((1, 16), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(out, t)
    assert (out[0].cpu().numpy() == np.arange(16)).all()
    shard = HipShard(e, *db, rank=0, world=1)
    sr = DeviceShardedRelocalizer(shard, 0, 1, dev, batch=8, depth=3, force_collective=True)
    assert sr.collective and sr.bases == [0]
    fdev = [e.to_device(f) for f in frames]
    seeds = [100 + f for f in range(8)]

    def check(res, exp):
        for got, x in zip(res, exp):
            assert got["outcome"] == x["outcome"] and got["n_inliers"] == x["n_inliers"] and got["lm_idx"] == x["lm_idx"], (got, x)
            assert got["n_candidates"] == x["n_candidates"]
            np.testing.assert_allclose(got["anchor_pose"], x["anchor_pose"], atol=1e-9)

    for _ in range(2):
        check(sr.tick_batch(fdev, base_poses, seeds), ref)
    flight = [sr.submit(fdev, base_poses, seeds) for _ in range(3)]          # three batches in flight on three side streams
    for b in flight:
        check(sr.result(b), ref)
    short = sr.submit(fdev[2:5], base_poses[2:5], seeds[2:5])
    full = sr.submit(fdev, base_poses, seeds)
    check(sr.result(short), ref[2:5])
    check(sr.result(full), ref)
    n_pub = sum(x["outcome"] == 0 for x in ref)
    assert n_pub >= 6, n_pub
    sr.close()
    shard.close()
    dist.barrier()
    tt = torch.tensor([1.5], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert float(tt.item()) == 1.5
    libs = {l.split()[-1] for l in open("/proc/self/maps").read().splitlines() if "libamdhip64" in l}
    assert len(libs) == 1, libs
    rccl = sorted({os.path.basename(l.split()[-1]) for l in open("/proc/self/maps").read().splitlines() if "librccl" in l})
    e.close()
    dist.destroy_process_group()
    print("rccl-world1 ok: backend", "nccl", "published", n_pub, "of 8; loaded", rccl)


if __name__ == "__main__":
    main()
