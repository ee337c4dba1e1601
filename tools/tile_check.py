"""Two (or more) ranks render interleaved bands of the same frames with the whole-frame step schedule and the
assembled frames are compared with a single-process render of the whole image.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tools/tile_check.py [--size 96] [--backend gloo]

On a one-GPU box every rank uses cuda:0 and the backend is gloo (RCCL refuses two ranks on one device); on a
multi-GPU node pass --backend nccl and each rank takes cuda:LOCAL_RANK.  Exit code 0 = identical (<= 1/255).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--schedule", default="verify")
    ap.add_argument("--gather-to", default="rank0", choices=["rank0", "all"])
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) if args.backend == "nccl" else 0
    torch.cuda.set_device(dev)
    dist.init_process_group(args.backend)
    from radnerf.parallel import TileParallelRenderer
    from radnerf.scene import SyntheticScene, default_opt
    size = args.size
    scene = SyntheticScene(H=size, W=size, n_frames=8, device=f"cuda:{dev}", opt=default_opt(engine="fused"))
    tpr = TileParallelRenderer(scene, rank, world, dist, band=8, schedule=args.schedule, gather_to=args.gather_to)
    with torch.no_grad():
        for i in range(args.frames):
            tpr.step(i)
        frames = [f.cpu() for f in tpr.finish()]
    print(f"rank {rank}: backend={dist.get_backend()} world={dist.get_world_size()} schedule={tpr.schedule} gather_to={tpr.gather_to}, "
          f"frames rendered again with the whole-frame schedule: {tpr.redone}", flush=True)
    assert (len(frames) == args.frames) if (rank == 0 or args.gather_to == "all") else (frames == [])
    ok = True
    if rank == 0:
        ref = SyntheticScene(H=size, W=size, n_frames=8, device=f"cuda:{dev}", opt=default_opt(engine="fused"))
        with torch.no_grad():
            for i in range(args.frames):
                whole = (ref.render(i)["image"].reshape(size, size, 3) * 255).to(torch.uint8).cpu()
                d = (frames[i].int() - whole.int()).abs()
                print(f"frame {i}: max |d| = {int(d.max())}/255, differing values = {int((d > 0).sum())} of {d.numel()}", flush=True)
                ok = ok and int(d.max()) <= 1 and float((d > 0).float().mean()) < 2e-3
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag) else 1)


if __name__ == "__main__":
    main()
