"""Two (or more) ranks render a pose / audio stream frame-parallel (rank r: frames r, r + W, ...) and the frames gathered on
rank 0 are compared with a single-process render of the same stream (the lip-smoothing EMA makes the stream sequential:
nerf/renderer.py:190-194).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        tools/frame_check.py [--size 64] [--steps 5] [--backend nccl] [--streams 2]

One GPU: every rank uses cuda:0 and the backend is gloo; several GPUs: --backend nccl, each rank on cuda:LOCAL_RANK.
Exit code 0 = every gathered frame equals the sequential one (<= 1/255 on a handful of values).
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
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--gather-every", type=int, default=2)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) if args.backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(args.backend)
    from radnerf.parallel import FrameParallelRenderer
    from radnerf.scene import SyntheticScene, default_opt
    size, n_frames = args.size, 32
    scene = SyntheticScene(H=size, W=size, n_frames=n_frames, device=f"cuda:{dev}", opt=default_opt(engine="fused"))
    fpr = FrameParallelRenderer(scene, rank, world, dist, gather_every=args.gather_every, audio_batch=4, streams=args.streams)
    with torch.no_grad():
        for s in range(args.steps):
            fpr.step(s)
        stacks = [t.cpu() for t in fpr.finish()]
    print(f"rank {rank}: backend={dist.get_backend()} world={dist.get_world_size()} streams={fpr.n_streams} stacks={len(stacks)}", flush=True)
    ok = True
    if rank == 0:
        assert len(stacks) == args.steps
        ref = SyntheticScene(H=size, W=size, n_frames=n_frames, device=f"cuda:{dev}", opt=default_opt(engine="fused"))
        with torch.no_grad():
            for g in range(args.steps * world):
                whole = ref.render(g, want_u8=True)["image_u8"].reshape(size, size, 3).cpu()
                d = (stacks[g // world][g % world].int() - whole.int()).abs()
                if int(d.max()) > 1 or float((d > 0).float().mean()) > 2e-3:
                    ok = False
                    print(f"frame {g}: max |d| = {int(d.max())}/255, differing values = {int((d > 0).sum())} of {d.numel()}", flush=True)
        print(f"frames checked: {args.steps * world}, identical: {ok}", flush=True)
    else:
        assert stacks == []
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=f"cuda:{dev}" if args.backend == "nccl" else "cpu")
    dist.broadcast(flag, 0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag) else 1)


if __name__ == "__main__":
    main()
