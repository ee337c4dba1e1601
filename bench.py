"""Benchmark of the render hot path: frames/s at 512x512 (BASELINE.json config 1) on N MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is one full frame of NeRFRenderer.render: near/far, audio encoding (+ lip EMA), the <=16-step
march / network / composite loop with early termination, the torso pass and the final blend -- on the
synthetic scene of SURVEY §8(d) (ellipsoid head occupancy, OrbitCamera pose stream, random audio features,
random-init weights), all inputs already resident in HBM.  Multi-GPU is frame-parallel: rank r renders
frames r, r+N, ...; finished frames are quantised to uint8 on the device and gathered over RCCL; no other
collective is on the data path (weak scaling: every rank renders K frames).

Rank 0 prints ONE JSON line with the contract fields plus `roofline` (dominant kernel, HIP-event timed on
its launch stream inside the timed region) and `cpu_baseline` (the oracle port of the same frame, timed on
the host cores, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rad-nerf_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=250)     # 20 warm-up + 250 timed frames (10 s of the 25 FPS stream)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--engine", default=os.environ.get("RN_ENGINE", "auto"), choices=["auto", "ops", "fused"])
    ap.add_argument("--grid", default="hash19", choices=sorted(GRIDS),
                    help="xyz grid: hash19 = BASELINE config[1] (hash, T=2^19); tiled16 = the reference's shipped model")
    ap.add_argument("--workload", default="render", choices=["render", "tile", "train"],
                    help="render = BASELINE config[1]/[3] (the headline metric, frame-parallel); tile = config[4], ONE "
                         "1024x1024 frame split over the ranks in interleaved row bands (pass --size 1024); train = "
                         "config[2], one optimisation step of 4096 rays (march_rays_train + network + "
                         "composite_rays_train, backward, Adam)")
    ap.add_argument("--rays", type=int, default=4096, help="rays per training step (--workload train)")
    ap.add_argument("--train-engine", default="graph", choices=["graph", "eager"],
                    help="--workload train: replay the steady-state step from a hipGraph (default) or enqueue it launch by launch")
    ap.add_argument("--mlp", default="f32", choices=["f32", "f32x2", "f16"],
                    help="arithmetic of the fused kernel's contractions: f32 = v_mfma_f32_32x32x2_f32 (headline, fp32 parity); "
                         "f32x2 = fp16 hi+lo operand pairs on the f16 matrix cores (fp32-grade); f16 = fp16 operands, fp32 "
                         "accumulation (the reference's -O/autocast arithmetic)")
    ap.add_argument("--no-loop-hint", action="store_true",
                    help="enqueue all max_steps loop iterations per frame instead of (iterations seen in the warm-up + 2); "
                         "with the hint the device flags any frame it was too small for and the run is repeated without it")
    ap.add_argument("--regime", default="B", choices=["A", "B"],
                    help="SURVEY 8(d): B = worst case (sigma ~ 1, no ray terminates on opacity; default, headline), "
                         "A = opaque (sigma row x80 -> sigma ~ 20..300 inside the head: early termination active)")
    ap.add_argument("--time-every", type=int, default=8,
                    help="time the dominant kernel's launches (HIP events on its dispatches) on every n-th step of the timed region; "
                         "a timed dispatch costs ~10 us of idle GPU, so timing all of them lowers the frame rate by 5 %%")
    ap.add_argument("--audio-batch", type=int, default=8,
                    help="frames whose audio codes / smoothing recurrence / bias blocks are computed together from the resident "
                         "feature stream (4 launches per batch instead of 4 per frame; 0 = per frame, as a live stream would)")
    ap.add_argument("--latency-every", type=int, default=8,
                    help="bracket every n-th timed frame with HIP events on the render stream: mean / p50 / p95 frame latency "
                         "(SURVEY 8(d); the GUI of the reference times a frame the same way, nerf/gui.py:174-202)")
    ap.add_argument("--streams", type=int, default=1,
                    help="frames in flight on this GPU (HIP streams the frames alternate between); 2 lets one frame's small kernels run "
                         "beside the other frame's network kernel (throughput mode; needs --audio-batch > 0)")
    ap.add_argument("--loop-launch", default="split", choices=["coop", "split"],
                    help="fused engine: compositor + compaction + next march of a loop iteration in one launch with a grid-wide "
                         "barrier inside (2 launches per iteration) / a launch each for compositor and compaction (3)")
    ap.add_argument("--half-tables", action="store_true",
                    help="the kernels read persistent fp16 copies of the grid tables (opt.half_tables; what load_checkpoint(half_tables=True) "
                         "sets up) -- with --mlp f16 the reference's -O mode without its per-call table casts (gridencoder/grid.py:43-44)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-record", action="store_true",
                    help="skip the bounded config[2] run (64 graph-replayed training steps) that the default single-GPU render line "
                         "carries under the key `train`")
    ap.add_argument("--cpu-baseline-size", type=int, default=0, help="0 = same size as the GPU workload")
    return ap.parse_args()


GRIDS = {"hash19": dict(xyz_grid="hashgrid", xyz_log2_hashmap_size=19),
         "tiled16": dict(xyz_grid="tiledgrid", xyz_log2_hashmap_size=16)}
GRID_TEXT = {"hash19": "L=16 hash grid T=2^19 F=2 (xyz; ambient/torso grids tiled T=2^16 as shipped)",
             "tiled16": "L=16 tiled grids T=2^16 F=2 (the reference's shipped model)"}


def pick_engine(name):
    if name != "auto":
        return name
    return "fused" if os.path.exists(os.path.join(ROOT, "rad-nerf_amd", "radnerf", "fused.py")) else "ops"


# Algorithmic bytes per sample: xyz grid 16 levels x 8 corners x 8 B = 1024 B gathered
# + 12 B of coordinates in + 128 B of features out.
GRID_XYZ_BYTES_PER_SAMPLE = 1024 + 12 + 128
# Fused per-sample kernel (fp32 tables): 1024 B (xyz grid) + 512 B (ambient grid: 16 levels x 4 corners x 8 B)
# gathered, + 12 B xyz + 12 B dir + 4 B delta in, + 4 B sigma + 12 B rgb out; encodings / activations stay on chip.
FUSED_BYTES_PER_SAMPLE = 1024 + 512 + 12 + 12 + 4 + 4 + 12
MLP_FLOP_PER_SAMPLE = 56704          # 2 x 28352 MAC
MFMA_F32_PEAK_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
MFMA_F16_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense f16/bf16 peak (the K=16 forms; the kernels use K=8)


def kernel_select(engine, acc):
    """Which C-ABI call is the dominant kernel; accumulates its algorithmic bytes per launch into `acc`."""
    if engine == "fused":
        def sel(name, args):
            return "nerf_fused" if name == "rn_nerf_forward_fused" else None
        return sel

    def sel(name, args):
        # xyz grid lookup: D == 3 (args: inputs, table, offsets, outputs, B, D, C, L, ...)
        if name == "rn_grid_encode_forward" and args[5] == 3:
            acc["grid_encode_xyz"] = acc.get("grid_encode_xyz", 0.0) + float(args[4]) * GRID_XYZ_BYTES_PER_SAMPLE
            return "grid_encode_xyz"
        return None
    return sel


def cpu_baseline(scene_kwargs, size, opt_overrides):
    """The oracle's port of the same frame (CPU, OpenMP over the host cores): bounded sample = 1 frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from radnerf.scene import SyntheticScene, default_opt
    scene = SyntheticScene(H=size, W=size, n_frames=8, device="cpu", opt=default_opt(**opt_overrides), **scene_kwargs)
    m = scene.model
    f = scene.frame(0)
    with torch.no_grad():
        enc_a = m.encode_audio(f["auds"])
    om = po.model_from_module(m)
    rc = po.render_cfg_from_module(m, scene.opt.dt_gamma, scene.opt.max_steps)
    args = (om, rc, f["rays_o"].numpy(), f["rays_d"].numpy(), enc_a.numpy(), m.individual_codes[0].detach().numpy(),
            f["eye"].numpy(), f["bg_coords"].numpy(), f["poses"].numpy(), m.individual_codes_torso[0].detach().numpy(),
            f["bg_color"].reshape(-1, 3).numpy())
    po.render_frame(*args)  # warm-up (page-in, thread pool)
    reps, t0 = 0, time.perf_counter()
    while True:
        _, _, stats = po.render_frame(*args)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 6.0 or reps >= 12:
            break
    return dict(value=reps / dt, unit="frames/s", cores=po.num_threads(), kind="port",
                sample=f"{reps} x frame 0 of the same {size}x{size} workload through oracle/ (orc_render_frame, "
                       f"fp32, OpenMP), {stats['live_samples']} samples/frame",
                samples_per_s=stats["live_samples"] * reps / dt)


def cpu_reference_flow(size, opt_overrides, budget_s, label):
    """SURVEY 8(d)'s CPU baseline: the reference's control flow (this tree's mirror of nerf/network.py + nerf/renderer.py,
    "ops" loop shape -- the reference's own files do not travel to this box) over CPU restatements of the kernels (oracle,
    OpenMP) and PyTorch CPU GEMMs for the MLPs, fp32, autocast off, all host cores (oracle/cpu_ops.py)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_ops
    import pyoracle as po
    from radnerf.scene import SyntheticScene, default_opt
    cores = po.num_threads()
    torch.set_num_threads(cores)
    scene = SyntheticScene(H=size, W=size, n_frames=8, device="cpu", opt=default_opt(**opt_overrides))
    with cpu_ops.cpu_operators(scene.model), torch.no_grad():
        scene.render(0)                         # warm-up frame
        reps, t0 = 0, time.perf_counter()
        while True:
            scene.render(1 + reps)
            reps += 1
            dt = time.perf_counter() - t0
            if dt > budget_s or reps >= 10:
                break
    return dict(value=reps / dt, unit="frames/s", cores=cores, kind="reference-flow", workload=label,
                sample=f"{reps} frame(s) of {size}x{size} after 1 warm-up: mirror of the reference's Python loop over oracle CPU operators "
                       f"+ torch CPU GEMMs ({torch.get_num_threads()} threads)")


def run_train(args):
    """BASELINE config[2] on one GPU: steps/s and samples/s of the training step (radnerf/train.py) on 4096 random rays of
    frame 0, head model (torso off, as the reference trains the head), perturb on, occupancy grid refreshed every 16
    steps inside the timed region.  --train-engine graph (default): the steady-state step replayed from a hipGraph
    (GraphedTrainer); eager: one Python-enqueued launch sequence per step.  A secondary line: the headline metric stays the
    render workload (whose line carries a bounded run of this one under the key `train`)."""
    print(json.dumps(train_record(args)))


def train_record(args, device_index=0):
    torch.cuda.set_device(device_index)
    import radnerf_hip as hip
    from radnerf.scene import SyntheticScene, default_opt
    from radnerf.train import GraphedTrainer, SyntheticTrainStream, Trainer
    size, K, W = args.size, args.steps, args.warmup
    scene = SyntheticScene(H=size, W=size, n_frames=8, device="cuda", opt=default_opt(engine="ops", torso=False, smooth_lips=False, **GRIDS[args.grid]))
    stream = SyntheticTrainStream(scene, n_rays=args.rays)
    graphed = args.train_engine == "graph"
    trainer = (GraphedTrainer if graphed else Trainer)(scene.model, scene.opt)
    m = scene.model
    W = max(W, 33)                       # past the first two grid refreshes, so mean_count is warm
    for _ in range(W):
        trainer.step(stream.batch())
    samples = torch.zeros((), dtype=torch.int64, device="cuda")
    segs = 5
    seg_ms, losses = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_seg = t0
    for i in range(K):
        losses.append(trainer.step(stream.batch()))
        samples += m.step_counter[(m.local_step - 1) % 16, 0]        # samples marched by this step (stays on the device)
        if (i + 1) % max(K // segs, 1) == 0 and len(seg_ms) < segs:
            torch.cuda.synchronize()
            now = time.perf_counter()
            seg_ms.append((now - t_seg) * 1e3 / max(K // segs, 1))
            t_seg = now
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    samples = int(samples.item())
    # The dominant kernel of the step (the xyz table's gradient scatter), timed with HIP events around its own C-ABI call on the
    # stream it is launched on.  A replayed hipGraph exposes no per-kernel events, so the SAME step is enqueued eagerly a few times
    # right after the timed region (same model state, same batches' distribution, same kernels and launch parameters as the
    # captured ones); profiles/r03_rocprofv3_kernel_stats_train_graph.csv holds the replayed kernels' durations for comparison.
    acc = {}

    def sel(name, a):
        if name == "rn_grid_scatter_jobs":       # (jobs, n_jobs, M, m_dev, workspace, bytes, stream): both grids' table gradients
            acc["calls"] = acc.get("calls", 0) + 1
            return "grid_scatter_xyz"
        if name == "rn_grid_encode_backward" and a[6] == 3:     # per-operator path (RN_TRAIN_HEAD=ops)
            acc["calls"] = acc.get("calls", 0) + 1
            return "grid_scatter_xyz"
        return None
    eager = Trainer.__new__(Trainer)             # the same model and optimizer state, stepped launch by launch
    eager.model, eager.opt, eager.optimizer, eager.update_extra_interval = m, scene.opt, trainer.optimizer, 0
    eager.iters, eager.lambda_amb, eager.global_step = trainer.iters, trainer.lambda_amb, trainer.global_step
    timer = hip.KernelTimer(sel)
    probe = torch.zeros((), dtype=torch.int64, device="cuda")
    n_probe = 8
    for i in range(n_probe + 2):
        if i == 2:
            hip.set_timer(timer)
        eager.step(stream.batch())
        if i >= 2:
            probe += m.step_counter[(m.local_step - 1) % 16, 0]
    hip.set_timer(None)
    probe_samples = int(probe.item()) / n_probe
    # cost of one occupancy refresh (update_extra_state: 2.1 M-point density query + dilation + decayed max + mean + packbits)
    evs = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        with torch.no_grad():
            m.update_extra_state()
        b.record()
        evs.append((a, b))
    torch.cuda.synchronize()
    refresh_ms = sorted(a.elapsed_time(b) for a, b in evs)[1]
    res = timer.results().get("grid_scatter_xyz")
    roof = None
    if res:
        # scatter-add of 16 levels x 8 corners x 8 B (read-modify-write counted once) + 128 B grad + 12 B coords for the xyz grid,
        # 16 x 4 x 8 B + 128 B + 8 B for the ambient grid: the call covers both (3 launches)
        per_launch = probe_samples * ((1024 + 128 + 12) + (512 + 128 + 8))
        ach = per_launch / (res["avg_ms"] * 1e-3) / 1e9
        roof = dict(bound="hbm", kernel="k_grid_scatter<3,2> (table-gradient scatter-add of the xyz and ambient grids, one launch)", achieved=ach, peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=ach / HBM_PEAK_GBS, traffic=None, launches=res["launches"], avg_launch_ms=res["avg_ms"],
                    algorithmic_bytes_per_launch=per_launch, samples_per_launch=probe_samples,
                    share_of_step=res["avg_ms"] * ((samples / K) / max(probe_samples, 1.0)) / (elapsed / K * 1e3),
                    source="HIP events around the kernel's C-ABI call in %d eagerly enqueued steps right after the timed region "
                           "(a replayed hipGraph has no per-kernel events); share_of_step scales the duration to the timed region's "
                           "samples per step" % n_probe,
                    note="a workgroup merges its samples' rows per 64-B line of the gradient table in LDS, then issues one float-atomic "
                         "request per touched line (memory-side atomics: ~20 G requests/s chip-wide; LDS float atomics: 3.6 clocks per "
                         "lane): bound by those two rates, not by bytes.  The call runs on a side stream beside the weight-gradient and "
                         "audio-backward kernels, so its duration here includes that contention")
    sps = [1e3 / t for t in seg_ms]
    return ({
        "metric": "training steps/sec @4096 rays", "value": K / elapsed, "unit": "steps/s", "n_gpus": 1, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"config[2]: single-GPU training step, {args.rays} rays of a {size}x{size} frame, {GRID_TEXT[args.grid]}, "
                               "march_rays_train + network + composite_rays_train fwd/bwd + grid_encode backward + Adam, "
                               "update_extra_state every 16 steps (inside the timed region)", "grid": args.grid,
                   "train_engine": args.train_engine},
        "samples_per_s": samples / elapsed, "samples_per_step": samples / K,
        "steps_per_s_by_segment": sps, "spread": (max(sps) - min(sps)) / (sum(sps) / len(sps)) if sps else None,
        "occupancy_refresh_ms": refresh_ms, "occupancy_refresh_share_of_step": refresh_ms / 16 / (elapsed / K * 1e3),
        "graph_captures": getattr(trainer, "captures", None), "graph_replays": getattr(trainer, "replays", None),
        "graph_capture_log": getattr(trainer, "capture_log", None),
        "loss_first": float(losses[0]), "loss_last": float(losses[-1]), "roofline": roof})


def ctypes_field(arg, name):
    """Field of a ctypes struct passed by reference (byref / pointer) to a C-ABI call, for the kernel selectors."""
    obj = getattr(arg, "_obj", None)
    if obj is None:
        obj = arg.contents
    return getattr(obj, name)


def main():
    args = parse()
    if args.workload == "train":
        assert args.gpus == 1, "training is replicas-only (SURVEY 8(e)); run --workload train with --gpus 1"
        return run_train(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RN_DIST_BACKEND", "nccl")      # "gloo": rehearse N ranks on ONE GPU (RCCL needs a GPU per rank)
        if backend == "gloo":
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    device = torch.device("cuda", local_rank if world > 1 else 0)

    import radnerf_hip as hip
    from radnerf.scene import SyntheticScene, default_opt
    from radnerf.parallel import FrameParallelRenderer, TileParallelRenderer

    engine = pick_engine(args.engine)
    size = args.size
    K, W = args.steps, args.warmup
    n_frames = 250
    scene = SyntheticScene(H=size, W=size, n_frames=n_frames, device=device,
                           opt=default_opt(engine=engine, mlp_dtype=args.mlp, loop_launch=args.loop_launch, half_tables=args.half_tables,
                                           **GRIDS[args.grid]))
    if args.regime == "A":
        with torch.no_grad():
            scene.model.sigma_net.net[-1].weight[0].abs_().mul_(80.0)
    tile = args.workload == "tile"
    fpr = (TileParallelRenderer(scene, rank, world, dist, speculate_loop=not args.no_loop_hint) if tile else
           FrameParallelRenderer(scene, rank, world, dist, speculate_loop=not args.no_loop_hint, audio_batch=args.audio_batch,
                                 streams=args.streams))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for s in range(W):
            fpr.step(s)
        fpr.finish()
        acc = {}
        timer = hip.KernelTimer(kernel_select(engine, acc))
        hip.set_timer(timer)
        # a short run still times >= 8 steps (kernel events) and >= 8 frames (latency): the driver's --steps 20 gives 10 + 10
        args.time_every = max(1, min(args.time_every, K // 8))
        args.latency_every = max(1, min(args.latency_every, K // 8))
        if engine == "fused":
            hip.prof_enable(os.environ.get("RN_BENCH_PROF", "1") != "0")   # 0: tools/trace_gaps.py checks the timing costs nothing
            c0 = fpr.loop_counters() or [0, 0, 0]
        from radnerf.parallel import LoopHintTooSmall
        for attempt in range(2):
            barrier()
            t0 = time.perf_counter()
            short = 0
            try:
                lat_events = []
                le = max(args.latency_every, 1)
                for s in range(W, W + K):
                    if engine == "fused":
                        hip.prof_pause((s - W) % max(args.time_every, 1) != 0)
                    if (s - W) % le == 1 % le:      # never a frame whose kernel dispatches carry timing events, nor one that opens an audio batch
                        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        rs = fpr.render_stream(s) if hasattr(fpr, "render_stream") else torch.cuda.current_stream()
                        ea.record(rs)
                        fpr.step(s)
                        eb.record(rs)
                        lat_events.append((ea, eb))
                    else:
                        fpr.step(s)
                fpr.finish()
            except LoopHintTooSmall:        # the device flagged a frame: this timing is void, repeat with every iteration
                short = 1
            barrier()
            elapsed = time.perf_counter() - t0
            flag = torch.tensor([short], dtype=torch.int32, device=device if dist is None or dist.get_backend() == "nccl" else "cpu")
            if dist is not None:
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) == 0:
                break
            fpr.speculate_loop = False
            from radnerf import fused as _fused
            _fused.set_loop_hint(scene.model, None)
            if engine == "fused":
                hip.prof_collect(); hip.prof_durations()      # drop the void attempt's kernel timings
                c0 = fpr.loop_counters() or [0, 0, 0]
        hip.set_timer(None)
        if engine == "fused":
            fused_launches, fused_ms = hip.prof_collect()
            fused_durs = sorted(hip.prof_durations(), reverse=True)
            hip.prof_enable(False)
            c1 = fpr.loop_counters()                        # device-side loop statistics (cumulative counters)
            live_total = (c1[1] - c0[1]) & 0xFFFFFFFF
            iters_total = (c1[0] - c0[0]) & 0xFFFFFFFF      # launches that had samples to process
            live_pf, slots_pf = live_total / K, ((c1[2] - c0[2]) & 0xFFFFFFFF) / K
        else:
            # untimed replay of a few of the timed frames to count live samples per frame
            live_pf, slots_pf = fpr.count_samples(list(range(W, W + min(K, 8))))

    el = torch.tensor([elapsed], dtype=torch.float64, device=device if dist is None or dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())
    total_frames = K if tile else K * world       # tile-parallel: every rank works on the same K frames
    fps = total_frames / elapsed

    if rank == 0:
        res = timer.results()
        roof = None
        lat = sorted(a.elapsed_time(b) for a, b in lat_events)
        latency = None
        if lat:
            latency = dict(mean=sum(lat) / len(lat), p50=lat[len(lat) // 2], p95=lat[min(len(lat) - 1, int(0.95 * len(lat)))],
                           frames_sampled=len(lat), method="HIP events on the render stream around every "
                           f"{max(args.latency_every, 1)}-th timed frame (frames are enqueued back to back, so this is the GPU time of a frame)")
        if engine == "fused" and fused_launches:
            # the kernel was timed on every --time-every-th step; the stream's frames hold the same number of samples to within
            # a fraction of a percent, so the timed steps' share of the counted samples is their share of the steps
            timed_steps = len(range(0, K, max(args.time_every, 1)))
            timed_frac = timed_steps / K
            live_timed = live_total * timed_frac
            iters_timed = int(round(iters_total * timed_frac))
            nbytes = live_timed * FUSED_BYTES_PER_SAMPLE
            achieved = nbytes / (fused_ms * 1e-3) / 1e9
            tflops = live_timed * MLP_FLOP_PER_SAMPLE / (fused_ms * 1e-3) / 1e12
            # The kernel does both of the path's heavy jobs (grid gathers and the MLP contraction).  With fp32 MFMA the
            # matrix-core roof is the closer (binding) one; on the 16-bit matrix cores the contraction is ~16x cheaper and
            # the gathers bind.  `bound` names the binding roof, the other view rides along.
            tkey = {"f32": "nerf_fused", "f16": "nerf_fused_h16", "f32x2": "nerf_fused_x2"}[args.mlp]
            # bytes per launch WITH work from the counter passes, spread over this run's launches (some of which are the
            # zero-sample launches behind the end of the loop)
            # (the passes were taken on config[1] itself -- 512x512, hash19, regime B -- and describe only that workload)
            profiled = args.grid == "hash19" and size == 512 and not tile and args.regime == "B"
            t_work = fpr.measured_traffic(tkey, "hbm_bytes_per_launch_with_work") if profiled else None
            traffic = t_work * iters_timed / fused_launches if t_work else (fpr.measured_traffic(tkey) if profiled else None)
            common = dict(traffic=traffic,
                          traffic_source="profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the committed profile "
                                         "run of this command (static in this line, not measured in this run): bytes per launch with "
                                         "work x this run's share of launches with work",
                          launches=fused_launches, avg_launch_ms=fused_ms / fused_launches,
                          share_of_step=fused_ms / (elapsed * 1e3 * timed_frac), launches_with_work=iters_timed,
                          avg_launch_ms_with_work=sum(fused_durs[:iters_timed]) / max(iters_timed, 1),
                          timed_steps=timed_steps,
                          note="HIP events on the kernel's dispatches, on every --time-every-th step of the timed region "
                               "(a timed dispatch costs ~10 us of idle GPU; timing all of them lowers the frame rate by 5 %); "
                               "launches/avg_launch_ms count every timed launch of the kernel (as rocprof "
                               "does); iterations past the end of the loop launch with zero samples and exit at once; "
                               "traffic = HBM bytes per launch from the FETCH_SIZE/WRITE_SIZE passes in profiles/")
            mfma_peak = MFMA_F32_PEAK_TFLOPS if args.mlp == "f32" else MFMA_F16_PEAK_TFLOPS
            mfma_view = dict(achieved=tflops, peak=mfma_peak, unit="TFLOP/s", frac=tflops / mfma_peak,
                             algorithmic_flop_per_launch=live_timed * MLP_FLOP_PER_SAMPLE / fused_launches,
                             algorithmic_flop_per_sample=MLP_FLOP_PER_SAMPLE)
            hbm_view = dict(achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                            algorithmic_bytes_per_launch=nbytes / fused_launches,
                            algorithmic_bytes_per_sample=FUSED_BYTES_PER_SAMPLE)
            if args.mlp == "f32":
                roof = dict(bound="mfma", kernel="k_nerf_fused (grid gathers + fp32 MFMA MLPs)", **mfma_view, **common, hbm=hbm_view)
            elif args.mlp == "f32x2":
                # 3 f16 MFMA products per fp32-grade product: the matrix-core view counts the issued FLOPs (3x)
                mfma_view.update(achieved=3 * tflops, frac=3 * tflops / mfma_peak, issued_over_algorithmic=3)
                roof = dict(bound="hbm", kernel="k_nerf_fused_x2 (grid gathers + split-precision f16 MFMA MLPs, fp32-grade)",
                            **hbm_view, **common, mfma=mfma_view)
            else:
                roof = dict(bound="hbm", kernel="k_nerf_fused_h16 (grid gathers + f16 MFMA MLPs, fp32 accumulate)", **hbm_view,
                            **common, mfma=mfma_view)
        for key, r in res.items():
            per_launch_bytes = acc.get(key, 0.0) / max(r["launches"], 1)
            achieved = per_launch_bytes / (r["avg_ms"] * 1e-3) / 1e9 if r["avg_ms"] > 0 else 0.0
            roof = dict(bound="hbm", kernel=key, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=achieved / HBM_PEAK_GBS, traffic=fpr.measured_traffic(key), launches=r["launches"],
                        avg_launch_ms=r["avg_ms"], algorithmic_bytes_per_launch=per_launch_bytes,
                        share_of_step=r["total_ms"] / (elapsed * 1e3))
        out = {
            "metric": f"rendered frames/sec @{size}x{size}", "value": fps, "unit": "frames/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong" if tile else "weak",
            "vs_baseline": None, "dtype": {"f32": "f32", "f16": "f16 operands / f32 accumulate",
                                          "f32x2": "f32 (operands as fp16 hi+lo pairs on MFMA, f32 accumulate)"}[args.mlp],
            "data": "synthetic",
            "config": {"workload": (f"config[4]: tile-parallel single {size}x{size} frame, interleaved 8-row bands, " if tile else
                                    f"config[{1 if world == 1 else 3}]: inference {size}x{size}, ") + f"{GRID_TEXT[args.grid]}, "
                                   "max 16 steps/ray, 25 FPS pose stream, torso pass on"
                                   + (", regime A (opaque: sigma ~ 20..300 inside the head, rays terminate on T < 1e-4)" if args.regime == "A" else ""),
                       "grid": args.grid, "engine": engine, "frames_per_gpu": K, "table_dtype": "f16 (persistent copies)" if args.half_tables else "f32",
                       "scene": {"occupancy_ellipsoid_semi_axes": [0.40, 0.42, 0.40], "camera": "OrbitCamera radius 3.35, fovy 21.24 deg, "
                                 "yaw 8 deg sin(2 pi t / 4 s), pitch 4 deg sin(2 pi t / 2.5 s)", "regime": args.regime,
                                 "note": "SURVEY 8(d) proposes semi-axes (0.33, 0.42, 0.33); (0.40, 0.42, 0.40) reproduces the published "
                                         "trace's 31 % of rays hitting the head at this pose"},
                       "audio_batch": args.audio_batch if engine == "fused" and not tile else 0,
                       "frames_in_flight": getattr(fpr, "n_streams", 1), "loop_launch": getattr(scene.opt, "loop_launch", "split"),
                       "collectives": (dict({"per_frame": 1, "kind": "gather of uint8 band rows to rank 0 (+ 68 B of loop counts that verify the "
                                             "band-local step schedules; the verdicts return in one broadcast per finish())",
                                             "schedule": getattr(fpr, "schedule", None), "frames_redone_exactly": getattr(fpr, "redone", 0)}
                                            if tile else {"per_frame": 1.0 / max(getattr(fpr, "gather_every", 1), 1), "kind": "gather of uint8 frames to rank 0, "
                                                          f"{getattr(fpr, 'gather_every', 1)} frames per collective"},
                                            backend=dist.get_backend(), world_seen_by_backend=dist.get_world_size())) if world > 1 else None,
                       "loop_iterations_enqueued": (getattr(scene.model, "_fused_loop_hint", None) or scene.opt.max_steps),
                       "parallelism": f"{'tile' if tile else 'frame'}-parallel x{world}"},
            # tile-parallel: rank 0 counts its own band's samples; the bands are interleaved, so x world is the frame's
            "samples_per_s": live_pf * (world if tile else 1) * fps, "samples_per_frame": live_pf * (world if tile else 1),
            "sample_slots_per_frame": slots_pf * (world if tile else 1),
            "frame_latency_ms": latency,
            "roofline": roof,
        }
        if world == 1 and not tile and not args.no_train_record and engine == "fused" and args.regime == "B":
            # BASELINE config[2] in front of the driver: a bounded run of the training step (64 replayed steps after the warm-up,
            # ~2 s), same code as `bench.py --workload train`
            import argparse
            targs = argparse.Namespace(**vars(args))
            targs.steps, targs.warmup, targs.train_engine, targs.rays, targs.workload = 64, 33, "graph", 4096, "train"
            del scene, fpr
            torch.cuda.empty_cache()
            rec = train_record(targs, device.index or 0)
            out["train"] = {"metric": rec["metric"], "steps_per_s": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"],
                            "steps": rec["steps"], "warmup": rec["warmup"], "dtype": rec["dtype"], "config": rec["config"],
                            "samples_per_s": rec["samples_per_s"], "samples_per_step": rec["samples_per_step"],
                            "steps_per_s_by_segment": rec["steps_per_s_by_segment"], "spread": rec["spread"],
                            "occupancy_refresh_ms": rec["occupancy_refresh_ms"], "graph_captures": rec["graph_captures"],
                            "roofline": rec["roofline"]}
        if world == 1 and not args.no_cpu_baseline and args.regime == "B":
            cb_size = args.cpu_baseline_size or size
            out["cpu_baseline"] = cpu_baseline({}, cb_size, GRIDS[args.grid])
            if not tile:
                # the reference-flow baseline on config[0] (one 256 x 256 frame, shipped tiled grids) and config[1]
                out["cpu_baseline_reference_flow"] = [
                    cpu_reference_flow(256, GRIDS["tiled16"], 4.0, "config[0]: 256x256, shipped tiled T=2^16 grids"),
                    cpu_reference_flow(cb_size, GRIDS[args.grid], 10.0, f"config[1]: {cb_size}x{cb_size}, {GRID_TEXT[args.grid]}")]
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
