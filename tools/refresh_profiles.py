"""Copy the evidence of tools/gpu_final_round.sh / tools/gpu_round2.sh (gpurun_out/final/, gpurun_out/final2/) into
profiles/ under this round's prefix.

    python tools/refresh_profiles.py r01 [final]      python tools/refresh_profiles.py r02 final2
Bench JSON lines, isolated-kernel timings, the rocprofv3 --kernel-trace --stats summary of the bench command per
arithmetic variant, the FETCH_SIZE / WRITE_SIZE counter rows of this library's kernels, and profiles/traffic.json."""
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(ROOT, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "final")
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    fs = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return fs[-1] if fs else None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    for f in glob.glob(os.path.join(F, "bench_*.json")):
        shutil.copy(f, os.path.join(P, f"{tag}_{os.path.basename(f)}"))
    for f in glob.glob(os.path.join(F, "rehearse4_*.json")):
        shutil.copy(f, os.path.join(P, f"{tag}_{os.path.basename(f)}"))
    for src, dst in (("kernels.json", "kernel_microbench.json"), ("fused_kernel.jsonl", "fused_kernel_isolated.jsonl"),
                     ("pytest_gpu.log", "pytest_gpu.log"), ("mlp_kernels.json", "mlp_train_kernels.json"),
                     ("train_probe.json", "train_step_probe.json")):
        if os.path.exists(os.path.join(F, src)):
            shutil.copy(os.path.join(F, src), os.path.join(P, f"{tag}_{dst}"))
    st = newest(os.path.join(F, "trace_train", "**", "*kernel_stats.csv"))
    if st:
        shutil.copy(st, os.path.join(P, f"{tag}_rocprofv3_kernel_stats_train.csv"))
    L = os.path.join(ROOT, "gpurun_out", "lookup")           # tools/gpu_lookup_profile.sh
    if os.path.isdir(L):
        os.makedirs(os.path.join(P, f"{tag}_lookup"), exist_ok=True)
        for f in glob.glob(os.path.join(L, "*.json")) + glob.glob(os.path.join(L, "*.jsonl")) + glob.glob(os.path.join(L, "*.txt")):
            shutil.copy(f, os.path.join(P, f"{tag}_lookup", os.path.basename(f)))
        st = newest(os.path.join(L, "**", "*kernel_stats.csv"))
        if st:
            shutil.copy(st, os.path.join(P, f"{tag}_lookup", "rocprofv3_kernel_stats.csv"))
    os.makedirs(os.path.join(P, f"{tag}_pmc"), exist_ok=True)
    pairs = []
    for m in ("f32", "f32x2", "f16"):
        for kind in ("fetch", "write"):
            shutil.rmtree(f"/tmp/pmc_{kind}_{m}", ignore_errors=True)
    for m in ("f32", "f32x2", "f16"):
        st = newest(os.path.join(F, f"trace_{m}", "**", "*kernel_stats.csv"))
        if st:
            shutil.copy(st, os.path.join(P, f"{tag}_rocprofv3_kernel_stats_hash19_{m}.csv"))
        for kind in ("fetch", "write"):
            f = newest(os.path.join(F, f"pmc_{kind}_{m}", "**", "*counter_collection.csv"))
            if not f:
                continue
            d = f"/tmp/pmc_{kind}_{m}"
            os.makedirs(d)
            shutil.copy(f, d)
            rd = csv.DictReader(open(f))
            rows = [r for r in rd if "rn::" in r["Kernel_Name"]]
            with open(os.path.join(P, f"{tag}_pmc", f"{kind}_size_hash19_{m}_rn_kernels.csv"), "w", newline="") as o:
                w = csv.DictWriter(o, fieldnames=rd.fieldnames)
                w.writeheader()
                w.writerows(rows)
        pairs += [f"/tmp/pmc_fetch_{m}", f"/tmp/pmc_write_{m}"]
    if all(os.path.isdir(d) and os.listdir(d) for d in pairs):      # only with a complete set of counter passes
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(P, "traffic.json")] + pairs, check=True,
                       stdout=subprocess.DEVNULL)
    print(sorted(os.listdir(P)))


if __name__ == "__main__":
    main()
