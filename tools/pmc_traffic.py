"""Parse two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/traffic.json: HBM-side bytes per launch of the dominant kernels.

Units / corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half
the bytes of a wide coalesced streaming read, so it is doubled (an upper bound for this kernel's mix of scalar
coalesced reads and L2-resident gathers -- uncalibrated for that pattern); WRITE_SIZE is exact.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json
"""
import collections
import csv
import glob
import json
import sys

KEYS = {"nerf_fused": "k_nerf_fused", "grid_encode_xyz": "k_grid_fwd_sample<float, 3u", "torso_fused": "k_torso_fused",
        "head_march": "k_head_march", "head_composite": "k_head_composite"}


def collect(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    fetch_dir, write_dir, dst = sys.argv[1:4]
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    res = {}
    for key, pat in KEYS.items():
        fv = [v for k, vs in fetch.items() if pat in k for v in vs]
        wv = [v for k, vs in write.items() if pat in k for v in vs]
        if not fv or not wv:
            continue
        f_kib, w_kib = sum(fv) / len(fv), sum(wv) / len(wv)
        res[key] = dict(launches_profiled=len(fv), fetch_size_kib_per_launch_raw=f_kib, write_size_kib_per_launch=w_kib,
                        hbm_bytes_per_launch=(2.0 * f_kib + w_kib) * 1024.0,
                        hbm_bytes_per_launch_uncorrected=(f_kib + w_kib) * 1024.0,
                        note="mean over every launch of the kernel in the profiled bench run (incl. zero-sample launches)")
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
