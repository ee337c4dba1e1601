"""Parse two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
profiles/traffic.json: HBM-side bytes per launch of the dominant kernels.

Units / corrections (MI355X_MICROARCH.md, HBM section): the counters are in KiB; on gfx950 FETCH_SIZE reports half
the bytes of a wide coalesced streaming read, so it is doubled (an upper bound for this kernel's mix of scalar
coalesced reads and L2-resident gathers -- uncalibrated for that pattern); WRITE_SIZE is exact.

    python tools/pmc_traffic.py profiles/traffic.json FETCH_DIR WRITE_DIR [FETCH_DIR2 WRITE_DIR2 ...]
(one FETCH/WRITE pair per profiled bench configuration; kernels found in a later pair do not overwrite earlier ones)
"""
import collections
import csv
import glob
import json
import sys

KEYS = {"nerf_fused": "k_nerf_fused<", "nerf_fused_h16": "k_nerf_fused_h16<", "nerf_fused_x2": "k_nerf_fused_x2<", "grid_encode_xyz": "k_grid_fwd_sample<float, 3u", "torso_fused": "k_torso_fused",
        "head_march": "k_head_march", "head_composite": "k_head_composite"}


def collect(d, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    dst, dirs = sys.argv[1], sys.argv[2:]
    res = {}
    for fetch_dir, write_dir in zip(dirs[0::2], dirs[1::2]):
        fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
        pairs(res, fetch, write)
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps(res, indent=1))


def pairs(res, fetch, write):
    for key, pat in KEYS.items():
        if key in res:
            continue
        fv = [v for k, vs in fetch.items() if pat in k for v in vs]
        wv = [v for k, vs in write.items() if pat in k for v in vs]
        if not fv or not wv:
            continue
        f_kib, w_kib = sum(fv) / len(fv), sum(wv) / len(wv)
        # launches with work: a zero-sample launch of the frame loop exits before it stages its weights (< 64 KiB fetched)
        work = sum(1 for v in fv if v >= 64.0)
        res[key] = dict(launches_profiled=len(fv), launches_with_work=work, fetch_size_kib_per_launch_raw=f_kib,
                        write_size_kib_per_launch=w_kib, hbm_bytes_per_launch=(2.0 * f_kib + w_kib) * 1024.0,
                        hbm_bytes_per_launch_with_work=(2.0 * sum(fv) + sum(wv)) * 1024.0 / max(work, 1),
                        hbm_bytes_per_launch_uncorrected=(f_kib + w_kib) * 1024.0,
                        note="hbm_bytes_per_launch: mean over every launch of the kernel in the profiled bench run (incl. zero-sample "
                             "launches, whose share depends on the run's warm-up / timed mix); _with_work: the same bytes over the "
                             "launches that had samples")


if __name__ == "__main__":
    main()
