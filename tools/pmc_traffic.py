
"""HBM traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes): FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B), both counters are
in KiB.  usage: pmc_traffic.py <fetch_dir> <write_dir> <images> [commit] > profiles/rNN_pyramid_traffic.json"""
import collections
import csv
import glob
import json
import sys

PYRAMID = ("k_upsample2x", "k_add_border", "k_gauss_pair", "k_gauss_strip", "k_gauss_fused", "k_gauss_mfma", "k_gauss_rm", "k_gauss_tile", "k_bin2x", "k_dogx", "k_dog_finalize", "k_dog", "k_init_minmax")


def short(name):
    for tok in ("k_upsample2x", "k_add_border", "k_gauss_pair", "k_gauss_strip", "k_gauss_fused", "k_gauss_mfma", "k_gauss_rm", "k_gauss_tile", "k_bin2x", "k_dogx", "k_dog_finalize", "k_dog", "k_init_minmax", "k_descriptors",
                "k_thetas", "k_polar", "k_extrema_flags", "k_refine", "k_scatter", "k_count", "k_scan", "k_flag_",
                "k_book_", "k_state_reset"):
        if tok in name:
            return tok
    return name[:60]


def collect(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(float)
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = short(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in calls.items()}


def main():
    fetch_dir, write_dir, images = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, calls = collect(fetch_dir, "FETCH_SIZE")
    write, _ = collect(write_dir, "WRITE_SIZE")
    per = {}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, 0) + write.get(k, 0))):
        per[k] = {"launches": calls.get(k, 0), "fetch_bytes_corrected": 2 * 1024 * fetch.get(k, 0.0),
                  "write_bytes": 1024 * write.get(k, 0.0)}
    pf = sum(v["fetch_bytes_corrected"] for k, v in per.items() if k in PYRAMID) / images
    pw = sum(v["write_bytes"] for k, v in per.items() if k in PYRAMID) / images
    print(json.dumps({
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/bench_sift_stages.py at 4096x4096; "
                  "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide streaming reads); KiB units; "
                  "per-kernel totals are over all %d profiled images" % images,
        "images": images, "commit": sys.argv[4] if len(sys.argv) > 4 else None,
        "pyramid_stage_bytes_per_image": pf + pw, "fetch_bytes_corrected": pf, "write_bytes": pw,
        "per_kernel": per}, indent=1))


if __name__ == "__main__":
    main()
