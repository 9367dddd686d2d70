"""Reduces rocprofv3 --pmc passes of tools/bench_sift_stages.py (--scene --stages 7, SSRLCV_SIFT_SERIAL=1) to the
describe-stage figures bench.py quotes, per kernel and for the stage, all from the PMC runs' own image:
  VALU wave-instructions (SQ_INSTS_VALU) and their rate against the issue peak (one wave64 instruction per 2 clocks per
  SIMD-32: 1024 SIMDs x 2.4 GHz / 2), HBM bytes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, both
  counters in KiB), duration, and the HBM fraction = algorithmic bytes / time / 8 TB/s.
usage: pmc_describe.py <fetch_dir> <write_dir> <sq_dir> <images> <features_per_image> <commit> > profiles/rNN_describe_pmc.json"""
import collections
import csv
import glob
import json
import sys

DESCRIBE = ("k_count", "k_scan", "k_scatter", "k_refine", "k_flag_", "k_book_", "k_state_reset", "k_polar",
            "k_build_ranges", "k_thetas", "k_desc_consts", "k_descriptors")
VALU_PEAK_GINST = 1024 * 2.4 / 2.0
HBM_PEAK_GBS = 8000.0


def short(name):
    for tok in DESCRIBE:
        if tok in name:
            return tok
    return None


def collect(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    dur = collections.defaultdict(float)
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    return tot, dur


def main():
    fetch_dir, write_dir, sq_dir = sys.argv[1:4]
    images, features, commit = int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    fetch, _ = collect(fetch_dir)
    write, _ = collect(write_dir)
    sq, dur = collect(sq_dir)
    side = 4096
    p = 5.3125 * side * side
    # algorithmic bytes per kernel (per image): polar tables read four Gaussian levels and write three 8-byte tables; the
    # flag compaction reads the flag bytes twice; the sampling kernels read every window sample once (8 B) -- counted from
    # the feature count with the mean window of the run is not attempted: their bound is the VALU issue rate
    algo = {"k_polar": p * (16 + 24), "k_count": 2 * p / 2, "k_scatter": 2 * p / 2 + features * 32 * 4, "k_descriptors": features * 152}
    per = {}
    for k in DESCRIBE:
        if k in sq or k in fetch:
            ms = dur.get(k, 0.0) / images
            valu = sq[k].get("SQ_INSTS_VALU", 0.0) / images
            per[k] = {"ms_per_image_under_pmc": ms,
                      "valu_wave_instructions_per_image": valu,
                      "salu_wave_instructions_per_image": sq[k].get("SQ_INSTS_SALU", 0.0) / images,
                      "lds_instructions_per_image": sq[k].get("SQ_INSTS_LDS", 0.0) / images,
                      "fetch_bytes_corrected_per_image": 2 * 1024 * fetch[k].get("FETCH_SIZE", 0.0) / images,
                      "write_bytes_per_image": 1024 * write[k].get("WRITE_SIZE", 0.0) / images,
                      "valu_frac": (valu / (ms * 1e6) / VALU_PEAK_GINST) if ms > 0 else None,
                      "hbm_frac": (algo[k] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (k in algo and ms > 0) else None}
    valu = sum(v["valu_wave_instructions_per_image"] for v in per.values())
    hbm = sum(v["fetch_bytes_corrected_per_image"] + v["write_bytes_per_image"] for v in per.values())
    ms_total = sum(v["ms_per_image_under_pmc"] for v in per.values())
    # algorithmic bytes of the stage: four Gaussian levels read once for the 3 polar tables (16 B per scale-space pixel),
    # the tables written and read once (2 x 24 B), flags (2 B), plus 152 B per feature written and 32 B per key point of
    # list traffic
    algo_stage = p * (16 + 48 + 2) + features * (152 + 64)
    print(json.dumps({
        "source": "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS} (three separate passes) of "
                  "SSRLCV_SIFT_SERIAL=1 python3 tools/bench_sift_stages.py --size 4096 --scene --stages 7",
        "commit": commit, "images": images, "features_per_image": features,
        "valu_peak_ginst": VALU_PEAK_GINST,
        "valu_wave_instructions_per_image": valu, "valu_wave_instructions_per_feature": valu / features,
        "ms_per_image_serial_under_pmc": ms_total,
        "valu_frac_serial": valu / (ms_total * 1e6) / VALU_PEAK_GINST,
        "hbm_bytes_per_image": hbm, "algorithmic_bytes_per_image": algo_stage,
        "hbm_frac_serial": algo_stage / (ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "per_kernel": per}, indent=1))


if __name__ == "__main__":
    main()
