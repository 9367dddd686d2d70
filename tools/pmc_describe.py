"""Reduces three rocprofv3 --pmc passes of tools/bench_sift_stages.py (--scene --stages 7) to the describe-stage figures
bench.py quotes: VALU wave-instructions per feature, HBM bytes per image (FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950, both counters in KiB) and per-kernel shares.
usage: pmc_describe.py <fetch_dir> <write_dir> <sq_dir> <images> <features_per_image> <commit> > profiles/rNN_describe_pmc.json"""
import collections
import csv
import glob
import json
import sys

DESCRIBE = ("k_extrema_flags", "k_count", "k_scan", "k_scatter", "k_refine", "k_flag_", "k_book_", "k_state_reset", "k_polar",
            "k_build_ranges", "k_thetas", "k_desc_consts", "k_descriptors")


def short(name):
    for tok in DESCRIBE + ("k_upsample2x", "k_add_border", "k_gauss_strip", "k_gauss_fused", "k_gauss_mfma", "k_gauss_tile", "k_bin2x", "k_dog",
                           "k_init_minmax"):
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
    per = {}
    for k in DESCRIBE:
        if k in sq or k in fetch:
            per[k] = {"ms_per_image_under_pmc": dur.get(k, 0.0) / images,
                      "valu_wave_instructions_per_image": sq[k].get("SQ_INSTS_VALU", 0.0) / images,
                      "salu_wave_instructions_per_image": sq[k].get("SQ_INSTS_SALU", 0.0) / images,
                      "lds_instructions_per_image": sq[k].get("SQ_INSTS_LDS", 0.0) / images,
                      "fetch_bytes_corrected_per_image": 2 * 1024 * fetch[k].get("FETCH_SIZE", 0.0) / images,
                      "write_bytes_per_image": 1024 * write[k].get("WRITE_SIZE", 0.0) / images}
    valu = sum(v["valu_wave_instructions_per_image"] for v in per.values())
    hbm = sum(v["fetch_bytes_corrected_per_image"] + v["write_bytes_per_image"] for v in per.values())
    # algorithmic bytes of the stage: the 5 raw DoG levels read once (20 B per scale-space pixel), the 3 polar tables
    # written and read once (2 x 24 B), flags (2 B), plus 152 B per feature written and 32 B per key point of list traffic
    side = 4096
    p = 5.3125 * side * side
    algo = p * (20 + 48 + 2) + features * (152 + 64)
    print(json.dumps({
        "source": "rocprofv3 --pmc {FETCH_SIZE | WRITE_SIZE | SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS} (three separate passes) of "
                  "SSRLCV_SIFT_SERIAL=1 python3 tools/bench_sift_stages.py --size 4096 --scene --stages 7",
        "commit": commit, "images": images, "features_per_image": features,
        "valu_wave_instructions_per_image": valu, "valu_wave_instructions_per_feature": valu / features,
        "hbm_bytes_per_image": hbm, "algorithmic_bytes_per_image": algo,
        "per_kernel": per}, indent=1))


if __name__ == "__main__":
    main()
