"""Per-launch floor table of the scale-space stage (S1-S8) from the passes of tools/collect_floor.sh.

usage: pyramid_floor.py <gpurun_out/tag> [size=4096] [commit] > profiles/rNN_pyramid_floor.json

Every pyramid launch of one image, in program order (serial run: SSRLCV_SIFT_SERIAL=1, nothing beside it on the chip), with
  solo_us          median duration over the profiled images (rocprofv3 --kernel-trace)
  traffic_bytes    FETCH_SIZE (doubled: gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md) + WRITE_SIZE,
                   separate --pmc passes, KiB units
  algorithmic_bytes what the launch must move: 4 B read + 4 B written per level pixel (u8 source: 1 B per 4 pixels; + the
                   2x2 bin on level 3), 24 B read + 1 B written per pixel for the DoG / extrema pass
  copy_floor_us    traffic_bytes / COPY_RATE, the rate a six-stream copy reaches on this part (tools/stream_rate.hip)
  issue_floor_us   multiply-add slots the formulation issues (zero taps of a padded band included) / the rate the
                   instruction sustains alone on all SIMDs (tools/mfma4_rate.hip, mfma_rate.hip, valu_rate.hip)
  gap_us           solo_us - max(copy_floor_us, issue_floor_us): what the launch loses to neither floor
  mfma_busy, clock_ghz   matrix-pipe busy share and mean clock of the launch (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE)
"""
import collections
import csv
import glob
import json
import re
import statistics
import sys

COPY_RATE = 5.3e12      # B/s: float4 copy, grid-stride, 1024-8192 blocks (tools/stream_rate.hip; 6.0-6.65 one float4 per thread + nt)
# sustained multiply-add slots per second, all 1024 SIMDs (measured alone, random data):
RATE_MFMA4 = 1024 * 256 / 8.5 * 2.0e9     # v_mfma_f32_4x4x1_16b: 256 slots per 8.5 cycles at the ~2.0 GHz it runs at
RATE_MFMA16 = 1024 * 1024 / 35.6 * 2.0e9  # v_mfma_f32_16x16x4 fed from LDS: 1024 slots per 35.6 cycles
RATE_VALU = 1024 * 64 / 4.2 * 2.3e9       # v_fma_f32 with an SGPR weight: 4.2 cycles per wave64 instruction
RATE_VALU_ANY = 1024 / 2.0 * 2.3e9        # wave64 vector instructions per second at one per 2 clocks per SIMD

PYR = ("k_init_minmax", "k_upsample2x", "k_add_border", "k_gauss_strip", "k_gauss_fused", "k_gauss_mfma2", "k_gauss_rm", "k_gauss_wm",
       "k_gauss_tile", "k_gauss_pair", "k_bin2x", "k_dogx", "k_dog_finalize")
CONV = ("k_gauss_strip", "k_gauss_fused", "k_gauss_mfma2", "k_gauss_rm", "k_gauss_wm", "k_gauss_tile", "k_gauss_pair")
TAPS = (13, 17, 23, 33, 47, 65)


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def is_pyr(name):
    return any(t in name for t in PYR)


def images_of(rows):
    """rows: (dispatch id, kernel name, value...) in dispatch order -> list of per-image lists, cut at k_init_minmax"""
    out = []
    for r in rows:
        if "k_init_minmax" in r[1]:
            out.append([])
        if out:
            out[-1].append(r)
    return out


def load_trace(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        if is_pyr(r["Kernel_Name"]):
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                         int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]), int(r["VGPR_Count"]), int(r["LDS_Block_Size"])))
    rows.sort()
    return images_of(rows)


def load_pmc(d, counters):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        return None
    per = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] in counters and is_pyr(r["Kernel_Name"]):
            k = int(r["Dispatch_Id"])
            e = per.setdefault(k, [k, r["Kernel_Name"], {}, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3])
            e[2][r["Counter_Name"]] = e[2].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return images_of([tuple(v) for _, v in sorted(per.items())])


def slots_per_px(name):
    """multiply-add slots per output pixel the formulation issues (both passes), and the rate that prices them"""
    m = re.search(r"<(\d+)", name)
    r = int(m.group(1)) if m else 0
    if "k_gauss_strip" in name or "k_gauss_fused" in name:
        return 2 * (2 * r + 1), RATE_VALU
    if "k_gauss_rm" in name or "k_gauss_wm" in name:
        return 2 * (4 + 2 * r), RATE_MFMA4
    if "k_gauss_mfma2" in name:
        return 2 * (16 + 2 * r), RATE_MFMA16
    if "k_gauss_tile" in name:
        hr = (64 + 2 * r + 15) // 16 * 16
        return (16 + 2 * r) * (hr / 64.0 + 1.0), RATE_MFMA16
    return 0, 1.0


def main():
    d = sys.argv[1]
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    commit = sys.argv[3] if len(sys.argv) > 3 else None
    trace = load_trace(d + "/trace")
    fetch = load_pmc(d + "/pmc_FETCH_SIZE", ("FETCH_SIZE",))
    write = load_pmc(d + "/pmc_WRITE_SIZE", ("WRITE_SIZE",))
    mfma = load_pmc(d + "/pmc_MFMA", ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES"))
    n = len(trace[0])
    trace = [im for im in trace if len(im) == n]
    rows = []
    conv_i = 0
    dog_i = 0
    for i in range(n):
        name = trace[0][i][1]
        sn = short(name)
        solo = statistics.median(im[i][2] for im in trace)
        row = collections.OrderedDict(launch=i, kernel=sn, grid_threads=trace[0][i][3], vgprs=trace[0][i][4], lds_bytes=trace[0][i][5], solo_us=round(solo, 1),
                                      solo_min_us=round(min(im[i][2] for im in trace), 1), solo_max_us=round(max(im[i][2] for im in trace), 1))

        def counter(imgs, c):
            if not imgs:
                return None
            vals = [im[i][2].get(c) for im in imgs if len(im) == n and short(im[i][1]) == sn]
            vals = [v for v in vals if v is not None]
            return statistics.median(vals) if vals else None
        fb, wb = counter(fetch, "FETCH_SIZE"), counter(write, "WRITE_SIZE")
        traffic = (2 * 1024 * fb + 1024 * wb) if fb is not None and wb is not None else None
        alg = None
        issue = None
        if "k_gauss_pair" in name:  # levels 0 + 1 of an octave in one launch: the level between them is written, not read back
            o, b = conv_i // 6, conv_i % 6
            conv_i += 2
            px = (2 * size >> o) ** 2
            row.update(octave=o, level="%d+%d" % (b, b + 1), taps="13+17", pixels=px)
            alg = px * 8 + (px // 4 if (o == 0 and "true" in name) else px * 4)
            if "_rm" in name:   # two register-marching engines, level a on 256 of every 240 columns, level b on 240 of 256 lanes
                spp, rate = (2 * 16 + 2 * 20) * 256.0 / 240.0, RATE_MFMA4
            else:
                spp, rate = (2 * 13) * 256.0 / 240.0 + 2 * 17, RATE_VALU
            issue = px * spp / rate * 1e6
            row["useful_mac_share"] = round(2 * (13 + 17) / spp, 3)
        elif any(t in name for t in CONV):
            o, b = conv_i // 6, conv_i % 6
            conv_i += 1
            px = (2 * size >> o) ** 2
            row.update(octave=o, level=b, taps=TAPS[b], pixels=px)
            alg = px * 8
            if o == 0 and b == 0:
                alg = px * 4 + px // 4  # u8 source read once
            if b == 3 and o < 3:
                alg += px  # the 2x2 bin (next octave's input)
            spp, rate = slots_per_px(name)
            issue = px * spp / rate * 1e6
            row["useful_mac_share"] = round(2 * TAPS[b] / spp, 3) if spp else None
        elif "k_dogx" in name:
            o = dog_i
            dog_i += 1
            px = (2 * size >> o) ** 2
            row.update(octave=o, pixels=px)
            alg = px * 25
            issue = px * 93 / 64 / RATE_VALU_ANY * 1e6  # ~93 lane-instructions per pixel (ISA count)
        if traffic is not None:
            row["traffic_bytes"] = int(traffic)
            row["copy_floor_us"] = round(traffic / COPY_RATE * 1e6, 1)
        if alg is not None:
            row["algorithmic_bytes"] = int(alg)
            row["algorithmic_floor_us"] = round(alg / COPY_RATE * 1e6, 1)
            if traffic is not None:
                row["traffic_over_algorithmic"] = round(traffic / alg, 3)
        if issue is not None:
            row["issue_floor_us"] = round(issue, 1)
        busy, gui = counter(mfma, "SQ_VALU_MFMA_BUSY_CYCLES"), counter(mfma, "GRBM_GUI_ACTIVE")
        if busy is not None and gui:
            row["mfma_busy"] = round(busy / (gui / 8 * 1024), 3)
            durs = [im[i][3] for im in mfma if len(im) == n]
            row["clock_ghz"] = round(gui / 8 / (statistics.median(durs) * 1e3), 2)
        floors = [row.get("copy_floor_us", 0.0), row.get("issue_floor_us", 0.0)]
        if max(floors) > 0:
            row["gap_us"] = round(solo - max(floors), 1)
        rows.append(row)
    tot = lambda k: round(sum(r.get(k, 0.0) or 0.0 for r in rows), 1)
    o0chain = [r for r in rows if r.get("octave") == 0 and "level" in r and (isinstance(r["level"], str) or r["level"] <= 3)]
    out = collections.OrderedDict(
        source="tools/collect_floor.sh + tools/pyramid_floor.py: SSRLCV_SIFT_SERIAL=1 tools/bench_sift_stages.py --size %d --scene (developer build); "
               "rocprofv3 --kernel-trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE, --pmc SQ_VALU_MFMA_BUSY_CYCLES ... in separate passes; "
               "median over %d images per launch" % (size, len(trace)),
        commit=commit, size=size, copy_rate_TBps=COPY_RATE / 1e12,
        issue_rates_TMACps={"v_mfma_f32_4x4x1": round(RATE_MFMA4 / 1e12, 1), "v_mfma_f32_16x16x4 (LDS-fed)": round(RATE_MFMA16 / 1e12, 1), "v_fma_f32 (SGPR weight)": round(RATE_VALU / 1e12, 1)},
        serial_sum_us=tot("solo_us"), traffic_bytes=int(sum(r.get("traffic_bytes", 0) for r in rows)), algorithmic_bytes=int(sum(r.get("algorithmic_bytes", 0) for r in rows)),
        copy_floor_sum_us=tot("copy_floor_us"), floor_sum_us=round(sum(max(r.get("copy_floor_us", 0.0), r.get("issue_floor_us", 0.0)) for r in rows), 1),
        gap_sum_us=tot("gap_us"),
        octave0_levels_0_3=dict(solo_us=round(sum(r["solo_us"] for r in o0chain), 1), floor_us=round(sum(max(r.get("copy_floor_us", 0.0), r.get("issue_floor_us", 0.0)) for r in o0chain), 1)),
        launches=rows)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
