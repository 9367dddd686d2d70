"""Instruction-class breakdown of a kernel's loops from its ISA (static counts per loop body, innermost-last).
usage: hipcc ... -S --cuda-device-only file.hip -o file.s ; python tools/isa_classes.py file.s <kernel substring>"""
import collections
import re
import sys


def classify(lines):
    c = collections.Counter()
    for l in lines:
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") or re.match(r"^[.\w$]+:", l):
            continue
        op = l.split()[0]
        if op.startswith("ds_add") or op.startswith("ds_max") or op.startswith("ds_min"):
            c["lds_atomic"] += 1
        elif op.startswith("ds_"):
            c["lds_other"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
        elif op.startswith("s_waitcnt"):
            c["s_waitcnt"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            c["branch"] += 1
        elif op.startswith(("s_load", "s_buffer_load")):
            c["smem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith(("v_fma", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac", "v_mac", "v_mad_f32")):
            c["valu_fp_arith"] += 1
        elif op.startswith(("v_cmp", "v_cndmask", "v_max", "v_min")):
            c["valu_cmp_select_minmax"] += 1
        elif op.startswith(("v_cvt", "v_rcp", "v_exp", "v_log", "v_floor", "v_fract", "v_rndne", "v_ldexp", "v_frexp", "v_sqrt", "v_rsq", "v_trunc", "v_ceil")):
            c["valu_convert_transcendental"] += 1
        elif op.startswith(("v_mov", "v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_swap")):
            c["valu_move"] += 1
        elif op.startswith("v_mfma"):
            c["mfma"] += 1
        elif op.startswith("v_"):
            c["valu_integer_logic"] += 1
        else:
            c["other"] += 1
    return c


def main():
    s = open(sys.argv[1]).read()
    m = re.search(r"^(\S*%s\S*):" % re.escape(sys.argv[2]), s, re.M)
    name = m.group(1)
    i = s.index(name + ":")
    body = s[i:s.index("s_endpgm", i)].split("\n")
    labels = {}
    for k, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l.strip())
        if mm:
            labels[mm.group(1)] = k
    loops = []
    for k, l in enumerate(body):
        mm = re.match(r"\s*s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < k:
            loops.append((labels[mm.group(1)], k))
    loops.sort(key=lambda x: -(x[1] - x[0]))
    print(name)
    tot = classify(body)
    print("whole kernel: %d instructions %s" % (sum(tot.values()), dict(tot)))
    for lo, hi in loops[:8]:
        c = classify(body[lo:hi + 1])
        valu = sum(v for k, v in c.items() if k.startswith("valu"))
        print("loop lines %d-%d: %d instructions, VALU %d, SALU %d, LDS atomics %d, other LDS %d, VMEM %d | %s" %
              (lo, hi, sum(c.values()), valu, c["salu"], c["lds_atomic"], c["lds_other"], c["vmem"], dict(c)))


if __name__ == "__main__":
    main()
