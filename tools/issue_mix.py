#!/usr/bin/env python3
"""tools/issue_mix.py [out.json] -- the vector instructions of every kernel of libmirt by ISSUE CLASS, and the issue ceiling that mix allows.

tools/ubench.hip (profiles/r04_ubench.txt, 8 waves per SIMD) measures what a SIMD of this chip issues per clock, one opcode at a time:

    class   wave-instr / clk / SIMD   cycles   members
    fast    0.425                     2.35     v_mul_f32 v_add_f32 v_sub_f32 v_subrev_f32 v_fma_f32 v_fmac_f32 v_mov_b32 -- with every operand a
                                               vector register or an inline constant (the guide's "2 cycles, SIMD-32")
    int     0.343                     2.92     v_and_b32 v_or_b32 v_xor_b32 v_not_b32 v_add_u32 v_sub_u32 v_subrev_u32, same operand rule
                                               (v_and / v_xor / v_add_u32 measured; the others assumed to share their path)
    slow    0.236                     4.24     everything else that was probed -- v_pk_* (two results each), compares, v_cndmask, min / max /
                                               med3, conversions, shifts, v_lshl_add, v_bfe, v_mul_lo_u32, v_mad_u32_u24, DPP moves, v_div_* --
                                               and ANY instruction with a scalar-register, vcc / exec or literal operand, or a DPP / SDWA modifier
    trans   0.121                     8.3      v_rcp v_rsq v_sqrt v_exp v_log v_sin v_cos

This script compiles the device sources to assembly with the Makefile's flags (no GPU needed), counts the classes per kernel and
prints  ceiling = instructions / sum(cycles)  -- a STATIC mix (every instruction of the kernel's text once, loops not weighted), which
is what bench.py prices `valu_issue` against instead of the four-cycle 0.24 it used through round 3.  The digest of the device
sources is stored with it (bench.csrc_digest), so that a stale table is seen."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "cpp-raytracer-rasterizer_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-w", "--cuda-device-only", "-S", "-x", "hip"]
DEVICE_SOURCES = ["rt_kernels.hip", "rt_tile.hip", "rt_binned.hip", "rt_trace.hip", "raster_kernels.hip", "dof_kernel.hip", "bin_bucket_sort.hip"]
CYCLES = {"fast": 1 / 0.425, "int": 1 / 0.343, "slow": 1 / 0.236, "trans": 1 / 0.121}
FAST = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32"}
INT = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32"}
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
INLINE = re.compile(r"^-?(\d+|0\.5|1\.0|2\.0|4\.0)$")


def classify(line):
    op = line.split()[0]
    base = re.sub(r"_e(32|64)$", "", op)
    if base.startswith(TRANS):
        return "trans"
    if base not in FAST and base not in INT:
        return "slow"
    if "_dpp" in op or "_sdwa" in op or " row_" in line or "quad_perm" in line or "dst_sel" in line:
        return "slow"
    operands = [o.strip() for o in line[len(op):].split(";")[0].split(",")]
    for o in operands[1:]:                              # sources
        o = o.strip("|").lstrip("-").strip("|")
        o = re.sub(r"^(abs|neg)\((.*)\)$", r"\2", o)
        if re.match(r"^v(\d+|\[\d+:\d+\])$", o):
            continue
        tok = o.split()[0] if o else ""
        if INLINE.match(tok):
            v = tok.lstrip("-")
            if "." in v or (v.isdigit() and int(v) <= 64):
                continue
        return "slow"                                   # s<N>, vcc, exec, literals, symbols
    return "fast" if base in FAST else "int"


def kernels_of(asm):
    names = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, flags=re.M))
    lines = asm.split("\n")
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(\S+):\s*(;.*)?$", lines[i])
        if m and m.group(1) in names:
            name = m.group(1)
            counts = {"fast": 0, "int": 0, "slow": 0, "trans": 0, "packed": 0, "salu": 0, "lds": 0, "vmem": 0}
            i += 1
            while i < len(lines) and ".Lfunc_end" not in lines[i]:
                t = lines[i].strip()
                i += 1
                if not t or t[0] in ";." or t.endswith(":"):
                    continue
                if t.startswith("v_"):
                    counts[classify(t)] += 1
                    if t.startswith("v_pk_"):
                        counts["packed"] += 1
                elif t.startswith("s_"):
                    counts["salu"] += 1
                elif t.startswith("ds_"):
                    counts["lds"] += 1
                elif t.startswith(("buffer_", "global_", "flat_", "scratch_")):
                    counts["vmem"] += 1
            out[name] = counts
        else:
            i += 1
    return out


def demangle(names):
    try:
        r = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True)
        return [re.sub(r"^void ", "", re.sub(r"\((?!anonymous).*$", "", x)).replace("(anonymous namespace)::", "") for x in r.stdout.strip().split("\n")]
    except Exception:
        return names


def main():
    import bench
    table = {}
    with tempfile.TemporaryDirectory() as tmp:
        for src in DEVICE_SOURCES:
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call(["hipcc"] + FLAGS + [os.path.join(CSRC, src), "-o", out])
            ks = kernels_of(open(out).read())
            for mangled, pretty in zip(ks.keys(), demangle(list(ks.keys()))):
                c = ks[mangled]
                valu = c["fast"] + c["int"] + c["slow"] + c["trans"]
                if not valu:
                    continue
                cyc = sum(c[k] * CYCLES[k] for k in ("fast", "int", "slow", "trans"))
                table[pretty] = dict(c, valu=valu, cycles_per_instruction=round(cyc / valu, 3), ceiling=round(valu / cyc, 4), source=src)
    doc = {"_comment": "static issue-class mix of every kernel's text and the issue ceiling it allows (tools/issue_mix.py; class rates: profiles/r04_ubench.txt)",
           "csrc_sha16": bench.csrc_digest(), "class_cycles": {k: round(v, 3) for k, v in CYCLES.items()}, "kernels": table}
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "%s_issue_mix.json" % bench.ROUND)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    for k, v in sorted(table.items(), key=lambda kv: -kv[1]["valu"])[:16]:
        print("%-44s VALU %5d  fast %5d int %4d slow %5d (packed %4d) trans %3d   %.2f cycles each -> ceiling %.3f" %
              (k[:44], v["valu"], v["fast"], v["int"], v["slow"], v["packed"], v["trans"], v["cycles_per_instruction"], v["ceiling"]))
    print("wrote", path)


if __name__ == "__main__":
    main()
