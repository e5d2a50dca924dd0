#!/usr/bin/env python3
"""Compile csrc/drt_hip.hip for gfx950 (device only, to assembly) and print one line per kernel:
VGPRs, SGPRs, scratch, LDS, occupancy (-Rpass-analysis=kernel-resource-usage), plus -- for the
kernels named on the command line -- an instruction-class histogram of their ISA.

  python tools/kernel_resources.py [--keep DIR] [substring of a demangled kernel name ...]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "differentiable-renderer_amd", "csrc", "drt_hip.hip")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.strip().split("\n")


def short(d):
    d = re.sub(r"\(.*", "", d)
    d = d.replace("void ", "")
    return d


def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")): return "valu_trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64", "v_mad_i64")): return "valu_imul"
    if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_div", "v_rcp_f64", "v_trig", "v_cvt_f64", "v_cvt_f32_f64")): return "valu_f64"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"


def main():
    args = sys.argv[1:]
    keep = None
    if args and args[0] == "--keep":
        keep = args[1]
        args = args[2:]
    d = keep or tempfile.mkdtemp(prefix="drt_isa_")
    os.makedirs(d, exist_ok=True)
    asm = os.path.join(d, "drt.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
           "--cuda-device-only", "-S", "-o", asm, SRC, "-Rpass-analysis=kernel-resource-usage"] + \
          [a for a in os.environ.get("DRT_EXTRA_FLAGS", "").split() if a]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stderr)
        sys.exit(1)
    kernels = []
    cur = None
    for line in res.stderr.splitlines():
        m = re.search(r"remark: +(\w[\w \[\]/]*): +(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k in ("Function Name", "Name"):
            cur = {"name": v}
            kernels.append(cur)
        elif cur is not None:
            cur[k] = v
    names = demangle([k["name"] for k in kernels])
    print(f"{'kernel':78s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s} {'sgpr-spill':>10s}")
    for k, dn in zip(kernels, names):
        k["short"] = short(dn)
        print(f"{k['short'][:78]:78s} {k.get('VGPRs', '?'):>5s} {k.get('AGPRs', '?'):>5s} {k.get('TotalSGPRs', k.get('SGPRs', '?')):>5s} "
              f"{k.get('ScratchSize [bytes/lane]', '?'):>8s} {k.get('LDS Size [bytes/block]', '?'):>7s} {k.get('Occupancy [waves/SIMD]', '?'):>4s} {k.get('SGPRs Spill', '?'):>10s}")
    if not args:
        return
    text = open(asm).read()
    for k in kernels:
        if not any(a in k["short"] for a in args):
            continue
        m = re.search(r"^%s:[^\n]*\n(.*?)^\s*s_endpgm" % re.escape(k["name"]), text, re.S | re.M)
        if not m:
            continue
        hist = collections.Counter()
        for line in m.group(1).splitlines():
            line = line.strip()
            if not line or line.startswith((";", ".", "//")) or line.endswith(":"):
                continue
            hist[classify(line.split()[0])] += 1
        print(f"\n{k['short']}: static instruction mix")
        for c, n in hist.most_common():
            print(f"  {c:12s} {n}")
    print(f"\nassembly kept in {asm}")


if __name__ == "__main__":
    main()
