#!/usr/bin/env python3
"""Issue-cycle model of a kernel's loops from the compiler's own assembly (runs on the CPU box; no GPU needed).

On gfx950 a wave's vector instructions do not all cost the same issue time (tools/microbench_imul.hip,
profiles/r04_microbench_valu_issue.txt): f32 fma / mul / add / sub, and / or / xor / mov, logical and arithmetic right shifts,
32-bit integer add / sub issue at the full rate (class 2: ~2 cycles per wave), everything else -- v_cndmask, every compare,
min / max / med3, left shifts, bfe, perm, every cvt, every VOP3-only integer op, SDWA and DPP forms, fma_mix, integer
multiplies -- at about half of it (class 4), rcp / rsq / sqrt at a quarter (class 8).  Counting instructions alone
(SQ_INSTS_VALU against 1228.8 G/s) therefore understates how busy the issue port is: this tool weights them.

Usage: tools/isa_cycles.py <mangled-name-prefix> [first-last ...]   (line ranges relative to the kernel's label;
       without ranges: every loop of the kernel)      --asm FILE reuses an assembly listing instead of compiling."""
import collections, os, re, subprocess, sys

FULL = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_readfirstlane_b32", "v_add_f64", "v_mul_f64", "v_fma_f64"}
QUARTER = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32",
           "v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"}


def issue_class(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if base in QUARTER:
        return 8
    if base in FULL and not op.endswith(("_sdwa", "_dpp")):
        return 2
    return 4


def listing(path):
    if path:
        return open(path).read().split("\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = "/tmp/drt_isa_cycles.s"
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", f"-I{root}/include", "-S",
                    "--cuda-device-only", f"{root}/differentiable-renderer_amd/csrc/drt_hip.hip", "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def histogram(body):
    cnt = collections.Counter()
    other = collections.Counter()
    for l in body:
        m = re.match(r"\s+(v_[a-z0-9_]+)", l)
        if m:
            cnt[m.group(1)] += 1
            continue
        m = re.match(r"\s+(s_|ds_|global_|scratch_|flat_|buffer_)", l)
        if m:
            other[m.group(1)] += 1
    by = collections.Counter()
    for op, n in cnt.items():
        by[issue_class(op)] += n
    return cnt, by, other


def report(tag, body):
    cnt, by, other = histogram(body)
    n = sum(cnt.values())
    cyc = sum(c * k for c, k in by.items())
    print(f"== {tag}: {n} vector instructions, {cyc} issue cycles per wave in the model ({cyc / max(1, n):.2f} per instruction); "
          f"full rate {by[2]}, half rate {by[4]}, quarter rate {by[8]}; scalar {other['s_']}, LDS {other['ds_']}, "
          f"memory {other['global_'] + other['scratch_'] + other['flat_'] + other['buffer_']}")
    merged = collections.Counter()
    for k, v in cnt.items():
        if issue_class(k) == 4:
            merged[re.sub(r"_(e32|e64)$", "", k)] += v
    print("   half rate: " + "  ".join(f"{k} {v}" for k, v in merged.most_common(14)))


def main():
    args = sys.argv[1:]
    asm = None
    if "--asm" in args:
        i = args.index("--asm")
        asm = args[i + 1]
        del args[i:i + 2]
    name, ranges = args[0], args[1:]
    lines = listing(asm)
    starts = [i for i, l in enumerate(lines) if l.startswith(name) and l.split(";")[0].strip().endswith(":")]
    if not starts:
        sys.exit(f"no kernel whose mangled name starts with {name}")
    start = starts[0]
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    print(f"# {lines[start].split(':')[0]}")
    if ranges:
        for r in ranges:
            a, b = map(int, r.split("-"))
            report(f"lines {a}-{b}", body[a:b])
        return
    label = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            label[m.group(1)] = i
    seen = set()
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in label and label[m.group(1)] < i and (label[m.group(1)], i) not in seen:
            seen.add((label[m.group(1)], i))
            if i - label[m.group(1)] > 40:
                report(f"loop {m.group(1)}, lines {label[m.group(1)]}-{i}", body[label[m.group(1)]:i])


if __name__ == "__main__":
    main()
