#!/usr/bin/env python3
"""Static instruction mix of one kernel of libsqmc_gpu from the compiler's assembly (no GPU needed): where a VALU-bound kernel's
instructions are.  usage: tools/isa_mix.py <substring of the mangled kernel name> [assembly file]
The assembly comes from:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude --save-temps -c sqmc_amd/csrc/sqmc_gpu.hip"""
import collections, re, sys


def main(key, path):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if key in l.split(":")[0] and re.match(r"^_Z\w+:", l)][0]
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    ins = [l.split()[0] for l in lines[start + 1:end] if l.startswith("\t") and l.split() and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)

    def group(i):
        if re.match(r"v_(div|rcp|rsq|sqrt|trig|frexp|ldexp).*_f64", i): return "f64 divide/rcp family"
        if re.match(r"v_.*_f64", i): return "f64 arithmetic"
        if re.match(r"v_mul_(lo|hi)_|v_mad_(u|i)64|v_mul_u32|v_mad_u32", i): return "integer multiply"
        if i.startswith("v_cmp") or i.startswith("v_cndmask"): return "compare / select"
        if i.startswith("v_"): return "other VALU"
        if i.startswith("s_"): return "SALU / control"
        if re.match(r"(global|flat|buffer|scratch)_", i): return "vector memory"
        if i.startswith("ds_"): return "LDS"
        return "other"
    g = collections.Counter()
    for i, n in c.items():
        g[group(i)] += n
    print("%s: %d static instructions" % (lines[start].split(":")[0][:60], len(ins)))
    for k, n in g.most_common():
        print("  %-26s %6d" % (k, n))
    print("  most frequent:", ", ".join("%s %d" % kv for kv in c.most_common(24)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "/tmp/isa/sqmc_gpu-hip-amdgcn-amd-amdhsa-gfx950.s")
