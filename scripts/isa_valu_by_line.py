#!/usr/bin/env python3
"""Static vector-instruction count of one kernel per source line (no GPU needed).

    python scripts/isa_valu_by_line.py k_refine_indel_waveILb1E [top] [v|s]

Compiles cutesv_amd/csrc/cutesv_hip.hip for gfx950 with line tables (-gline-tables-only -S), walks the kernel's assembly and
charges every v_* instruction to the source line of the last .loc directive.  The kernels of this library run at 55 - 86 % of the
vector ALUs' issue rate (profiles/r04_cfg*_insts.txt), so the instruction count is the time; this lists
where the instructions are (all template instantiations inlined into the kernel together, rare paths included: static, not
dynamic - weigh with the SQ_INSTS_VALU counters)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cutesv_amd", "csrc")


def main():
    want = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    unit = sys.argv[3] if len(sys.argv) > 3 else "v"          # "s": the scalar unit's instructions instead (s_waitcnt / s_nop excluded)
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                        "--cuda-device-only", "-gline-tables-only", "-S", "-o", asm, "cutesv_hip.hip"], cwd=CSRC, check=True,
                       stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(want), l)]
    if not starts:
        raise SystemExit("no kernel symbol contains %r" % want)
    src = {}
    cnt, ops = collections.Counter(), collections.Counter()
    cur = None
    for l in lines[starts[0]:]:
        if "s_endpgm" in l:
            break
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        m = re.match(r"\s+(%s_\w+)" % unit, l)
        if m and not m.group(1).startswith(("s_waitcnt", "s_nop", "s_endpgm")):
            cnt[cur] += 1
            ops[m.group(1)] += 1
    print("%s: %d %s instructions (static)" % (lines[starts[0]].split(":")[0], sum(cnt.values()), "scalar" if unit == "s" else "vector"))
    print("by opcode:", ", ".join("%s %d" % kv for kv in ops.most_common(12)))
    for (f, ln), c in cnt.most_common(top):
        if f not in src:
            p = os.path.join(CSRC, f)
            src[f] = open(p).read().split("\n") if os.path.exists(p) else None
        text = src[f][ln - 1].strip()[:110] if src[f] and 0 < ln <= len(src[f]) else ""
        print("%5d  %s:%d  %s" % (c, f, ln, text))


if __name__ == "__main__":
    main()
