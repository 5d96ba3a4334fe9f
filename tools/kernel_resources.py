#!/usr/bin/env python
"""Register / scratch / LDS figures of every device kernel of libcfmm_hip.so, from the ISA listing of a `-save-temps` build:

    make -C cfmm-routing-code_amd/csrc asm          # -> /tmp/cfmm_hip-hip-amdgcn-amd-amdhsa-gfx950.s  (~1 min)
    python tools/kernel_resources.py [/path/to/listing.s] [--scratch] [pattern ...]

--scratch: only kernels with a non-zero private segment (spills).  What VERDICT's ISA paragraph counts by hand."""
import re
import sys


def demangle_short(name):
    m = re.match(r"_ZN4cfmm\d+([a-z0-9_]+?)(I.*)?E[vP]", name)
    return name if not m else m.group(1) + (("<" + m.group(2)[1:] + ">") if m.group(2) else "")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args and args[0].endswith(".s") else "/tmp/cfmm_hip-hip-amdgcn-amd-amdhsa-gfx950.s"
    pats = [a for a in args if not a.endswith(".s")]
    only_scratch = "--scratch" in sys.argv
    s = open(path).read()
    rows = []
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", s, re.S):
        name, body = m.group(1), m.group(2)
        g = lambda k: int((re.search(r"\.%s:\s+(\d+)" % k, body) or [0, 0])[1])
        rows.append((name, g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
    for name, vg, ag, sg, scr, lds in sorted(rows):
        if only_scratch and not scr:
            continue
        if pats and not any(p in name for p in pats):
            continue
        print(f"{name[:110]:110} vgpr {vg:3d} agpr {ag:3d} sgpr {sg:3d} scratch {scr:4d} lds {lds}")
    print(f"{len(rows)} kernels, {sum(1 for r in rows if r[4])} with scratch")


if __name__ == "__main__":
    main()
