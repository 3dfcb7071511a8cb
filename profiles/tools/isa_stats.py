#!/usr/bin/env python3
"""Per-kernel ISA statistics of the exact-shape builds: compiles rware_static.hip for the given table groups to assembly
(device only) and prints, per rw::rware_step_kernel instantiation, the instruction count, VGPRs, SGPRs, scratch and LDS.
Usage: isa_stats.py OUTDIR [group ...]   (default groups: 0).  Used to check that a refactor leaves the BASELINE builds'
instruction streams alone: diff two outputs."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "robotic-warehouse_amd", "csrc")


def main():
    out = sys.argv[1]
    groups = [int(g) for g in sys.argv[2:]] or [0]
    os.makedirs(out, exist_ok=True)
    procs = []
    for g in groups:
        s = os.path.join(out, f"static_g{g}.s")
        procs.append((g, s, subprocess.Popen(
            ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + CSRC, "-mllvm",
             "-amdgpu-kernarg-preload-count=16", f"-DRW_STATIC_GROUP={g}", "--cuda-device-only", "-S", "-o", s,
             os.path.join(CSRC, "rware_static.hip")])))
    for g, s, p in procs:
        assert p.wait() == 0
        lines = open(s).read().splitlines()
        starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN2rw17rware_step_kernel.*:\s*(;.*)?$", l)]
        for a, b in zip(starts, starts[1:] + [len(lines)]):
            body = lines[a:b]
            name = subprocess.run(["c++filt", lines[a].split(":")[0]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"^void rw::rware_step_kernel", "", name).split("(")[0]
            n_inst = sum(1 for l in body if re.match(r"^\s+[a-z_][a-z0-9_]*(\s|$)", l) and not l.strip().startswith((".", ";")))
            ops = [l.split()[0] for l in body if re.match(r"^\s+[a-z_][a-z0-9_]*(\s|$)", l) and not l.strip().startswith((".", ";"))]
            import hashlib
            oph = hashlib.sha1(" ".join(ops).encode()).hexdigest()[:10]
            def meta(key):
                for l in body:
                    m = re.match(rf"^; {key}: (\d+)", l)
                    if m:
                        return int(m.group(1))
                return -1
            print(f"g{g} {name}: inst {n_inst} vgpr {meta('NumVgprs')} sgpr {meta('NumSgprs')} scratch {meta('ScratchSize')} occ {meta('Occupancy')} ophash {oph}")


if __name__ == "__main__":
    main()
