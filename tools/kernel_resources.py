#!/usr/bin/env python3
"""Registers, scratch (spills) and LDS of every kernel in built objects (default build/obj/*.o), from the code objects' metadata
notes: `python tools/kernel_resources.py [objects] > file` -- diff two builds to see what a change cost a kernel's register budget."""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"
TARGET = "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"


def device_code_object(path: str, td: str):
    """the gfx950 code object bundled in a host object / shared library, or None"""
    co = os.path.join(td, "dev.co")
    fat = os.path.join(td, "fat.bin")
    r = subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], capture_output=True)
    if r.returncode != 0 or not os.path.isfile(fat) or os.path.getsize(fat) == 0:
        return None
    r = subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", TARGET, f"--output={co}"],
                       capture_output=True)
    return co if r.returncode == 0 and os.path.isfile(co) and os.path.getsize(co) else None


def main():
    paths = sys.argv[1:] or sorted(glob.glob("build/obj/*.o"))       # one code object per translation unit
    rows = []
    for path in paths:
        with tempfile.TemporaryDirectory() as td:
            co = device_code_object(path, td)
            if co is None:
                continue
            txt = subprocess.run([LLVM + "llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count:")[1:]:
            def g(k):
                m = re.search(rf"\.{k}:\s*(\S+)", blk)
                return m.group(1) if m else "?"
            rows.append((g("name"), g("vgpr_count"), blk.split("\n")[0].strip(), g("sgpr_count"), g("private_segment_fixed_size"),
                         g("group_segment_fixed_size"), g("vgpr_spill_count")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        print(f"{n[:160]:160s} vgpr {r[1]:>4s} agpr {r[2]:>4s} sgpr {r[3]:>4s} scratch {r[4]:>5s} lds {r[5]:>6s} spills {r[6]:>4s}")


if __name__ == "__main__":
    main()
