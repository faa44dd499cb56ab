#!/usr/bin/env python3
"""Which kernels of libm4ri_hip.so does the GPU parity suite launch?  (round 4: the 512-tile transposition was wrong for half a
round because no test reached its size threshold.)

  1. the kernels IN the library: `.amdhsa_kernel` names of the device assembly of csrc/*.hip (hipcc -S --cuda-device-only)
  2. the kernels LAUNCHED: a rocprofv3 --kernel-trace --stats run of `python3 -m pytest tests -m gpu`, *kernel_stats.csv
       python tools/kernel_coverage.py <dir with *kernel_stats.csv> [> profiles/rNN_kernel_coverage.txt]
Exit code 1 if a kernel of the library was never launched."""
import csv
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "m4ri-rust_amd", "csrc")


def library_kernels():
    names = set()
    for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
        asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", f],
                             capture_output=True, text=True).stdout
        mangled = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm, re.M)
        if mangled:
            dem = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True).stdout.split("\n")
            names.update(norm(d) for d in dem if d)
    return names


def norm(name):
    name = name.strip().strip('"')
    name = re.sub(r"\(.*$", "", name)          # argument list
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"\.kd$", "", name)
    return name.replace(" ", "")


def main():
    d = sys.argv[1]
    launched = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = norm(r["Name"])
            launched[k] = launched.get(k, 0) + int(r["Calls"])
    lib = library_kernels()
    missing = sorted(k for k in lib if k not in launched)
    print("# kernels of libm4ri_hip.so (device assembly of csrc/*.hip): %d; launched by `pytest tests -m gpu`: %d; never launched: %d"
          % (len(lib), len(lib) - len(missing), len(missing)))
    for k in sorted(lib):
        print("%10d  %s" % (launched.get(k, 0), k))
    if missing:
        print("# NEVER LAUNCHED:")
        for k in missing:
            print("#   " + k)
    return 1 if missing else 0


if __name__ == "__main__":
    sys.exit(main())
