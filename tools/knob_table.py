#!/usr/bin/env python3
"""Generates the table of every M4RI_HIP_* environment variable the library reads:
     python tools/knob_table.py > /tmp/knobs.md     (tests/test_host_abi.py checks that INTEGRATION.md lists every knob)
A knob is a call env_int("M4RI_HIP_X", default) or getenv("M4RI_HIP_X") in m4ri-rust_amd/csrc; variables read only inside
#ifdef GF2K_DEV_VARIANTS blocks are marked "development builds only" (tools/libm4ri_hip_dev.so)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "m4ri-rust_amd", "csrc")


# what the variables of the shipped library do (the development ones carry the comment at their point of use)
DESCRIPTIONS = {
    "M4RI_HIP_DEVICES": "unset: the current device; `auto` / `all` / a list of ordinals: rows of A and C of a host product divided among devices; ONE ordinal pins every host entry point to that device (INTEGRATION 4c)",
    "M4RI_HIP_HOST_SMALL_WORK": "size dispatch of the drop-in entry points: products of at most this many word operations (default 2^20) run on the host; 0 = everything on the device",
    "M4RI_HIP_PIN_MIN_BYTES": "host blocks from this size on (default 1 MiB) are pinned with hipHostMalloc",
    "M4RI_HIP_PIN_CACHE_BYTES": "freed pinned blocks kept for reuse, in bytes (default 8 GiB); see gf2_mzd_prewarm",
    "M4RI_HIP_TRANSPOSE_GPU_MIN_BITS": "mzd_transpose on host matrices uses the device kernel from this many bits on (default 2^24)",
    "M4RI_HIP_ELIM_BLOCK_WORDS": "64-bit word columns per elimination block (default 32 = 2048 columns)",
    "M4RI_HIP_STRASSEN_FUSE3": "0: the round-1 level plans (pairs of fused levels) instead of three fused levels and a virtual fourth",
    "M4RI_HIP_STRASSEN_PAD": "0: shapes that do not divide run as plain M4RM instead of padded / peeled Strassen plans",
    "M4RI_HIP_STRASSEN_MAX_LEVELS": "cap of the automatic Strassen level count",
    "M4RI_HIP_STRASSEN_LEAF_MIN": "smallest leaf dimension the automatic level choice accepts (mzd_mul's cutoff argument overrides it)",
    "M4RI_HIP_SPLITK_WS_MIB": "cap of the per-stream scratch for partial tiles (stream-K / split-K), MiB",
    "M4RI_HIP_HOST_PIPELINE_BLOCKS": "row blocks of the host upload / multiply / download pipeline (default 4; < 2 switches it off)",
    "M4RI_HIP_APACK": "0: Strassen leaves of A stay row-major (no row-group-packed layout)",
    "M4RI_HIP_PLAIN_APACK": "0: plain products never pack A first",
    "M4RI_HIP_RESULT_SIDE_COLS": "a product into a NULL destination with at most this many columns (default 8) and >= 1 MiB of rows also comes back in its packed transposed form, from which mzd_transpose of that product is served (INTEGRATION 4b, 4d); 0 = never",
    "M4RI_HIP_ELIM_LOOKAHEAD": "0: the pivot search of an elimination step runs as its own launch instead of on an extra workgroup of the previous step's update launch (DESIGN 7.1)",
    "M4RI_HIP_ELIM_SPECULATE": "0: the trailing product of an elimination block waits for the block's record instead of being enqueued ahead of it",
    "M4RI_HIP_KERNEL_CENSUS_FILE": "path: the launch counts of the process (gf2_kernel_census) are appended to this file when the library is unloaded; the GPU test suite sets it so that kernels launched by its child processes count (tests/test_zz_kernel_census.py)",
    "M4RI_HIP_ELIM_FAULT": "test hook: 1 = update workgroup 0 of every look-ahead launch never raises its counters, so that the look-ahead workgroup's bounded wait runs out and the elimination reports the failure (tests/test_gpu_elim.py::test_lookahead_failure_is_reported_not_hung)",
    "M4RI_HIP_HOST_PLAN": "schedule of a large product on host matrices: 0 = the one the time model ends first (default), 1 = row blocks of A and C, 2-12 = the slab schedules in the numbering of gf2_host_plan_model (include/m4ri_hip.h); read per call (parity tests run every schedule)",
    "M4RI_HIP_M4RM_CFG": "force one tile-kernel variant (7, 8, 9-12, 20, 81, 82) for A/B runs; anything else is ignored with a message",
}


def knobs():
    found = {}
    for fn in sorted(os.listdir(SRC)):
        if not fn.endswith((".cpp", ".hip", ".inc", ".h")):
            continue
        depth_dev = []  # stack of booleans: is this #if level a GF2K_DEV_VARIANTS block
        func = "file scope"
        for ln, line in enumerate(open(os.path.join(SRC, fn)), 1):
            st = line.strip()
            if st.startswith("#if"):
                depth_dev.append("GF2K_DEV_VARIANTS" in st and not st.startswith("#ifndef"))
            elif st.startswith("#else") and depth_dev:
                depth_dev[-1] = False
            elif st.startswith("#endif") and depth_dev:
                depth_dev.pop()
            fm = re.match(r'^(?:static |extern "C" |inline )*[\w:<>\*&]+[\s\*&]+(\w+)\([^;]*$', line)
            if fm and not line.startswith((" ", "\t", "#", "//")):
                func = fm.group(1)
            for m in re.finditer(r'(?<![a-zA-Z_])(dev_env_int|env_int|GF2K_DEV_ENV|getenv)\("(M4RI_HIP_[A-Z0-9_]+)"(?:\s*,\s*([^)]+))?\)', line):
                name, default = m.group(2), (m.group(3) or "").strip()
                devmacro = m.group(1) in ("dev_env_int", "GF2K_DEV_ENV")
                if m.group(1) == "getenv":
                    d = re.search(r'getenv\("%s"\)\)?\s*\)?\s*:\s*([^;]+);' % name, line)
                    default = d.group(1).strip() if d else default
                e = found.setdefault(name, {"default": default, "where": [], "dev": True, "note": ""})
                if default and not e["default"]:
                    e["default"] = default
                e["where"].append("%s:%d (%s)" % (fn, ln, func))
                e["dev"] = e["dev"] and (any(depth_dev) or devmacro)
                c = line.split("//", 1)
                if len(c) == 2 and not e["note"]:
                    e["note"] = c[1].strip().strip("()")
    return found


def render():
    import io
    import contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        _print_table()
    return buf.getvalue()


def main():
    if "--write" in sys.argv:  # splice the table into INTEGRATION.md between its markers
        path = os.path.join(ROOT, "INTEGRATION.md")
        doc = open(path).read()
        a = doc.index("<!-- knob table: begin")
        a = doc.index("\n", a) + 1
        b = doc.index("<!-- knob table: end")
        open(path, "w").write(doc[:a] + render() + doc[b:])
        return
    _print_table()


def _print_table():
    k = knobs()
    for dev in (False, True):
        print("**%s**\n" % ("Read by the shipped library (`libm4ri_hip.so`)" if not dev else
                            "Development builds only (`tools/libm4ri_hip_dev.so`, `-DGF2K_DEV_VARIANTS`): fitted model constants and A/B switches; "
                            "the shipped library uses the default"))
        print("| variable | default | read in (function) | what it does |")
        print("|---|---|---|---|")
        for name in sorted(k):
            e = k[name]
            if e["dev"] != dev:
                continue
            note = DESCRIPTIONS.get(name) or (e["note"][:200] + ("..." if len(e["note"]) > 200 else ""))
            print("| `%s` | `%s` | %s | %s |" % (name, e["default"] or "unset", ", ".join("`%s`" % w for w in sorted(set(e["where"]))[:2]), note.replace("|", "\\|")))
        print()
    print("%d variables, %d of them read by the shipped library." % (len(k), sum(1 for e in k.values() if not e["dev"])))


if __name__ == "__main__":
    main()
