"""LAST test of the GPU suite (the file name sorts behind every other test file): every kernel that libm4ri_hip.so CONTAINS has been
LAUNCHED by the suite -- in this process (gf2_kernel_census) or in one of its child processes (M4RI_HIP_KERNEL_CENSUS_FILE, set by
conftest.py).  Round 4: the 512-tile transposition ran wrong for half a round because its size threshold had moved above every test;
`tools/kernel_coverage.py` found 28 such kernels but needs a rocprofv3 run by hand.  This one needs nothing but the shared object:
the kernels in it are read from the device code objects embedded in its .hip_fatbin section (symbols `<name>.kd`)."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_in_shared_object(path):
    """Mangled names of the kernels in every gfx950 code object of the offload bundles inside `path` (pure Python: the clang offload
    bundle header, then the ELF64 symbol tables)."""
    blob = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    names = set()
    for m in re.finditer(re.escape(magic), blob):
        base = m.start()
        (nent,) = struct.unpack_from("<Q", blob, base + len(magic))
        pos = base + len(magic) + 8
        for _ in range(nent):
            off, size, idlen = struct.unpack_from("<QQQ", blob, pos)
            ident = blob[pos + 24:pos + 24 + idlen].decode()
            pos += 24 + idlen
            if "gfx950" not in ident or size == 0:
                continue
            elf = blob[base + off:base + off + size]
            assert elf[:4] == b"\x7fELF" and elf[4] == 2, ident
            shoff, = struct.unpack_from("<Q", elf, 0x28)
            shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
            secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
            for (_, stype, _, _, soff, ssize, link, _, _, entsize) in secs:
                if stype != 2:  # SHT_SYMTAB
                    continue
                stroff = secs[link][4]
                for j in range(ssize // entsize):
                    st_name, st_info, _, _, _, _ = struct.unpack_from("<IBBHQQ", elf, soff + j * entsize)
                    end = elf.index(b"\0", stroff + st_name)
                    nm = elf[stroff + st_name:end].decode()
                    if nm.endswith(".kd"):
                        names.add(nm[:-3])
    return names


def _parse(text, into):
    for ln in text.splitlines():
        parts = ln.split(None, 1)
        if len(parts) == 2 and parts[0].isdigit():
            into[parts[1].strip()] = into.get(parts[1].strip(), 0) + int(parts[0])


def test_bundle_parser_finds_the_kernels(built):
    """CPU part: the parser sees the kernels of the library (no GPU needed) -- every kernel family by name."""
    import m4ri_rust_amd as pkg
    ks = kernels_in_shared_object(pkg._lib.LIB_PATH)
    dem = subprocess.run(["c++filt"], input="\n".join(sorted(ks)), capture_output=True, text=True).stdout
    assert len(ks) >= 100, len(ks)
    for family in ("gf2_m4rm_kernel_v8<8, 2, 1, 2, 1>", "gf2_lpn256_kernel", "gf2_lpnvec_kernel", "gf2_transpose512_kernel", "gf2_strassen_split3_kernel",
                   "gf2_streamk_reduce_kernel"):
        assert family in dem, family


@pytest.mark.gpu
def test_every_kernel_of_the_library_was_launched(built):
    import ctypes
    import m4ri_rust_amd as pkg
    from m4ri_rust_amd import device
    device.require_gpu()
    L = pkg._lib.lib()
    want = kernels_in_shared_object(pkg._lib.LIB_PATH)
    counts = {}
    need = L.gf2_kernel_census(None, 0)
    buf = ctypes.create_string_buffer(need + 1)
    L.gf2_kernel_census(buf, need + 1)
    _parse(buf.value.decode(), counts)
    in_process = len(counts)
    path = os.environ.get("M4RI_HIP_KERNEL_CENSUS_FILE")
    if path and os.path.exists(path):
        _parse(open(path).read(), counts)
    # only meaningful at the end of the WHOLE suite: a run of a few selected tests (-k, a single file) is not a census
    if in_process < 40:
        pytest.skip("only %d kernels launched in this process: not a run of the whole GPU suite" % in_process)
    missing = sorted(k for k in want if counts.get(k, 0) == 0)
    if missing:
        dem = subprocess.run(["c++filt"], input="\n".join(missing), capture_output=True, text=True).stdout
        pytest.fail("%d of %d kernels of libm4ri_hip.so were never launched by the GPU suite:\n%s" % (len(missing), len(want), dem))
    unknown = sorted(k for k in counts if k not in want and k != "?")
    assert not unknown, unknown  # a launch the library's own code objects do not explain
    print("kernel census: %d kernels in the library, all launched (%d launches in this process and its children)" % (len(want), sum(counts.values())))
