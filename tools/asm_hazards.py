#!/usr/bin/env python3
"""Hazard audit of the library's gfx950 device code around INLINE ASM (tests/test_host_abi.py runs it).

hipcc treats an `asm` statement as one opaque instruction: it pads no hazard whose producer or consumer lies inside the string
(`cdna_hip_programming.md` section 5.7 item 2) -- that is how a `v_permlane16_swap` in inline asm read a register two `v_perm_b32`
had just written (round 4: rows 16.. of every 64 x 64 block of the large transposition were wrong for half a round).  This tool
compiles csrc/*.hip to device assembly (the flags of csrc/Makefile plus -S --cuda-device-only), walks every function linearly and
checks, for each (producer, consumer) pair of which AT LEAST ONE instruction stands between `;;#ASMSTART` and `;;#ASMEND`, the
software-inserted wait states the ISA asks for (CDNA3 ISA "manually inserted wait states", LLVM's GCNHazardRecognizer for gfx940 /
gfx950).  A wait state = one issued instruction of the wave; `s_nop N` counts N + 1.

  rule                                                                  states
  VALU writes VGPR        -> v_permlane16/32_swap reads / rewrites it      2
  VALU writes VGPR        -> DPP instruction reads it as src0              2
  VALU writes EXEC        -> DPP instruction                               5
  VALU writes VGPR        -> v_readlane / v_readfirstlane reads it         1
  VALU writes SGPR / VCC  -> v_readlane / v_writelane uses it as lane      4
  VALU writes SGPR / VCC  -> VMEM reads it (descriptor, soffset, saddr)    5
  SALU writes M0          -> ds_*_addtid, LDS-DMA (`... lds`), s_sendmsg,
                             s_movrel*                                      1
  VMEM / FLAT store of more than 64 bits -> VALU rewrites its data VGPRs   2
  VALU writes VCC         -> v_div_fmas                                     4
  SALU writes M0          -> v_readlane / v_writelane with M0 as lane      1   (not in the ISA's table: kept as for the other readers of M0)

    python tools/asm_hazards.py            # prints the findings (function, line of the .s, pair, distance); exit 1 if any
"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "m4ri-rust_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-Wno-unused-lambda-capture", "-Wno-pass-failed"]

_REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec|m0|vcc_lo|vcc_hi|exec_lo|exec_hi)\b")


def regs_of(operand):
    """Set of register names an operand text mentions: 'v12', 's4', 'vcc', 'm0', 'exec' (ranges expanded)."""
    out = set()
    for m in _REG.finditer(operand):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        elif m.group(3):
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add(m.group(3) + str(i))
        else:
            out.add(m.group(6).split("_")[0])
    return out


def split_operands(rest):
    """Operands of an instruction line (commas at bracket depth 0); modifiers after the last operand stay with it."""
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


class Ins:
    __slots__ = ("mn", "ops", "text", "line", "in_asm", "states")

    def __init__(self, mn, ops, text, line, in_asm):
        self.mn, self.ops, self.text, self.line, self.in_asm = mn, ops, text, line, in_asm
        self.states = 1
        if mn == "s_nop":
            try:
                self.states = int(ops[0], 0) + 1
            except Exception:
                self.states = 1

    # -- classification --
    def is_valu(self):
        return self.mn.startswith("v_")

    def is_salu(self):
        return self.mn.startswith("s_") and not self.mn.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_cbranch", "s_branch",
                                                                    "s_load", "s_buffer_load", "s_store", "s_dcache", "s_sleep", "s_setprio",
                                                                    "s_sendmsg", "s_memtime", "s_memrealtime", "s_icache"))

    def is_vmem(self):
        return self.mn.startswith(("buffer_", "global_", "flat_", "scratch_", "tbuffer_"))

    def is_dpp(self):
        return "_dpp" in self.mn or re.search(r"\b(quad_perm|row_shl|row_shr|row_ror|row_bcast|row_mirror|row_half_mirror|row_newbcast|wave_shl|wave_shr|wave_rol|wave_ror|row_share|row_xmask)\b", self.text) is not None

    def is_swap(self):
        return self.mn.startswith(("v_permlane16_swap", "v_permlane32_swap"))

    # -- registers written / read --
    def valu_vgpr_writes(self):
        if not self.is_valu() or not self.ops:
            return set()
        if self.is_swap() or self.mn.startswith("v_swap"):
            return {r for o in self.ops[:2] for r in regs_of(o) if r[0] == "v"}
        if self.mn.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
            return set()
        return {r for r in regs_of(self.ops[0]) if r[0] == "v"}

    def valu_sgpr_writes(self):
        """SGPRs / VCC / EXEC a VALU instruction writes: compares, readlane / readfirstlane, carry-outs, v_div_scale."""
        if not self.is_valu() or not self.ops:
            return set()
        out = set()
        if self.mn.startswith(("v_readlane", "v_readfirstlane")):
            out |= {r for r in regs_of(self.ops[0]) if r[0] != "v"}
        elif self.mn.startswith("v_cmpx"):
            out.add("exec")
            if not self.mn.endswith("_e32"):
                out |= {r for r in regs_of(self.ops[0]) if r[0] != "v"}
        elif self.mn.startswith("v_cmp"):
            out |= {r for r in regs_of(self.ops[0]) if r[0] != "v"} or {"vcc"}
        elif re.match(r"v_(add|sub|subrev)(_co|c_co)_|v_addc_|v_subb|v_div_scale|v_mad_(u|i)64", self.mn) and len(self.ops) > 1:
            out |= {r for r in regs_of(self.ops[1]) if r[0] != "v"}
        return out

    def salu_writes(self):
        if not self.is_salu() or not self.ops:
            return set()
        if self.mn.startswith(("s_cmp", "s_bitcmp", "s_setreg", "s_setpc", "s_call")):
            return set()
        return {r for r in regs_of(self.ops[0]) if r[0] != "v"}

    def reads(self, first=None):
        srcs = self.ops[1:] if first is None else self.ops[first:]
        return {r for o in srcs for r in regs_of(o)}


def parse_functions(asm_text):
    """-> {function name: [Ins]} from hipcc's device assembly."""
    funcs, cur, name, in_asm = {}, None, None, False
    for ln, line in enumerate(asm_text.splitlines(), 1):
        st = line.strip()
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith((".L", "BB")):
            name = m.group(1)
            cur = funcs.setdefault(name, [])
            in_asm = False
            continue
        if cur is None:
            continue
        if st.startswith(".Lfunc_end") or st.startswith(".size"):
            cur = None
            continue
        if st.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if st.startswith(";;#ASMEND"):
            in_asm = False
            continue
        code = st.split(";", 1)[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        parts = code.split(None, 1)
        mn = parts[0]
        if not re.match(r"^[a-z_][a-z0-9_]*$", mn):
            continue
        cur.append(Ins(mn, split_operands(parts[1]) if len(parts) > 1 else [], code, ln, in_asm))
    return funcs


def check_function(name, ins, examined=None):
    """Findings [(line, rule, producer text, consumer text, states between, states needed)]; `examined` (optional dict) counts, per
    rule, the consumers standing inside inline asm that were checked."""
    out = []

    def look_back(i, need, pred):
        """Producers within `need` wait states before instruction i for which pred(producer) is true."""
        dist, j, hits = 0, i - 1, []
        while j >= 0 and dist < need:
            p = ins[j]
            if pred(p):
                hits.append((p, dist))
            dist += p.states
            j -= 1
        return hits

    for i, c in enumerate(ins):
        checks = []  # (rule, states, predicate on the producer)
        if c.is_swap():
            rd = {r for o in c.ops[:2] for r in regs_of(o)}
            checks.append(("VALU write -> v_permlane*_swap operand", 2, lambda p, rd=rd: bool(p.valu_vgpr_writes() & rd)))
        if c.is_valu() and c.is_dpp() and len(c.ops) > 1:
            rd = regs_of(c.ops[1])
            checks.append(("VALU write -> DPP src0", 2, lambda p, rd=rd: bool(p.valu_vgpr_writes() & rd)))
            checks.append(("VALU writes EXEC -> DPP", 5, lambda p: "exec" in p.valu_sgpr_writes()))
        if c.mn.startswith(("v_readlane", "v_readfirstlane")) and len(c.ops) > 1:
            rd = {r for r in regs_of(c.ops[1]) if r[0] == "v"}
            checks.append(("VALU write -> v_readlane / v_readfirstlane source", 1, lambda p, rd=rd: bool(p.valu_vgpr_writes() & rd)))
        if c.mn.startswith(("v_readlane", "v_writelane")) and len(c.ops) > 2:
            sel = {r for r in regs_of(c.ops[2]) if r[0] != "v"}
            checks.append(("VALU writes SGPR -> lane select of v_readlane / v_writelane", 4, lambda p, sel=sel: bool(p.valu_sgpr_writes() & sel)))
        if c.is_vmem():
            rd = {r for o in c.ops for r in regs_of(o) if r[0] == "s" or r in ("vcc", "m0")}
            checks.append(("VALU writes SGPR -> VMEM reads it", 5, lambda p, rd=rd: bool(p.valu_sgpr_writes() & rd)))
        if ("addtid" in c.mn) or (c.is_vmem() and re.search(r"\blds\b", c.text)) or c.mn.startswith(("global_load_lds", "s_sendmsg", "s_movrel")):
            checks.append(("SALU writes M0 -> add-TID LDS / LDS-DMA / s_sendmsg", 1, lambda p: "m0" in p.salu_writes()))
        if c.mn.startswith(("v_readlane", "v_writelane")) and len(c.ops) > 2 and "m0" in regs_of(c.ops[2]):
            checks.append(("SALU writes M0 -> v_readlane / v_writelane lane select in M0", 1, lambda p: "m0" in p.salu_writes()))
        if c.mn.startswith("v_div_fmas"):
            checks.append(("VALU writes VCC -> v_div_fmas", 4, lambda p: "vcc" in p.valu_sgpr_writes()))
        if c.is_valu():
            wr = c.valu_vgpr_writes()
            if wr:
                def wide_store(p, wr=wr):
                    if not (p.is_vmem() and "store" in p.mn and re.search(r"(x3|x4|b96|b128)\b", p.mn)):
                        return False
                    data = regs_of(p.ops[1]) if p.mn.startswith(("global_", "flat_", "scratch_")) and len(p.ops) > 1 else regs_of(p.ops[0]) if p.ops else set()
                    return bool(data & wr)
                checks.append(("VMEM store of > 64 bits -> VALU rewrites its data", 2, wide_store))
        for rule, need, pred in checks:
            if examined is not None:  # consumers the compiler cannot have padded by itself: inside asm, or with asm inside their window
                if c.in_asm or look_back(i, need, lambda p: p.in_asm):
                    examined[rule] = examined.get(rule, 0) + 1
            for p, dist in look_back(i, need, pred):
                if p.in_asm or c.in_asm:  # the compiler pads the pairs it can see on both sides
                    out.append((c.line, rule, p.text, c.text, dist, need))
    return out


def device_asm(path):
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-S", "--cuda-device-only", "-o", "-", path], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc -S failed for %s:\n%s" % (path, r.stderr[-2000:]))
    return r.stdout


def audit(sources=None):
    """-> (findings, statistics) over csrc/*.hip."""
    findings, stats = [], {"functions": 0, "instructions": 0, "asm_instructions": 0, "examined": {}}
    for f in sources or sorted(glob.glob(os.path.join(SRC, "*.hip"))):
        funcs = parse_functions(device_asm(f))
        for name, ins in funcs.items():
            stats["functions"] += 1
            stats["instructions"] += len(ins)
            stats["asm_instructions"] += sum(1 for i in ins if i.in_asm)
            for fd in check_function(name, ins, stats["examined"]):
                findings.append((os.path.basename(f), name) + fd)
    return findings, stats


def main():
    findings, stats = audit()
    print("# %d functions, %d instructions, %d of them inside inline asm" % (stats["functions"], stats["instructions"], stats["asm_instructions"]))
    for rule, cnt in sorted(stats["examined"].items()):
        print("#   '%s': %d consumers inside inline asm or within reach of it" % (rule, cnt))
    for (src, fn, line, rule, p, c, dist, need) in findings:
        dem = subprocess.run(["c++filt", fn], capture_output=True, text=True).stdout.strip()
        print("%s: %s  (.s line %d)\n    %s: %d of %d wait states\n      producer: %s\n      consumer: %s" % (src, dem[:120], line, rule, dist, need, p, c))
    print("# %d findings" % len(findings))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
