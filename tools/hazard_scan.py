#!/usr/bin/env python3
"""Scan gfx950 device assembly for software-managed DOT hazards that the compiler cannot see through inline asm.

On gfx90a+ (LLVM GCNHazardRecognizer::checkMAIVALUHazards) a v_dot* result must not be read by
  * a different VALU / VMEM / DS / FLAT instruction within 3 wait states, or
  * the same dot opcode through src A / B within 3 wait states (src C, the accumulator, is forwarded),
and must not be overwritten by a different VALU within 4 wait states.  The compiler pads with s_nop for dots it
emitted itself; a dot written as inline asm is an opaque INLINEASM to it and gets no padding.

    hipcc --offload-arch=gfx950 ... --cuda-device-only -S -o k.s k.hip ; python tools/hazard_scan.py k.s
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.startswith(".") or line.endswith(":") or line.startswith("//"):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    if not re.match(r"^(v_|s_|ds_|global_|flat_|buffer_|scratch_)", op):
        return None
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    return op, ops


def norm(op):
    return re.sub(r"_e(32|64)$", "", op)


def family(op):
    """v_dot4_i32_i8 (VOP3P) and v_dot4c_i32_i8 (VOP2) forward src C to each other on gfx950 (measured with
    tools/ubench/dot_hazard.hip; LLVM is more conservative and treats them as different opcodes)."""
    m = re.match(r"v_dot(\d+)c?_(.*)$", op)
    return f"dot{m.group(1)}_{m.group(2)}" if m else op


def scan(path):
    """-> list of (function, dot line, dot text, consumer line, consumer text, kind, dot came from inline asm)."""
    hazards = []
    func = "?"
    window = []   # [wait states since the dot, dot opcode, dst regs, line no, text, inline]
    inline = False
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            s = raw.strip()
            if s.endswith(":") and not s.startswith(".") and not s.startswith(";"):
                func = s[:-1]
                window = []   # a new function starts
            if ";;#ASMSTART" in raw:
                inline = True
            if ";;#ASMEND" in raw:
                inline = False
            p = parse(raw)
            if p is None:
                continue
            op, ops = p
            op = norm(op)
            is_valu = op.startswith("v_")
            is_mem = op.startswith(("ds_", "global_", "flat_", "buffer_", "scratch_"))
            if is_valu or is_mem:
                loads = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "global_load", "flat_load", "buffer_load", "scratch_load")) or "_rtn" in op
                writes_first = is_valu or loads
                dst = vregs(ops[0]) if (ops and writes_first) else set()
                srcs = list(enumerate(ops[1:] if writes_first else ops))
                if is_valu and re.match(r"v_dot\d+c_", op):      # two-address dot: the destination is also the accumulator (src C)
                    srcs.append((2, ops[0]))
                for (since, dop, ddst, dln, dtxt, dinl) in window:
                    for k, o in srcs:
                        if vregs(o) & ddst and not (family(op) == family(dop) and k == 2) and since < 3:
                            hazards.append((func, dln, dtxt, ln, s, f"RAW after {since} wait states", dinl))
            ws = int(ops[0], 0) + 1 if op == "s_nop" else 1
            window = [[w[0] + ws] + w[1:] for w in window if w[0] + ws < 4]
            if op.startswith("v_dot"):
                window.append([0, op, vregs(ops[0]), ln, s, inline])
    return hazards


DPP_CTRL = re.compile(r"\b(quad_perm|row_shl|row_shr|row_ror|wave_shl|wave_shr|wave_rol|wave_ror|row_mirror|row_half_mirror|row_bcast|row_newbcast|row_share|row_xmask)\b")


def scan_dpp(path):
    """A VGPR written by a VALU instruction must not be read as the DPP operand (src0) of a DPP instruction within the next
    2 wait states (LLVM GCNHazardRecognizer::checkDPPHazards).  The compiler pads for writers it emitted; a writer inside
    inline asm is invisible to it.  -> list of (function, writer line, writer text, reader line, reader text, wait states)."""
    out = []
    func = "?"
    window = []   # [wait states since the write, dst regs, line, text, writer is inline asm]
    inline = False
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            s = raw.strip()
            if s.endswith(":") and not s.startswith(".") and not s.startswith(";"):
                func, window = s[:-1], []
            if ";;#ASMSTART" in raw:
                inline = True
            if ";;#ASMEND" in raw:
                inline = False
            p = parse(raw)
            if p is None:
                continue
            op, ops = p
            op = norm(op)
            if op.startswith("v_") and DPP_CTRL.search(raw) and len(ops) > 1:
                src0 = vregs(ops[1].split()[0])
                for (since, dst, dln, dtxt, winl) in window:
                    if src0 & dst and since < 2 and (winl or inline):      # writer or reader invisible to the compiler
                        out.append((func, dln, dtxt, ln, s, since))
            ws = int(ops[0], 0) + 1 if op == "s_nop" else 1
            window = [[w[0] + ws] + w[1:] for w in window if w[0] + ws < 2]
            if op.startswith("v_") and ops and not op.startswith("v_cmp"):
                window.append([0, vregs(ops[0]), ln, s, inline])
    return out


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit("usage: hazard_scan.py <device assembly .s> ...   (hipcc --offload-arch=gfx950 ... --cuda-device-only -S -o k.s k.hip)")
    bad = 0
    for p in sys.argv[1:]:
        dz = scan_dpp(p)
        for (func, dln, dtxt, ln, s, since) in dz:
            print(f"{p}:{dln}: [VALU write read through DPP after {since} wait states, one of them inline asm] {dtxt}\n   -> {ln}: {s}\n   in {func}")
        print(f"{p}: {len(dz)} potential DPP hazards behind inline asm")
        bad += len(dz)
        hz = scan(p)
        for (func, dln, dtxt, ln, s, kind, inl) in hz:
            print(f"{p}:{dln}: [{kind}{' inline-asm' if inl else ''}] {dtxt}\n   -> {ln}: {s}\n   in {func}")
        print(f"{p}: {len(hz)} potential DOT hazards ({sum(1 for h in hz if h[6])} from inline asm)")
        bad += len(hz)
    sys.exit(1 if bad else 0)
