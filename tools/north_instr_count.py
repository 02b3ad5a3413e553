#!/usr/bin/env python3
"""Static instruction counts of xcorr_north_fft4_kernel<4>'s hot loop (one iteration = one PAIR of planes), from the device code the
build produced: what `roofline.issue_frac` / `roofline.valu_frac` of bench.py are computed from ("report both fractions").

    python tools/north_instr_count.py            -> profiles/round6_north_instr.json (+ a table on stdout)

The hot loop is the aligned fast path's pair loop.  Classes: packed fp32 VALU math (v_pk_*), other VALU, AGPR moves
(v_accvgpr_*), LDS (ds_*), global / scratch memory, scalar ALU / control, s_nop, s_waitcnt.  Every instruction costs a wave one
issue slot; a lone wave on a SIMD gets one slot per 4 clocks.  The record carries the SHA-256 of xcorr_fft.hip
so that bench.py can tell a stale count."""
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "build", "obj", "xcorr_fft.o")
SRC = os.path.join(ROOT, "hdn_amd", "csrc", "xcorr_fft.hip")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
KERNEL = "xcorr_north_fft4_kernel"


def main():
    if not os.path.exists(OBJ):
        sys.exit("build first: python -c 'import __graft_entry__ as g; g.build()'")
    subprocess.run([OBJDUMP, "--offloading", OBJ], check=True, capture_output=True)
    co = OBJ + ".0.hipv4-amdgcn-amd-amdhsa--gfx950"
    asm = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
    for f in (co, OBJ + ".0.host-x86_64-unknown-linux-gnu-"):
        if os.path.exists(f):
            os.remove(f)
    # the function's instructions: (address, mnemonic, operands)
    ins, on = [], False
    for line in asm.split("\n"):
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            on = KERNEL in m.group(2) and "Li4E" in m.group(2)
            continue
        if on:
            m = re.match(r"^\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
    if not ins:
        sys.exit("kernel not found in the disassembly")
    addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
    # backward branches -> loops; the hot loop = the longest one
    loops = []
    for i, (a, op, args) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"<[^>]*\+0x([0-9a-f]+)>", args) or re.search(r"(\d+)\s*$", args)
            tgt = None
            mm = re.search(r"<\S+\+0x([0-9a-f]+)>", line) if False else None
            # llvm-objdump prints the branch offset in dwords as a signed 16-bit immediate
            mi = re.match(r"^\s*(\d+)", args)
            if mi:
                imm = int(mi.group(1))
                if imm >= 0x8000:
                    imm -= 0x10000
                tgt = a + 4 + 4 * imm
            if tgt is not None and tgt < a and tgt in addr_index:
                loops.append((i - addr_index[tgt] + 1, addr_index[tgt], i))
    if not loops:
        sys.exit("no backward branch found")
    # the function holds two pair loops: the fast path (16-byte aligned planes, whole pairs) and the guarded one (any alignment, the odd
    # last plane) with ~2,800 more address / mask instructions.  The hot loop is the one with the highest density of packed math.
    def density(l):
        _, lo_, hi_ = l
        return sum(1 for _, op, _ in ins[lo_:hi_ + 1] if op.startswith("v_pk_")) / float(hi_ - lo_ + 1)
    loops = [l for l in loops if l[0] > 1000]
    loops.sort(key=density, reverse=True)
    n, lo, hi = loops[0]
    body = ins[lo:hi + 1]

    def cls(op):
        if op.startswith("v_pk_"):
            return "valu_packed_fp32"
        if op.startswith("v_accvgpr"):
            return "agpr_move"
        if op.startswith("v_"):
            return "valu_other"
        if op.startswith("ds_"):
            return "lds"
        if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            return "vmem"
        if op == "s_nop":
            return "s_nop"
        if op == "s_waitcnt":
            return "s_waitcnt"
        if op.startswith("s_"):
            return "scalar"
        return "other"

    counts = {}
    for _, op, _ in body:
        counts[cls(op)] = counts.get(cls(op), 0) + 1
    total = sum(counts.values())
    rec = {
        "kernel": "hdn::xcorr_north_fft4_kernel<4>", "unit": "instructions per pair of planes (one iteration of the hot loop)",
        "hot_loop": {"first_address": hex(body[0][0]), "last_address": hex(body[-1][0]), "pair_loops_found": len(loops),
                     "note": "the aligned fast path; the guarded loop (any alignment / odd last plane) is %d instructions" % (loops[1][0] if len(loops) > 1 else 0)},
        "issued_per_pair": total, "by_class": dict(sorted(counts.items(), key=lambda kv: -kv[1])),
        "valu_packed_per_pair": counts.get("valu_packed_fp32", 0),
        "flops_per_pair": counts.get("valu_packed_fp32", 0) * 64 * 4,   # a packed op = 2 fp32 lanes' FMA-class operation on 64 lanes, priced as 2 flop each
        "function_total_instructions": len(ins),
        "kernel_source": "xcorr_fft.hip", "kernel_source_sha256": hashlib.sha256(open(SRC, "rb").read()).hexdigest(),
        "how": "tools/north_instr_count.py: llvm-objdump -d of the gfx950 code object in build/obj/xcorr_fft.o; hot loop = the pair loop with the highest density of packed math (the aligned fast path)",
    }
    out = os.path.join(ROOT, "profiles", "round6_north_instr.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
