#!/usr/bin/env python3
"""ISA lint for the gfx950 code objects of libccnet_hip: refuses instruction forms that measured WRONG on MI355X.

Rule 1 (scripts/hazard/ln_chain_min.hip, profiles/r03_hazard_*.txt): a packed-fp32 VOP3P instruction (v_pk_add_f32, v_pk_mul_f32,
v_pk_fma_f32, v_pk_mov_b32) whose LOW lane selects the HIGH dword of a 64-bit source pair (an `op_sel:[..1..]` bit) reads ZERO for
that operand in lanes 48..63 whenever an MFMA of another wave is executing on the same SIMD -- even with s_nop 7 on both sides, so no
wait state in the issuing wave cures it.  `op_sel_hi:[..0..]` (high lane takes the low dword) measured clean.  LLVM emits the bad
form whenever a scalar that is broadcast into a packed op happens to sit in the ODD register of a pair (SLP-vectorised epilogues,
LayerNorm's x - mean), i.e. it comes and goes with register allocation.  Kernels without MFMAs are checked too: they share SIMDs with
MFMA kernels on the plan's side streams.

usage: check_isa.py <object or .s files...>     exit status 1 and one line per offending instruction when a rule is broken"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("CCNET_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
PK = re.compile(r"\b(v_pk_(?:add|mul|fma)_f32|v_pk_mov_b32)\b")
OPSEL = re.compile(r"\bop_sel:\[([01,]+)\]")
LABEL = re.compile(r"^([0-9a-f]+ <)?(_Z\w+|[A-Za-z_]\w*)>?:")


def device_asm(path):
    """Disassembly (or the text itself for .s files) of the gfx950 code in `path`."""
    if path.endswith(".s"):
        with open(path) as f:
            return f.read()
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.o")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return ""   # no device code in this object
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        "--input=" + fat, "--output=" + dev], check=True)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", dev], check=True, capture_output=True, text=True).stdout


def offences(text):
    out, func = [], "?"
    for line in text.splitlines():
        s = line.strip()
        m = LABEL.match(s)
        if m and not s.startswith(("v_", "s_", "ds_", "global_", "buffer_", "scratch_", "flat_")):
            func = m.group(2)
            continue
        if PK.search(s):
            sel = OPSEL.search(s)
            if sel and "1" in sel.group(1):
                out.append((func, s.split("//")[0].strip()))
    return out


def main(paths):
    bad = 0
    for p in paths:
        for func, ins in offences(device_asm(p)):
            print("%s: %s: low lane reads the high dword of a source pair (wrong in lanes 48..63 beside MFMAs): %s" % (os.path.basename(p), func, ins))
            bad += 1
    if bad:
        print("check_isa: %d offending instruction(s); launder the broadcast operand into a real pair (see transformer_fused.hip: pair2) or "
              "build the file with -fno-slp-vectorize" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
