#!/usr/bin/env python3
"""VGPR / SGPR / spill / LDS / scratch figures of every kernel in the built library, read from the metadata notes of every gfx950 code object
in its .hip_fatbin section (one offload bundle per translation unit). With --scratch it also disassembles each kernel and counts its
scratch_load / scratch_store instructions, split at the label of the packed kernels' tail (the first s_cbranch that jumps over the inlined
general path is not recoverable from the ISA; the split used here is the source-level marker the kernels emit: `s_nop 7 ; s_nop 6` pairs are
not used — instead the tail is a __noinline__-free region AFTER the last global qdot store of the packed path, see count_scratch()).
usage: tools/kernel_resources.py [--scratch] [path/to/libwbc_hip.so]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(so):
    """-> list of gfx950 ELF images (bytes) found in the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, fat])
        blob = open(fat, "rb").read()
    out, pos = [], blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + 1)
    return out


def notes(elf):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        return subprocess.check_output([LLVM + "/llvm-readelf", "--notes", f.name], text=True)


def disasm(elf):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf)
        f.flush()
        return subprocess.check_output([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", f.name], text=True)


def demangle(names):
    return subprocess.check_output(["c++filt"] + names, text=True).splitlines()


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    so = args[0] if args else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mech5845m-wbc-for-legged-manipulator_amd", "csrc", "build", "libwbc_hip.so")
    want_scratch = "--scratch" in sys.argv
    rows = []
    for elf in code_objects(so):
        txt = notes(elf)
        scr = {}
        if want_scratch:
            cur = None
            for line in disasm(elf).splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    scr[cur] = [0, 0]
                elif cur and "scratch_load" in line:
                    scr[cur][0] += 1
                elif cur and "scratch_store" in line:
                    scr[cur][1] += 1
        for blk in re.split(r"\n\s*- \.agpr_count", txt)[1:]:
            def f(k):
                m = re.search(r"\." + k + r":\s*(\S+)", blk)
                return m.group(1) if m else "?"
            name = f("name")
            rows.append((name, f("vgpr_count"), f("vgpr_spill_count"), f("sgpr_count"), f("sgpr_spill_count"), f("group_segment_fixed_size"),
                         f("private_segment_fixed_size"), scr.get(name)))
    nice = demangle([r[0] for r in rows])
    for r, nm in sorted(zip(rows, nice), key=lambda t: t[1]):
        nm = re.sub(r"\(.*", "", nm.replace("void wbc::", ""))
        extra = "" if r[7] is None else "  scratch_load %3d scratch_store %3d" % tuple(r[7])
        print("%-52s vgpr %3s (spill %3s)  sgpr %3s (spill %3s)  lds %6s  scratch %5s%s" % (nm[:52], r[1], r[2], r[3], r[4], r[5], r[6], extra))


if __name__ == "__main__":
    main()
