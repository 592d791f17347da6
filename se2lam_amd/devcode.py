"""What device code does libse2gpu.so carry?  -> {kernel name: sha of the code object it lives in}

A counter capture (profiles/pmc_traffic.json) is valid for the kernels it ran, not for a source file: a host-only edit of
csrc/ba.hip, or a change to another translation unit, must not make it stale.  The stamp is therefore
taken from the library itself: every translation unit's clang offload bundle sits in the .so's `.hip_fatbin` data; the gfx950
entry of a bundle is an ELF code object; its loadable contents (.text, .rodata, .data - the instructions, kernel descriptors
and constants, NOT the symbol names, so a changed compilation-unit id does not show) are hashed, and every kernel whose
descriptor (`<name>.kd`) the object defines is mapped to that hash.  Pure Python, no tools needed on the GPU box."""
from __future__ import annotations

import hashlib
import os
import re
import struct
import subprocess

from . import capi

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _bundles(blob: bytes):
    for m in re.finditer(MAGIC, blob):
        base = m.start()
        (n,) = struct.unpack_from("<Q", blob, base + len(MAGIC))
        p = base + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode(errors="replace")
            p += 24 + tlen
            if "gfx950" in triple and size:
                yield blob[base + off:base + off + size]


def _elf_sections(elf: bytes):
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "not a 64-bit ELF code object"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize)
        secs.append(dict(name=name, type=typ, off=off, size=size, link=link, entsize=entsize))
    strtab = secs[shstrndx]
    for s in secs:
        e = elf.index(b"\0", strtab["off"] + s["name"])
        s["name"] = elf[strtab["off"] + s["name"]:e].decode()
    return secs


def _demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, timeout=20)
        if r.returncode == 0:
            return r.stdout.split("\n")[:len(names)]
    except (OSError, subprocess.SubprocessError):
        pass
    return list(names)


def _base(demangled: str) -> str:
    """`void (anonymous namespace)::k_cell_retain<false>(Geom, ...)` -> `k_cell_retain` (the name the profiles use)"""
    s = re.sub(r"\(anonymous namespace\)::", "", demangled).split("(")[0].replace("void ", "")
    s = re.sub(r"<.*", "", s)
    return s.split("::")[-1].strip()


def kernel_code_hashes(lib_path: str | None = None) -> dict:
    path = lib_path or capi.LIB_PATH
    blob = open(path, "rb").read()
    out = {}
    for elf in _bundles(blob):
        secs = _elf_sections(elf)
        h = hashlib.sha256()
        for s in secs:
            if s["name"] in (".text", ".rodata", ".data") and s["type"] != 8:   # 8 = NOBITS
                h.update(s["name"].encode())
                h.update(elf[s["off"]:s["off"] + s["size"]])
        sha = h.hexdigest()[:16]
        sym = next((s for s in secs if s["name"] == ".symtab"), None)
        if sym is None:
            continue
        st = secs[sym["link"]]
        kds = []
        for i in range(sym["size"] // 24):
            n, = struct.unpack_from("<I", elf, sym["off"] + 24 * i)
            e = elf.index(b"\0", st["off"] + n)
            nm = elf[st["off"] + n:e].decode(errors="replace")
            if nm.endswith(".kd"):
                kds.append(nm[:-3])
        for d in _demangle(kds):
            out[_base(d)] = sha
    return out


if __name__ == "__main__":
    for k, v in sorted(kernel_code_hashes().items()):
        print(v, k)
