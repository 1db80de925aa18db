#!/usr/bin/env python3
"""Give the extend_w64 kernels the accumulation registers their assembly text uses - AT ASSEMBLY LEVEL.

extend_w64.hip keeps O^T and the Q fragments in a[0:191] (the persistent form: a[0:235]), named only inside
`asm volatile` text.  For the compiler to be UNABLE to put anything of its own there (it does, as soon as architectural
registers run short, and an AGPR named in any asm constraint or clobber makes all of them allocatable), the source never
mentions an accumulation register to it and is built for a 256-register budget (__launch_bounds__(256, 2)): hipcc then
reserves every AGPR, and the kernel descriptor and metadata it writes allocate none.

Until round 4 a script raised GRANULATED_WORKITEM_VGPR_COUNT in the descriptors of the LINKED library (byte surgery; the
.amdgpu_metadata note kept saying agpr_count 0).  Now the build (scratchpad_amd/build.py) compiles this one file in
stages - device assembly (`hipcc --cuda-device-only -S`), THIS rewrite, `clang -x assembler`, lld, the offload bundle,
the host object with `-fcuda-include-gpubinary` - and the rewrite edits the two places the compiler states the
allocation, for the kernels named in KERNELS only:

  .amdhsa_next_free_vgpr  N   ->  ACCUM_OFFSET + 256     (gfx90a+: the unified total = accum offset + AGPRs; the
                                                           assembler derives the descriptor's granule count from it
                                                           and validates it against the register file)
  .amdgpu_metadata:  .agpr_count 0 -> 256,  .vgpr_count N -> ACCUM_OFFSET + 256

so descriptor and note agree and nothing is patched after linking.  The kernels' text addresses a[i] on the assumption
that the compiler keeps its own values in v[0:ACCUM_OFFSET): a toolchain that lays the register file out differently
(another ACCUM_OFFSET for the same source) fails the build here (EXPECTED), to be looked at by a person.

  python tools/w64_asm.py rewrite IN.s OUT.s      # the build step
  python tools/w64_asm.py check LIB.so|OBJ.hsaco   # descriptor == metadata note == EXPECTED, for every w64 kernel
"""
import re
import struct
import subprocess
import sys

W64_AGPRS = 256     # a[0:191] O^T and Q; a[192:235]: the persistent form (the next item's kv slots and half of its Q rows)
KERNELS = ("extend_w64_kernel", "extend_w64p_kernel")
# kernel -> ACCUM_OFFSET with hipcc of ROCm 7.2.0 (build.py records the compiler version next to the library)
EXPECTED = {"extend_w64_kernel": 228, "extend_w64p_kernel": 248}
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_of(symbol: str):
    for k in KERNELS:
        if "2sp%d%sI" % (len(k), k) in symbol:
            return k
    return None


def rewrite(text: str) -> str:
    out, seen = [], {}
    lines = text.split("\n")
    i = 0
    # ---- the .amdhsa_kernel blocks
    cur = None
    block_start = None
    for idx, line in enumerate(lines):
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", line)
        if m:
            cur, block_start = (m.group(1) if kernel_of(m.group(1)) else None), idx
        elif re.match(r"\s*\.end_amdhsa_kernel", line):
            if cur:
                blk = range(block_start, idx)
                acc = [j for j in blk if re.match(r"\s*\.amdhsa_accum_offset\s+\d+", lines[j])]
                nfv = [j for j in blk if re.match(r"\s*\.amdhsa_next_free_vgpr\s+\d+", lines[j])]
                assert len(acc) == 1 and len(nfv) == 1, f"{cur}: descriptor directives not found"
                accum = int(lines[acc[0]].split()[-1])
                have = int(lines[nfv[0]].split()[-1])
                want = EXPECTED[kernel_of(cur)]
                if accum != want:
                    raise SystemExit(f"{cur}: ACCUM_OFFSET {accum}, expected {want}: the compiler lays this kernel's registers "
                                     "out differently from the toolchain extend_w64.hip was written against - inspect before use")
                if have > accum:
                    raise SystemExit(f"{cur}: the compiler itself allocates accumulation registers (next_free_vgpr {have} > "
                                     f"accum_offset {accum}): the asm text's a[0:{W64_AGPRS - 1}] would collide with them")
                lines[nfv[0]] = re.sub(r"\d+\s*$", str(accum + W64_AGPRS), lines[nfv[0]])
                seen[cur] = accum
            cur = None
    # ---- the metadata note (YAML between .amdgpu_metadata and .end_amdgpu_metadata): one entry per kernel, "  - " opens it
    try:
        a = next(j for j, l in enumerate(lines) if l.strip() == ".amdgpu_metadata")
        b = next(j for j, l in enumerate(lines) if l.strip() == ".end_amdgpu_metadata")
    except StopIteration:
        raise SystemExit("no .amdgpu_metadata block in the assembly")
    starts = [j for j in range(a, b) if lines[j].startswith("  - .") and not lines[j].startswith("      ")]
    starts = [j for j in starts if re.match(r"  - \.\w+:", lines[j])]
    noted = set()
    for n, s in enumerate(starts):
        e = starts[n + 1] if n + 1 < len(starts) else b
        name = next((re.match(r"\s+\.name:\s+(\S+)", lines[j]).group(1) for j in range(s, e)
                     if re.match(r"\s+\.name:\s+\S+", lines[j]) and not lines[j].startswith("      ")), None)
        if name not in seen:
            continue
        for j in range(s, e):
            if re.match(r"\s+(- )?\.agpr_count:\s+\d+", lines[j]) and not lines[j].startswith("      "):
                lines[j] = re.sub(r"\d+\s*$", str(W64_AGPRS), lines[j])
            elif re.match(r"\s+(- )?\.vgpr_count:\s+\d+", lines[j]) and not lines[j].startswith("      "):
                lines[j] = re.sub(r"\d+\s*$", str(seen[name] + W64_AGPRS), lines[j])
        noted.add(name)
    if len(seen) != 2 * len(KERNELS) or noted != set(seen):
        raise SystemExit(f"expected {2 * len(KERNELS)} w64 kernels with descriptor and note, found descriptors {sorted(seen)}, notes {sorted(noted)}")
    return "\n".join(lines)


# ------------------------------------------------------------------------------------------------ check (reads ELF)
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_elves(data):
    """(offset, size) of the gfx code objects inside a host library / fat object (offload bundles), or the file itself"""
    if data[:4] == b"\x7fELF" and struct.unpack_from("<H", data, 18)[0] == 224:      # EM_AMDGPU: a bare code object
        yield 0, len(data)
        return
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl]
            off += tl
            if triple.startswith(b"hip") and size:
                yield i + o, size
        pos = i + len(MAGIC)


def descriptors(data, base):
    """(symbol, file offset of the 64-byte kernel descriptor) of the w64 kernels of the ELF at `base`"""
    assert data[base:base + 4] == b"\x7fELF" and data[base + 4] == 2
    shoff, = struct.unpack_from("<Q", data, base + 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", data, base + 0x3A)
    sections = [struct.unpack_from("<IIQQQQIIQQ", data, base + shoff + k * shentsize) for k in range(shnum)]
    for sec in sections:
        if sec[1] not in (2, 11):      # SHT_SYMTAB / SHT_DYNSYM
            continue
        strtab = sections[sec[6]]
        for k in range(sec[5] // 24):
            name_off, _info, _other, shndx, value, _size = struct.unpack_from("<IBBHQQ", data, base + sec[4] + 24 * k)
            end = data.index(b"\0", base + strtab[4] + name_off)
            name = data[base + strtab[4] + name_off:end].decode()
            if name.endswith(".kd") and kernel_of(name) and 0 < shndx < shnum:
                s = sections[shndx]
                yield name[:-3], base + s[4] + (value - s[3])


def notes(path_of_code_object):
    """kernel symbol -> (vgpr_count, agpr_count) from the code object's .amdgpu_metadata note (llvm-readelf --notes)"""
    txt = subprocess.run([READELF, "--notes", path_of_code_object], check=True, capture_output=True, text=True).stdout
    out, cur = {}, {}
    for line in txt.splitlines():
        m = re.match(r"\s+(?:- )?\.(agpr_count|vgpr_count|name):\s+(\S+)\s*$", line)
        if not m or line.startswith("        "):       # (deeper indentation: an argument's .name)
            if re.match(r"\s+- \.", line) and not line.startswith("      "):
                cur = {}
            continue
        if line.lstrip().startswith("- "):
            cur = {}
        cur[m.group(1)] = m.group(2)
        if {"agpr_count", "vgpr_count", "name"} <= set(cur) and kernel_of(cur["name"]):
            out[cur["name"]] = (int(cur["vgpr_count"]), int(cur["agpr_count"]))
    return out


def host_flag_offset(data):
    """file offset of the int `sp_w64_descriptor_patched` in a host ELF (the shared library), None if it has none"""
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return None
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", data, 0x3A)
    sections = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + k * shentsize) for k in range(shnum)]
    for sec in sections:
        if sec[1] not in (2, 11):
            continue
        strtab = sections[sec[6]]
        for k in range(sec[5] // 24):
            name_off, _info, _other, shndx, value, _size = struct.unpack_from("<IBBHQQ", data, sec[4] + 24 * k)
            end = data.index(b"\0", strtab[4] + name_off)
            if data[strtab[4] + name_off:end] == b"sp_w64_descriptor_patched" and 0 < shndx < shnum:
                s = sections[shndx]
                return None if s[1] == 8 else s[4] + (value - s[3])       # (.bss: zero, i.e. not set)
    return None


def check(path: str, verbose: bool = True) -> bool:
    import os
    import tempfile
    data = open(path, "rb").read()
    ok, found = True, 0
    for base, size in device_elves(data):
        descs = dict(descriptors(data, base))
        if not descs:
            continue
        size = min(size, len(data) - base)
        with tempfile.NamedTemporaryFile(suffix=".hsaco", delete=False) as f:
            f.write(data[base:base + size])
        try:
            note = notes(f.name)
        finally:
            os.unlink(f.name)
        for name, off in sorted(descs.items()):
            found += 1
            rsrc3, rsrc1 = struct.unpack_from("<II", data, off + 44)
            accum = ((rsrc3 & 0x3F) + 1) * 4
            regs = ((rsrc1 & 0x3F) + 1) * 8
            want_accum = EXPECTED[kernel_of(name)]
            total = want_accum + W64_AGPRS
            nv, na = note.get(name, (None, None))
            good = (accum == want_accum and regs == (total + 7) // 8 * 8 and nv == total and na == W64_AGPRS)
            ok &= good
            if verbose:
                print(f"{name}: descriptor accum_offset {accum}, {regs} registers; note vgpr_count {nv}, agpr_count {na}; "
                      f"expected accum_offset {want_accum}, {total} registers of which {W64_AGPRS} accumulation -> "
                      f"{'ok' if good else 'MISMATCH'}")
    if found != 2 * len(KERNELS):
        if verbose:
            print(f"{path}: {found} extend_w64 kernel descriptors found, expected {2 * len(KERNELS)}")
        ok = False
    if data[:4] == b"\x7fELF" and struct.unpack_from("<H", data, 18)[0] != 224:      # a host library: its launch flag
        off = host_flag_offset(data)
        flag = None if off is None else struct.unpack_from("<i", data, off)[0]
        if verbose:
            print(f"sp_w64_descriptor_patched = {flag}")
        ok &= flag == 1
    return ok


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "rewrite":
        open(sys.argv[3], "w").write(rewrite(open(sys.argv[2]).read()))
    elif len(sys.argv) >= 3 and sys.argv[1] == "check":
        sys.exit(0 if check(sys.argv[2]) else 1)
    else:
        print(__doc__)
        sys.exit(2)


if __name__ == "__main__":
    main()
