#!/usr/bin/env python3
"""Give the extend_w64 kernels the accumulation registers their assembly text uses.

extend_w64.hip keeps O^T and the Q fragments in a[0:191], named only inside `asm volatile` text.  For the compiler to
be UNABLE to put anything of its own there (it does, as soon as architectural registers run short, and an AGPR named in
any asm constraint or clobber makes all of them allocatable), the source never mentions an accumulation register to it
and is built for a 256-register budget (__launch_bounds__(256, 2)): hipcc then reserves every AGPR and the kernel
descriptor it writes allocates none.  This script, run on the built library (scratchpad_amd/build.py,
tools/build_w64_variant.sh), raises GRANULATED_WORKITEM_VGPR_COUNT in compute_pgm_rsrc1 of those kernels'
descriptors to ACCUM_OFFSET + 256 registers - the unified register file of gfx90a+ places a[i] at register
ACCUM_OFFSET + i of the wave's allocation (amdhsa kernel descriptor, LLVM AMDGPUUsage 'Kernel Descriptor').
Once every descriptor is in place the script sets the library's host-side flag `sp_w64_descriptor_patched` to 1;
extend_w64.hip refuses to launch its kernels while that flag is 0 (a library linked without this step).
  python tools/patch_w64_descriptor.py LIB.so [--check] [--expect]
--check: change nothing, exit 1 unless every descriptor and the flag are in place.
--expect: also exit 1 unless each kernel's ACCUM_OFFSET and patched granule count are the ones this source tree was
written for (EXPECTED below): the kernels' assembly text addresses a[0:235] on the assumption that the compiler keeps
its own values in v[0:ACCUM_OFFSET) - a toolchain that lays the register file out differently (another ACCUM_OFFSET
for the same source) has to be looked at by a person, at build time, not found by a failing parity test on a GPU.
"""
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
W64_AGPRS = 256     # a[0:191] O^T and Q; a[192:235]: the persistent form (the next item's kv slots and half of its Q rows)
KERNELS = (b"extend_w64_kernel", b"extend_w64p_kernel")
# kernel -> (ACCUM_OFFSET, register granules after the patch) with hipcc of ROCm 7.2.0 (build.py records the version)
EXPECTED = {"extend_w64_kernel": (228, 61), "extend_w64p_kernel": (248, 63)}


def device_elves(data):
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
    """(name, file offset of the 64-byte kernel descriptor) of the w64 kernels of the ELF at `base`"""
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
            name = data[base + strtab[4] + name_off:end]
            if name.endswith(b".kd") and any(kn in name for kn in KERNELS) and 0 < shndx < shnum:
                s = sections[shndx]
                yield name.decode(), base + s[4] + (value - s[3])


def host_flag_offset(data):
    """file offset of the int `sp_w64_descriptor_patched` in the host ELF (the shared library itself)"""
    assert data[:4] == b"\x7fELF" and data[4] == 2
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
                assert s[1] != 8, "the flag must live in .data (initialised), not .bss"
                return s[4] + (value - s[3])
    return None


def main():
    path = sys.argv[1]
    check = "--check" in sys.argv
    expect = "--expect" in sys.argv
    data = bytearray(open(path, "rb").read())
    seen, changed = {}, 0
    for base, _size in device_elves(bytes(data)):
        for name, off in descriptors(bytes(data), base):
            rsrc3, rsrc1 = struct.unpack_from("<II", data, off + 44)
            accum = ((rsrc3 & 0x3F) + 1) * 4
            want = (accum + W64_AGPRS + 7) // 8 - 1
            have = rsrc1 & 0x3F
            state = "ok" if have >= want else "NOT PATCHED"
            if have < want and not check:
                struct.pack_into("<I", data, off + 48, (rsrc1 & ~0x3F) | want)
                changed += 1
                state = "patched"
            seen[name] = (accum, have, want, state)
    if not seen:
        print(f"{path}: no extend_w64 kernel descriptor found")
        sys.exit(1)
    ok = True
    for name, (accum, have, want, state) in sorted(seen.items()):
        ok &= state != "NOT PATCHED"
        print(f"{name}: accum_offset {accum}, register granules {have + 1} -> {max(have, want) + 1} ({(max(have, want) + 1) * 8} registers) {state}")
    flag = host_flag_offset(bytes(data))
    if flag is None:
        print(f"{path}: host flag sp_w64_descriptor_patched not found")
        sys.exit(1)
    have_flag, = struct.unpack_from("<i", data, flag)
    if ok and not check and have_flag != 1:
        struct.pack_into("<i", data, flag, 1)
        changed += 1
        have_flag = 1
    print(f"sp_w64_descriptor_patched = {have_flag}")
    ok &= have_flag == 1
    if expect:
        for name, (accum, have, want, _state) in sorted(seen.items()):
            kernel = next(k for k in EXPECTED if ("2sp%d%sI" % (len(k), k)) in name)
            if (accum, max(have, want) + 1) != EXPECTED[kernel]:
                print(f"{name}: accum_offset / granules {(accum, max(have, want) + 1)} differ from the expected "
                      f"{EXPECTED[kernel]}: the compiler lays this kernel's registers out differently from the toolchain "
                      "extend_w64.hip was written against - inspect before use")
                ok = False
    if changed:
        open(path, "wb").write(data)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
