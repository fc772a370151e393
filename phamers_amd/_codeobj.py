"""Reads the register / scratch figures of the kernels inside libphamers_hip.so: the gfx950 code objects of the library's
offload bundles, their AMDGPU metadata notes (msgpack).  Used by tests/test_build_resources.py and tools/kernel_resources.py;
nothing on the product path imports it."""
import re
import struct

_BUNDLE = b"__CLANG_OFFLOAD_BUNDLE__"


def _bundle_entries(data, at):
    n, = struct.unpack_from("<Q", data, at + len(_BUNDLE))
    p = at + len(_BUNDLE) + 8
    for _ in range(n):
        off, size, tlen = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tlen].decode()
        p += 24 + tlen
        yield triple, data[at + off:at + off + size]


def _elf_notes(elf):
    if elf[:4] != b"\x7fELF":
        return
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:    # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz].rstrip(b"\0")
            d0 = p + 12 + ((namesz + 3) & ~3)
            yield name, ntype, elf[d0:d0 + descsz]
            p = d0 + ((descsz + 3) & ~3)


def kernel_resources(path):
    """{demangled-ish kernel symbol: {"vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "scratch_bytes", "lds_bytes"}} of
    every gfx950 kernel in the library."""
    import msgpack
    data = open(path, "rb").read()
    out = {}
    for m in re.finditer(re.escape(_BUNDLE), data):
        for triple, blob in _bundle_entries(data, m.start()):
            if "gfx950" not in triple:
                continue
            for name, ntype, desc in _elf_notes(blob):
                if name != b"AMDGPU" or ntype != 32:
                    continue
                meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    out[k[".name"]] = {"vgpr_count": k.get(".vgpr_count"), "vgpr_spill_count": k.get(".vgpr_spill_count", 0),
                                       "sgpr_spill_count": k.get(".sgpr_spill_count", 0),
                                       "scratch_bytes": k.get(".private_segment_fixed_size", 0),
                                       "lds_bytes": k.get(".group_segment_fixed_size", 0)}
    return out
