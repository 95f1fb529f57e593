"""Read per-kernel resource usage (VGPRs, spills, LDS) out of the gfx950 code objects embedded in a HIP
shared library: the clang offload bundles in .hip_fatbin, then the AMDGPU metadata note (msgpack) of each ELF."""
import struct

import msgpack

_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path, arch="gfx950"):
    blob = open(path, "rb").read()
    pos = 0
    while True:
        i = blob.find(_MAGIC, pos)
        if i < 0:
            return
        n, = struct.unpack_from("<Q", blob, i + 24)
        o = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, o)
            triple = blob[o + 24:o + 24 + tl].decode()
            o += 24 + tl
            if arch in triple and size:
                yield blob[i + off:i + off + size]
        pos = i + 24


def kernels(elf):
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    for k in range(shnum):
        sh = elf[shoff + k * shentsize: shoff + (k + 1) * shentsize]
        if struct.unpack_from("<I", sh, 4)[0] != 7:  # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", sh, 0x18)
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            name = elf[p + 12:p + 12 + namesz]
            d0 = p + 12 + ((namesz + 3) & ~3)
            desc = elf[d0:d0 + descsz]
            p = d0 + ((descsz + 3) & ~3)
            if ntype == 32 and name.startswith(b"AMDGPU"):  # NT_AMDGPU_METADATA
                for kern in msgpack.unpackb(desc, raw=False, strict_map_key=False).get("amdhsa.kernels", []):
                    yield kern


def kernel_table(path):
    """{mangled kernel name: metadata dict} for every gfx950 kernel in the library"""
    return {k[".name"]: k for co in code_objects(path) for k in kernels(co)}
