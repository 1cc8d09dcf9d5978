#!/usr/bin/env python3
"""Kernel resource usage read from the gfx950 code object inside plonky2_amd/libp2hot.so (tooling + a build test).

hipcc embeds a clang offload bundle in the `.hip_fatbin` section; the gfx950 entry is an ELF whose NT_AMDGPU_METADATA
note (msgpack) lists, per kernel, VGPR / SGPR counts, spills, scratch and LDS bytes.  A compiler bump that turns the
hand-scheduled kernels' register budget into spills shows up here, without a GPU.

    python tools/codeobj.py [substring ...]      # table of the matching kernels"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "plonky2_amd", "libp2hot.so")


def gfx950_elf(path=SO):
    b = open(path, "rb").read()
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if i < 0:
        raise RuntimeError("no clang offload bundle in %s" % path)
    n = struct.unpack_from("<Q", b, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, tl = struct.unpack_from("<QQQ", b, off)
        off += 24
        triple = b[off:off + tl].decode()
        off += tl
        if "gfx950" in triple:
            return b[i + o:i + o + s]
    raise RuntimeError("no gfx950 entry in the bundle")


def kernel_metadata(path=SO):
    """{kernel symbol: metadata dict} from the code object's AMDGPU note"""
    import msgpack
    e = gfx950_elf(path)
    assert e[:4] == b"\x7fELF" and e[4] == 2
    shoff, = struct.unpack_from("<Q", e, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", e, 0x3A)
    for k in range(shnum):
        sh = shoff + k * shentsize
        sh_type, = struct.unpack_from("<I", e, sh + 4)
        sh_offset, sh_size = struct.unpack_from("<QQ", e, sh + 0x18)
        if sh_type != 7:  # SHT_NOTE
            continue
        p, end = sh_offset, sh_offset + sh_size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", e, p)
            p += 12
            name = e[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = e[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b"AMDGPU" and ntype == 32:
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                return {k_[".name"]: k_ for k_ in md["amdhsa.kernels"]}
    raise RuntimeError("no AMDGPU metadata note")


def waves_per_simd(vgprs, threads):
    """occupancy bound from the unified 512-entry VGPR file per lane-slot (granule 8), whole workgroups of `threads`"""
    g = (vgprs + 7) // 8 * 8
    by_regs = min(8, 512 // max(g, 8))
    waves_per_wg_per_simd = max(1, threads // 64 // 4)
    return by_regs // waves_per_wg_per_simd * waves_per_wg_per_simd


if __name__ == "__main__":
    md = kernel_metadata()
    subs = sys.argv[1:]
    for name in sorted(md):
        if subs and not any(s in name for s in subs):
            continue
        k = md[name]
        print("%-110s vgpr %3d spill %3d sgpr %3d scratch %5d lds %6d wg %4d" % (
            name[:110], k[".vgpr_count"], k.get(".vgpr_spill_count", 0), k[".sgpr_count"], k[".private_segment_fixed_size"],
            k[".group_segment_fixed_size"], k[".max_flat_workgroup_size"]))
