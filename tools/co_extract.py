#!/usr/bin/env python3
"""Extract the gfx950 code object of a hipcc-built object / shared library (.hip_fatbin: clang offload bundle, plain or
zstd-compressed 'CCOB') and print per-kernel register usage from its notes.  Development tool.
usage: co_extract.py <file.o|.so> [out.co] [name-filter]"""
import re
import struct
import subprocess
import sys

LLVM = '/opt/rocm/lib/llvm/bin/'


def bundles(blob):
    out = []
    pos = 0
    while True:
        i = blob.find(b'__CLANG_OFFLOAD_BUNDLE__', pos)
        j = blob.find(b'CCOB', pos)
        if i < 0 and j < 0:
            break
        if j >= 0 and (i < 0 or j < i):
            # compressed bundle: magic, version u16, method u16, [total size u32/u64 (v2/v3)], uncompressed size, hash u64
            ver, method = struct.unpack_from('<HH', blob, j + 4)
            if ver == 1:
                hdr = 4 + 4 + 4 + 8
                total = None
            elif ver == 2:
                total, = struct.unpack_from('<I', blob, j + 8)
                hdr = 4 + 4 + 4 + 4 + 8
            else:
                total, = struct.unpack_from('<Q', blob, j + 8)
                hdr = 4 + 4 + 8 + 8 + 8
            import zstandard  # noqa
            raise SystemExit('compressed bundle: not handled')
        n, = struct.unpack_from('<Q', blob, i + 24)
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from('<QQQ', blob, p)
            triple = blob[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            out.append((triple, blob[i + off:i + off + size]))
        pos = i + 24
    return out


def main():
    path = sys.argv[1]
    outp = sys.argv[2] if len(sys.argv) > 2 else '/tmp/extracted.co'
    filt = sys.argv[3] if len(sys.argv) > 3 else ''
    blob = open(path, 'rb').read()
    cos = [b for t, b in bundles(blob) if 'gfx950' in t and len(b) > 0]
    if not cos:
        raise SystemExit('no gfx950 code object in ' + path)
    rows = []
    for idx, co in enumerate(cos):
        p = outp if len(cos) == 1 else '%s.%d' % (outp, idx)
        open(p, 'wb').write(co)
        notes = subprocess.run([LLVM + 'llvm-readelf', '--notes', p], capture_output=True, text=True).stdout
        for k in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
            g = lambda key: int(re.search(r'\.%s:\s+(\d+)' % key, k).group(1))
            rows.append((re.search(r'\.name:\s+(\S+)', k).group(1), g('vgpr_count'), int(re.match(r':\s+(\d+)', k).group(1)),
                         g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size'),
                         g('group_segment_fixed_size')))
    names = subprocess.run(['c++filt'] + [r[0] for r in rows], capture_output=True, text=True).stdout.split('\n')
    for r, n in zip(rows, names):
        n = n.replace('void dlwpcs::', '').split('(')[0]
        if filt in n:
            print('%-100s v%3d a%3d s%3d spill v%d s%d priv %d' % (n, r[1], r[2], r[3], r[4], r[5], r[6]))


if __name__ == '__main__':
    main()
