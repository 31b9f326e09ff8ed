"""Byte layout of a `.fmi` file (FMIndex<4,uint32_t,...>{LOOKUP_LEN=0}, fm_index.hpp:591-615) and the
one place where "byte-identical" is not defined: the reference builds `b_` with reserve + fill_n of the elements only
(container/xbit_vector.hpp:1290-1309), so the bits of the last `b_` word beyond N -- and likewise the dibits of the last
`bwt` byte beyond N -- are whatever the allocator left there.  Our builder and the oracle write zeros; a file written by
the reference may hold anything.  Every `.fmi` byte comparison in tests/ and tools/ goes through canonical()."""
import struct


def sections(buf):
    """-> dict name -> (offset of the payload, payload bytes) ; N"""
    off = 20
    out = {}
    N = None
    for name, esz in (("bwt", None), ("occ1", 16), ("occ2", 4), ("sa", 4), ("lookup", 4), ("b", None), ("b_occ", 4)):
        (count,) = struct.unpack_from("<Q", buf, off)
        off += 8
        if name == "bwt":
            N = count
            nbytes = (count + 3) // 4
        elif name == "b":
            assert count == N
            nbytes = ((count + 63) // 64) * 8
        else:
            nbytes = count * esz
        out[name] = (off, nbytes)
        off += nbytes
    assert off == len(buf), "trailing or missing bytes in .fmi"
    return out, N


def canonical(buf):
    """the file with the undefined padding bits (last bwt byte, last b_ word) forced to zero"""
    b = bytearray(buf)
    sec, N = sections(b)
    off, nbytes = sec["bwt"]
    if N % 4:
        b[off + nbytes - 1] &= (1 << (2 * (N % 4))) - 1
    off, nbytes = sec["b"]
    if N % 64:
        last = int.from_bytes(b[off + nbytes - 8:off + nbytes], "little") & ((1 << (N % 64)) - 1)
        b[off + nbytes - 8:off + nbytes] = last.to_bytes(8, "little")
    return bytes(b)


def with_garbage_padding(buf, pattern=0xA5):
    """the same index as a reference build might have written it: padding bits set to junk"""
    b = bytearray(canonical(buf))
    sec, N = sections(b)
    off, nbytes = sec["bwt"]
    if N % 4:
        b[off + nbytes - 1] |= (pattern << (2 * (N % 4))) & 0xFF
    off, nbytes = sec["b"]
    if N % 64:
        junk = int.from_bytes(bytes([pattern]) * 8, "little") & ~((1 << (N % 64)) - 1) & ((1 << 64) - 1)
        last = int.from_bytes(b[off + nbytes - 8:off + nbytes], "little") | junk
        b[off + nbytes - 8:off + nbytes] = last.to_bytes(8, "little")
    return bytes(b)
