"""A reader of BGZF files and tabix indexes written from the specifications alone (SAM/BAM specification section 4.1 for BGZF,
section 5.3 for the binning scheme; the tabix format description for the TBI layout) -- the checker of palace_amd/host/depthgz.hpp.
Independent of the writer: nothing here is shared with it.  Test infrastructure."""
import struct
import zlib

EOF_MEMBER = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


def bgzf_members(data: bytes):
    """-> [(file offset, text bytes)] of every member, each checked: gzip magic, FEXTRA with the BC subfield, BSIZE, CRC32, ISIZE"""
    out, at = [], 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04", f"member at {at}: not a gzip member with FEXTRA"
        xlen = struct.unpack_from("<H", data, at + 10)[0]
        extra, bsize, k = data[at + 12:at + 12 + xlen], None, 0
        while k < len(extra):
            si1, si2, slen = extra[k], extra[k + 1], struct.unpack_from("<H", extra, k + 2)[0]
            if (si1, si2) == (66, 67):
                assert slen == 2
                bsize = struct.unpack_from("<H", extra, k + 4)[0]
            k += 4 + slen
        assert bsize is not None, "no BC subfield"
        total = bsize + 1
        cdata = data[at + 12 + xlen:at + total - 8]
        text = zlib.decompress(cdata, -15)
        crc, isize = struct.unpack_from("<II", data, at + total - 8)
        assert crc == zlib.crc32(text) and isize == len(text) and len(text) <= 0x10000
        out.append((at, text))
        at += total
    assert at == len(data)
    return out


def read_tbi(path):
    raw = open(path, "rb").read()
    mem = bgzf_members(raw)
    assert raw[-28:] == EOF_MEMBER
    t = b"".join(x for _, x in mem)
    assert t[:4] == b"TBI\x01"
    n_ref, fmt, col_seq, col_beg, col_end, meta, skip, l_nm = struct.unpack_from("<8i", t, 4)
    names = t[36:36 + l_nm].split(b"\0")[:-1]
    assert len(names) == n_ref
    at = 36 + l_nm
    refs = []
    for _ in range(n_ref):
        n_bin = struct.unpack_from("<i", t, at)[0]; at += 4
        bins = {}
        for _ in range(n_bin):
            b, n_chunk = struct.unpack_from("<Ii", t, at); at += 8
            bins[b] = [struct.unpack_from("<QQ", t, at + 16 * k) for k in range(n_chunk)]
            at += 16 * n_chunk
        n_intv = struct.unpack_from("<i", t, at)[0]; at += 4
        ioff = list(struct.unpack_from(f"<{n_intv}Q", t, at)); at += 8 * n_intv
        refs.append(dict(bins=bins, ioff=ioff))
    n_no_coor = struct.unpack_from("<Q", t, at)[0] if at + 8 <= len(t) else None
    assert at + (8 if n_no_coor is not None else 0) == len(t)
    return dict(format=fmt, col_seq=col_seq, col_beg=col_beg, col_end=col_end, meta=meta, skip=skip, names=names, refs=refs, n_no_coor=n_no_coor)


def reg2bins(beg, end):
    """bins that may hold a record overlapping [beg, end) -- SAM specification section 5.3 (min_shift 14, 5 levels)"""
    end -= 1
    out = [0]
    for shift, off in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        out += list(range(off + (beg >> shift), off + (end >> shift) + 1))
    return out


class TabixFile:
    def __init__(self, gz_path):
        self.data = open(gz_path, "rb").read()
        self.members = bgzf_members(self.data)
        assert self.data[-28:] == EOF_MEMBER and self.members[-1][1] == b""
        self.text = b"".join(x for _, x in self.members)
        self.start, acc = {}, 0                       # member file offset -> offset of its text in the whole text
        for off, x in self.members:
            self.start[off] = acc
            acc += len(x)
        self.tbi = read_tbi(gz_path + ".tbi")

    def _abs(self, voff):
        coff, uoff = voff >> 16, voff & 0xffff
        assert coff in self.start, f"virtual offset {voff:#x} does not name a member"
        return self.start[coff] + uoff

    def fetch(self, name, beg=0, end=1 << 29):
        """lines of `name` whose interval [pos - 1, pos) overlaps [beg, end): through the bins and the linear index only"""
        tbi = self.tbi
        if name.encode() not in tbi["names"]:
            raise ValueError(f"unknown sequence {name}")
        ref = tbi["refs"][tbi["names"].index(name.encode())]
        w = beg >> 14
        min_off = ref["ioff"][w] if w < len(ref["ioff"]) else (ref["ioff"][-1] if ref["ioff"] else 0)
        chunks = sorted(c for b in reg2bins(beg, end) if b in ref["bins"] and b != 37450 for c in ref["bins"][b] if c[1] > min_off)
        out = []
        for cb, ce in chunks:
            at, stop = self._abs(max(cb, min_off)), self._abs(ce)
            while at < stop:
                nl = self.text.index(b"\n", at)
                f = self.text[at:nl].split(b"\t")
                p = int(f[1])
                if f[0] == name.encode() and p - 1 < end and p > beg:
                    out.append(self.text[at:nl + 1])
                at = nl + 1
        return out
