"""palace_bgzf_inflate (csrc/inflate.hip: one wavefront per BGZF member) against zlib: every kind of DEFLATE block (stored, fixed,
dynamic), every zlib strategy and level, members of 0 .. 65536 bytes, several blocks per member, payloads at arbitrary byte
offsets, and damaged streams -- where the rule is the one of the loader's CPU decoder (host/inflate_fast.hpp): a member is
either refused (status != 0; zlib then decides it on the host) or decoded to exactly the bytes zlib gives."""
import zlib

import numpy as np
import pytest

from palace_amd import capi, synth

pytestmark = pytest.mark.gpu


def deflate(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, mem=8, cut=None) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, mem, strategy)
    if cut is None:
        return c.compress(data) + c.flush()
    out = b""
    for i in range(0, len(data), cut):                     # a full flush between the pieces: several blocks, empty stored blocks
        out += c.compress(data[i:i + cut]) + c.flush(zlib.Z_FULL_FLUSH if (i // cut) % 2 else zlib.Z_SYNC_FLUSH)
    return out + c.flush()


def zlib_says(comp: bytes, out_len: int):
    """(ok, bytes): what zlib makes of a raw DEFLATE stream that should give out_len bytes and end inside `comp`"""
    d = zlib.decompressobj(-15)
    try:
        out = d.decompress(comp, out_len + 1)
    except zlib.error:
        return False, b""
    return bool(d.eof and len(out) == out_len), out


def payloads(rng):
    kinds = []
    bamish = bytes(rng.choice(np.frombuffer(b"ACGT!#5?IIII\x00\x11\x22\x44\x88", np.uint8), size=65280))      # few symbols: short codes, many matches
    text = (b"EDGE_123_length_4567_cov_8.9\t150M\tNM:i:2\tSA:Z:EDGE_77,1201,+,90S60M,60,0;\n" * 900)[:65000]
    noise = bytes(rng.integers(0, 256, size=65536, dtype=np.uint8))                                          # incompressible: stored blocks / long codes
    runs = b"".join(bytes([int(b)]) * int(n) for b, n in zip(rng.integers(0, 256, 400), rng.integers(1, 600, 400)))[:65536]   # long matches, distance 1
    skew = bytes(np.minimum(255, rng.geometric(0.02, size=60000)).astype(np.uint8))                          # many distinct symbols, codes up to 15 bits
    for name, data in (("bamish", bamish), ("text", text), ("noise", noise), ("runs", runs), ("skew", skew), ("zeros", bytes(65536)),
                       ("one", b"x"), ("empty", b""), ("short", b"abcabcabcabc" * 3)):
        kinds.append((name, data))
    return kinds


def run_members(ctx, members, rng):
    """members: list of (compressed bytes, out_len).  Returns (status array, list of output bytes)"""
    L = capi.lib()
    in_off, out_off, blob, o = [], [], bytearray(), 0
    for comp, out_len in members:
        blob += bytes(rng.integers(0, 256, size=int(rng.integers(0, 7)), dtype=np.uint8))     # arbitrary alignment of the payload
        in_off.append(len(blob))
        blob += comp
        out_off.append(o)
        o += out_len + int(rng.integers(0, 5))                                               # ... and of the output
    blob += bytes(8)
    n = len(members)
    d_in = ctx.upload(np.frombuffer(bytes(blob), np.uint8))
    d_io, d_il = ctx.upload(np.array(in_off, np.int64)), ctx.upload(np.array([len(c) for c, _ in members], np.int32))
    d_oo, d_ol = ctx.upload(np.array(out_off, np.int64)), ctx.upload(np.array([k for _, k in members], np.int32))
    d_out = ctx.upload(np.full(o + 8, 0xEE, np.uint8))
    d_st = ctx.upload(np.full(n, -1, np.int32))
    capi._check(L.palace_bgzf_inflate(ctx.h, d_in.ptr, n, d_io.ptr, d_il.ptr, d_oo.ptr, d_ol.ptr, d_out.ptr, d_st.ptr), "palace_bgzf_inflate")
    ctx.sync()
    st, out = d_st.to_host(), d_out.to_host().tobytes()
    for b in (d_in, d_io, d_il, d_oo, d_ol, d_out, d_st):
        b.free()
    # nothing outside the members' ranges may have been written
    mask = np.ones(o + 8, bool)
    for off, (_, k) in zip(out_off, members):
        mask[off:off + k] = False
    assert (np.frombuffer(out, np.uint8)[mask] == 0xEE).all(), "bytes outside a member's output range were written"
    return st, [out[off:off + k] for off, (_, k) in zip(out_off, members)]


def test_every_block_kind_strategy_and_level_equals_zlib():
    rng = synth.rng_for(404)
    members, want, names = [], [], []
    for name, data in payloads(rng):
        for level in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED):
                comp = deflate(data, level, strat)
                if len(comp) > 70000:
                    continue
                members.append((comp, len(data))); want.append(data); names.append((name, level, strat))
        for cut in (1000, 20000):                              # several blocks and empty stored blocks inside one member
            members.append((deflate(data, 6, cut=cut), len(data))); want.append(data); names.append((name, "cut", cut))
        members.append((deflate(data, 9, mem=1), len(data))); want.append(data); names.append((name, "mem1", 0))   # small hash: many short blocks
    with capi.Ctx(0) as ctx:
        st, got = run_members(ctx, members, rng)
    bad = [(names[i], int(st[i])) for i in range(len(members)) if st[i] != 0]
    assert not bad, f"valid streams refused: {bad[:10]}"
    for i, (g, w) in enumerate(zip(got, want)):
        assert g == w, names[i]
    assert len(members) > 200


def test_bam_members_of_the_synthetic_writer(tmp_path):
    """the members of a BAM as the test writer makes them (BGZF framing walked here: 18-byte header with the BC field, payload,
    CRC32 + ISIZE), all inflated in one call: the concatenation is the BAM stream"""
    import gzip
    import struct
    rng = synth.rng_for(11)
    targets, fai_text, recs, avg = synth.random_graph_case(rng, 300, 40000)
    path = str(tmp_path / "s.bam")
    synth.write_bam(path, targets, recs)
    raw = open(path, "rb").read()
    members, p = [], 0
    while p < len(raw):
        assert raw[p:p + 4] == b"\x1f\x8b\x08\x04"
        xlen = struct.unpack_from("<H", raw, p + 10)[0]
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1     # (the writer puts BC first: SI1 SI2 SLEN BSIZE)
        assert raw[p + 12:p + 14] == b"BC"
        isize = struct.unpack_from("<I", raw, p + bsize - 4)[0]
        members.append((raw[p + 12 + xlen:p + bsize - 8], isize))
        p += bsize
    assert len(members) > 20
    with capi.Ctx(0) as ctx:
        st, got = run_members(ctx, members, rng)
    assert (st == 0).all(), st[st != 0][:10]
    assert b"".join(got) == gzip.decompress(raw)


def test_damaged_streams_are_refused_or_decoded_as_zlib_decodes_them():
    rng = synth.rng_for(505)
    base = []
    for name, data in payloads(rng):
        if len(data) == 0:
            continue
        for level, strat in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)):
            comp = deflate(data[:20000], level, strat)
            base.append((comp, len(data[:20000])))
    members = []
    for comp, n in base:
        for _ in range(12):
            c = bytearray(comp)
            kind = int(rng.integers(0, 5))
            if kind == 0 and len(c) > 2:
                c = c[: int(rng.integers(1, len(c)))]                                  # truncated
            elif kind == 1:
                c[int(rng.integers(0, len(c)))] ^= 1 << int(rng.integers(0, 8))          # one bit flipped
            elif kind == 2:
                for _k in range(4):
                    c[int(rng.integers(0, min(len(c), 40)))] = int(rng.integers(0, 256))   # the block header / code lengths hit
            elif kind == 3:
                c += bytes(rng.integers(0, 256, size=5, dtype=np.uint8))                # bytes behind the end of the stream
            else:
                n2 = max(0, n + int(rng.integers(-3, 4)))                               # the size is wrong, the stream is fine
                members.append((bytes(c), n2))
                continue
            members.append((bytes(c), n))
    with capi.Ctx(0) as ctx:
        st, got = run_members(ctx, members, rng)
    accepted = refused = 0
    for (comp, n), s, g in zip(members, st.tolist(), got):
        ok, out = zlib_says(comp, n)
        if s == 0:
            assert ok and g == out, "the device decoder accepted a member zlib does not decode to these bytes"
            accepted += 1
        else:
            refused += 1
    assert refused > 100 and accepted > 10
