"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/palace_hip.h declares.  No compute calls (no GPU here)."""
import ctypes

from palace_amd import capi


def test_library_exports_every_declared_symbol():
    capi.build()
    lib = capi.lib()
    names = capi.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert b"gfx950" in lib.palace_version()


def test_every_declared_symbol_has_a_python_signature():
    have = set(capi._SIGS) | {"palace_last_error", "palace_version", "palace_stream", "palace_match_result_free", "palace_eref_packed_bytes"}
    have |= {"palace_match_result_" + k for k in ("count", "offsets", "verts", "kind", "iter", "open_at", "bare", "bare_count")}
    assert set(capi.declared_symbols()) <= have


def test_ctx_create_without_gpu_fails_loudly_or_works():
    import torch
    h = ctypes.c_void_p()
    rc = capi.lib().palace_ctx_create(0, ctypes.byref(h))
    if torch.cuda.is_available():
        assert rc == 0
        capi.lib().palace_ctx_destroy(h)
    else:
        assert rc < 0 and capi.lib().palace_last_error()


def test_struct_layouts_match_header():
    """sizes the C side static-asserts too (csrc/graph.hip)"""
    assert ctypes.sizeof(capi.GraphParams) == 32
    assert ctypes.sizeof(capi.BamCols) == 8 + 13 * 8


def test_rccl_library_exports_what_its_header_declares():
    """include/palace_rccl.h -> libpalace_rccl.so (the multi-GPU exchange for a C++ host); loading it pulls RCCL in, which
    needs no GPU.  No calls."""
    import os
    import re
    capi.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "palace_rccl.h")).read()
    names = sorted(set(re.findall(r"\b(palace_[a-z0-9_]+)\s*\(", text)) - {"palace_eref_table_pack_low", "palace_eref_table_merge_slices_packed"})
    names = [n for n in names if n != "palace_eref_set_key_buckets"]                  # (libpalace_hip.so's, named in a comment)
    assert names == ["palace_eref_entry_counts_exchange", "palace_eref_key_share", "palace_eref_key_share_gather", "palace_eref_key_share_gather_sparse",
                     "palace_eref_rows_allgather", "palace_eref_table_exchange"]
    capi.lib()                                                   # libpalace_hip.so first: the rccl library links it by name
    lib = ctypes.CDLL(os.path.join(root, "palace_amd", "libpalace_rccl.so"), mode=ctypes.RTLD_GLOBAL)
    for n in names:
        assert hasattr(lib, n), n
    # the C and the Python statement of a rank's share agree
    from palace_amd import multigpu
    for world in (1, 2, 4, 8, 16):
        for rank in range(world):
            m = (ctypes.c_uint32 * 4)()
            assert lib.palace_eref_key_share(rank, world, m) == 0
            assert [b for b in range(128) if (m[b >> 5] >> (b & 31)) & 1] == multigpu.key_buckets_of(rank, world)
