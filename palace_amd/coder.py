"""The 400-byte coder header of an eref index (E1), host side.  Product code: bench.py and the tools build a header
without touching oracle/.  Layout (extract_ref.cpp:680-682, 1104-1122): 100 little-endian u32 words; the low 16 bits of
word j hold choose_coder[j] (j < 96: k-mer offset j // 3, channel j % 3), the high 16 bits repeat entry j + 1 (the
reference writes 4 bytes from a 2-byte array); words 96..99 are zero.  The six orders of (0, 1, 2) that random_coder
draws from (extract_ref.cpp:1082-1102) are ORDERS; palace_amd/host/eref_main.cpp make_header() writes the same layout."""
import numpy as np

ORDERS = np.array([[0, 1, 2], [0, 2, 1], [1, 2, 0], [1, 0, 2], [2, 0, 1], [2, 1, 0]], dtype=np.uint16)


def header_from_picks(picks) -> np.ndarray:
    """Header for one of the six projection orders per k-mer offset (picks: 32 values in 0..5)."""
    p = np.asarray(picks, dtype=np.int64)
    assert p.shape == (32,) and p.min() >= 0 and p.max() < 6
    cc = np.zeros(100, dtype=np.uint16)
    cc[:96] = ORDERS[p].reshape(-1)
    words = cc[:100].astype(np.uint32)
    words[:95] |= cc[1:96].astype(np.uint32) << 16          # entry 95's upper half is entry 96 = 0
    words[96:] = 0
    return words.astype("<u4").view(np.uint8).copy()
