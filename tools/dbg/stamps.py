"""GPU box, diagnostic build tools/ab/lib_stamps.so (-DPALACE_STAMPS): mean time between the phase stamps of the level-1
partition kernel, per workgroup (s_memrealtime, 100 MHz).  usage: PALACE_HIP_SO=tools/ab/lib_stamps.so python tools/dbg/stamps.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from palace_amd import capi, coder
dev = torch.device("cuda", 0)
sample = bench.make_sample(torch, dev, 1_000_000, 5000)
hdr = coder.header_from_picks(np.random.Generator(np.random.PCG64(1)).integers(0, 6, size=32))
L = capi.lib()
with capi.Ctx(0) as ctx:
    ctx.eref_set_coder(hdr)
    n = sample["n_reads_side"]
    for _ in range(3):
        ctx.eref_table_reset()
        capi._check(L.palace_eref_count_reads(ctx.h, sample["r12"].data_ptr(), sample["read_off"].data_ptr(), 2 * n, None, 2 * n * 150), "count")
    ctx.sync()
    buf = np.zeros(8 * 65536, dtype=np.uint64)
    L.palace_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    capi._check(L.palace_debug_stamps(ctx.h, buf.ctypes.data, len(buf)), "stamps")
s = buf.reshape(-1, 8).astype(np.int64)
s = s[s[:, 0] > 0]
d = np.diff(s[:, :7], axis=1) / 100.0          # microseconds
names = ["entry -> keys computed, histogram adds issued", "barrier (all histogram adds done)", "prefix (wave 0) + barrier", "reserve issued, placement, barrier", "sweep issued", "stores acked"]
print(f"{len(s)} workgroups sampled; total residency {np.mean(s[:, 6] - s[:, 0]) / 100:.2f} us (median {np.median(s[:, 6] - s[:, 0]) / 100:.2f})")
for k, nm in enumerate(names):
    print(f"  {nm:40s} mean {d[:, k].mean():6.2f} us   median {np.median(d[:, k]):6.2f}   p90 {np.percentile(d[:, k], 90):6.2f}")
