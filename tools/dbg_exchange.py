import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from palace_amd import capi, multigpu, synth
from oracle import binding as orc
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29545")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
rng=synth.rng_for(1); hdr=orc.header_from_picks(rng.integers(0,6,size=32))
ctx=capi.Ctx(0); ctx.eref_set_coder(hdr)
planes=[torch.zeros(1<<29,dtype=torch.uint8,device=dev) for _ in range(3)]
ctx.eref_table_attach([t.data_ptr() for t in planes])
rs=synth.vector_reads(rng, synth.random_dna(rng, 200000), 20000, 150)
db,do=ctx.upload(rs.bases),ctx.upload(rs.offsets)
ctx.eref_table_reset(); ctx.eref_count_reads(db,do,rs.n); ctx.sync()
print("after count", ctx.eref_table_popcounts(), [int((p!=0).sum()) for p in planes])
ex=multigpu.Exchange(torch,dist,0,1)
def merge_fn(parts,n,off,nb):
    torch.cuda.synchronize(); print(" parts nonzero", int((parts!=0).sum()), parts.shape, off, nb)
    ctx.eref_table_merge_slices(parts.data_ptr(), n, off, nb); ctx.sync()
    print(" after merge", ctx.eref_table_popcounts())
ex.merge_planes(planes, merge_fn); torch.cuda.synchronize()
print("after exchange", ctx.eref_table_popcounts())
