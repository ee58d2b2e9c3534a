import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dev=torch.device("cuda",0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
for mib in (64, 256, 512, 1024, 1536):
    n = mib << 20
    a = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev); b = torch.zeros_like(a)
    dist.all_to_all_single(b, a); torch.cuda.synchronize()
    bad = int((a != b).sum())
    c = torch.zeros_like(a); dist.all_gather_into_tensor(c, a); torch.cuda.synchronize()
    print(mib, "MiB a2a mismatches:", bad, " allgather mismatches:", int((a != c).sum()), flush=True)
dist.destroy_process_group()
