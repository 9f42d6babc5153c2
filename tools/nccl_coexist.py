import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
dist.barrier()
t = torch.tensor([3.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); print("all_reduce", t.item())
import vcf2multialign_amd as v2m
from vcf2multialign_amd.sharding import max_over_ranks
import __graft_entry__ as g
g.smoke()
print("max_over_ranks", max_over_ranks(1.25, dist, dev))
flags = torch.tensor([1, 3], dtype=torch.int64, device=dev); dist.all_reduce(flags, op=dist.ReduceOp.MIN); print(flags.tolist())
dist.barrier(); dist.destroy_process_group(); print("nccl + libv2m_hip coexist: ok")
