import os, sys
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533")
import torch, torch.distributed as dist
import bench
from efa_xray_amd.distributed import HipEngine
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
eng = HipEngine(0)
print(bench.init_library_comm(eng, 0, 1, dist, torch))
dist.destroy_process_group()
