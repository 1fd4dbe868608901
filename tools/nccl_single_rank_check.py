import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
from grok_alpha_zero_amd import parallel
import grok_alpha_zero_amd.parallel as P
# force the collective path although world_size is 1
t = torch.from_numpy(np.array([1, 2, 3], np.int64)).to("cuda"); dist.all_reduce(t); print("all_reduce", t.cpu().tolist(), dist.get_backend())
dist.barrier(); tt = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(tt, op=dist.ReduceOp.MAX); print("max", float(tt.item()))
dist.destroy_process_group(); print("ok")
