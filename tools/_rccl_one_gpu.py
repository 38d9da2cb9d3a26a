"""Does RCCL accept two ranks on ONE GPU?  (the GPU box has a single device; the driver's 8-GPU
node is the only place the real xGMI path runs)"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', rank=rank, world_size=int(os.environ['WORLD_SIZE']))
    t = torch.full((4,), float(rank), device='cuda:0')
    out = [torch.zeros(4, device='cuda:0') for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    torch.cuda.synchronize()
    print('rank', rank, 'all_gather ok:', [o[0].item() for o in out], flush=True)
    dist.destroy_process_group()
except Exception as e:
    print('rank', rank, 'RCCL on one GPU failed:', type(e).__name__, str(e)[:300], flush=True)
    sys.exit(0)
