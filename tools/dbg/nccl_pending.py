"""dev: how long does an async all-reduce issued behind a spin kernel stay pending (1-rank nccl group)?"""
import os, socket, time
import torch, torch.distributed as dist
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
side = torch.cuda.Stream()
buf = torch.ones(1 << 16, device='cuda')
dist.all_reduce(buf)
torch.cuda.synchronize()
for ticks in (20_000_000, 50_000_000, 50_000_000, 200_000_000, 200_000_000):
    for same_stream in (True, False):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.time()
        with torch.cuda.stream(side):
            e0.record()
            torch.cuda._sleep(ticks)
            e1.record()
            if same_stream:
                w = dist.all_reduce(buf, async_op=True)
        if not same_stream:
            torch.cuda.current_stream().wait_stream(side)
            w = dist.all_reduce(buf, async_op=True)
        t_issue = time.time() - t0
        n = 0
        while not w.is_completed():
            time.sleep(0.005); n += 1
        t_done = time.time() - t0
        e1.synchronize()
        print(f'ticks {ticks:>11d} issued-in-side-ctx {same_stream}: issue {t_issue*1e3:6.1f} ms, work completed after {t_done*1e3:7.1f} ms, '
              f'spin kernel {e0.elapsed_time(e1):7.1f} ms', flush=True)
        w.wait(); torch.cuda.synchronize()
dist.destroy_process_group()
