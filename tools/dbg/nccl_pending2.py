"""dev: does an async all-reduce behind a 1.5 s spin kernel stay pending while another stream captures (thread_local)?"""
import os, socket, sys, time, threading
import torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'transtacos-retunegan_amd'))
from rtg.lib import new_stream
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
side = torch.cuda.Stream()
cap = new_stream()
buf = torch.ones(1 << 16, device='cuda')
x = torch.ones(1 << 20, device='cuda')
dist.all_reduce(buf)
torch.cuda.synchronize()


def poll(w, box, stop):
    while not stop.is_set():
        try:
            box.append(('done' if w.is_completed() else 'pending', time.time()))
        except Exception as e:  # noqa: BLE001
            box.append((str(e).split('\n')[0][:80], time.time()))
        time.sleep(0.025)


keep = []
for mode in ('nocapture', 'capture-keep', 'capture-keep', 'capture-drop', 'capture-drop', 'nocapture', 'capture-global-mode-skip'):
    if mode.endswith('skip'):
        continue
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(side):
        e0.record()
        torch.cuda._sleep(3_600_000_000)
        e1.record()
        w = dist.all_reduce(buf, async_op=True)
    box, stop = [], threading.Event()
    th = threading.Thread(target=poll, args=(w, box, stop))
    t0 = time.time()
    th.start()
    if mode == 'capture-drop':
        keep.clear()
    if mode.startswith('capture'):
        g = torch.cuda.CUDAGraph()
        cap.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap, capture_error_mode='thread_local'):
                y = x * 2
                time.sleep(0.4)
                y = y + 1
        torch.cuda.current_stream().wait_stream(cap)
        keep.append(g)
    else:
        time.sleep(0.5)
    t1 = time.time()
    stop.set(); th.join()
    inside = [r for r, t in box if t0 < t < t1]
    t_w = time.time()
    w.wait(); torch.cuda.synchronize()
    print(f'{mode:14s} window {t1 - t0:5.2f} s: polls inside {len(inside)} pending {inside.count("pending")} done {inside.count("done")} '
          f'other {[r for r in inside if r not in ("pending", "done")][:1]}; spin {e0.elapsed_time(e1):7.1f} ms; sync after window {time.time() - t_w:5.2f} s', flush=True)
dist.destroy_process_group()
