"""dev: what HIP (this ROCm) lets ANOTHER thread do with events while a stream captures — the facts behind
train.CAPTURE_ERROR_MODE and rtg/lib.py:new_stream.  Run on the GPU box: python tools/dbg/capture_semantics.py"""
import os
import sys
import threading

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'transtacos-retunegan_amd'))
from rtg.lib import new_stream  # noqa: E402


def query_from_thread(ev):
    box = []

    def run():
        try:
            box.append(('ok', ev.query()))
        except Exception as e:  # noqa: BLE001
            box.append(('error', str(e).split('\n')[0][:120]))
    t = threading.Thread(target=run)
    t.start()
    t.join()
    return box[0]


def case(name, mode, same_stream, external):
    mk = new_stream if external else torch.cuda.Stream
    a, b = mk(), mk()
    x = torch.ones(1 << 20, device='cuda')
    torch.cuda.synchronize()
    ev = torch.cuda.Event()
    rec = a if same_stream else b
    with torch.cuda.stream(rec):
        y = x * 2
        ev.record()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    res = None
    try:
        with torch.cuda.graph(g, stream=a, capture_error_mode=mode):
            z = x + 1
            res = query_from_thread(ev)
            z = z + 1
        end = 'capture ended ok'
    except Exception as e:  # noqa: BLE001
        end = 'capture failed: ' + str(e).split('\n')[0][:100]
    print(f'{name:62s} mode={mode:12s} query from another thread -> {res}; {end}', flush=True)
    torch.cuda.synchronize()


if __name__ == '__main__':
    import subprocess
    if len(sys.argv) > 1:                    # one case per process: a failed query invalidates the capture for good
        name, mode, same, ext = sys.argv[1:5]
        case(name, mode, same == '1', ext == '1')
        sys.exit(0)
    print('torch', torch.__version__, 'hip', torch.version.hip, flush=True)
    cases = []
    for ext in (0, 1):
        tag = 'own stream (rtg_stream_create)' if ext else 'pooled torch stream'
        cases += [(f'{tag}: event recorded on ANOTHER stream', 'thread_local', 0, ext),
                  (f'{tag}: event recorded on the CAPTURING stream before capture', 'thread_local', 1, ext),
                  (f'{tag}: event recorded on ANOTHER stream', 'relaxed', 0, ext),
                  (f'{tag}: event recorded on ANOTHER stream', 'global', 0, ext)]
    for name, mode, same, ext in cases:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), name, mode, str(same), str(ext)], capture_output=True, text=True)
        out = [l for l in r.stdout.splitlines() if 'mode=' in l]
        print(out[0] if out else f'{name} mode={mode}: process ended with {r.returncode}: {r.stderr.strip().splitlines()[-1][:160] if r.stderr.strip() else ""}', flush=True)
    # the pool wraps: the 33rd pooled stream is the 1st again
    first = torch.cuda.Stream()
    ids = [torch.cuda.Stream().cuda_stream for _ in range(40)]
    print('pooled streams repeat after', 1 + ids.index(first.cuda_stream) if first.cuda_stream in ids else None, 'creations;',
          'own streams distinct:', len({new_stream().cuda_stream for _ in range(8)}) == 8)
