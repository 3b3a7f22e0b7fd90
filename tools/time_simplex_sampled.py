import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from statdepth_amd import engine, _native
import ctypes, os
lib = _native.load()
dev = torch.device('cuda:0')
for d, n, S in ((3, 1000000, 4096), (2, 1000000, 4096), (4, 200000, 4096)):
    P = torch.from_numpy(np.random.default_rng(1237).normal(size=(n, d))).to(dev)
    o = torch.empty(n, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    wss = int(lib.sd_simplex_sampled_workspace_bytes(n, 0, d, S)) if os.environ.get('SD_NO_WS') is None else 0
    ws = torch.empty(max(wss, 8), dtype=torch.uint8, device=dev)
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        rc = lib.sd_pointcloud_simplex_sampled(P.data_ptr(), n, d, 0, n, 1e-7, S, 1237, o.data_ptr(), ws.data_ptr(), wss, st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"d={d} n={n} S={S}: {dt*1e3:.1f} ms, {n*S/dt:.3e} tests/s, checksum {int(o.sum())}", flush=True)
