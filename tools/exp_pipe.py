#!/usr/bin/env python3
"""Large-n route with pipelined batches (front stages on a helper stream, two copies of the scratch): the cross-check library
under SD_BIG_PIPE_MB (the budget that sets the rows per batch; apply tools/experiments/big_pipelined_batches.patch first), totals compared with the one-stream form on the same matrix,
then timed with several matrices in rotation.  usage: exp_pipe.py n T [reps]; SD_PIPE_MBS = budgets in MiB (0: one stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from statdepth_amd import engine, _native
lib = os.environ.get("SD_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                "statdepth_amd", "lib", "libstatdepth_hip_xcheck.so")
_native.LIB_PATH = os.path.abspath(lib)
_native._LIB = _native.open_library(_native.LIB_PATH)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
from _rankdata import rank_data
X = rank_data(n, T)
ROT = int(os.environ.get("SD_ROTATE", "3"))
Xs = [engine.to_device_matrix(X + float(k)) for k in range(ROT)]
os.environ.pop("SD_BIG_PIPE_MB", None)
os.environ["SD_BIG_PIPE_MB"] = "-1"
want = [engine.mbd_counts(x, None, 2, algo="rank", return_tensor=True).clone() for x in Xs]
torch.cuda.synchronize()
for mb in os.environ.get("SD_PIPE_MBS", "-1,52,103,205").split(","):
    os.environ["SD_BIG_PIPE_MB"] = mb
    engine.release_workspace()
    bad = 0
    for rnd in range(3):
        for k, x in enumerate(Xs):
            got = engine.mbd_counts(x, None, 2, algo="rank", return_tensor=True)
            bad += int(not torch.equal(got, want[k]))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        engine.mbd_counts(Xs[i % ROT], None, 2, algo="rank", return_tensor=True)
    e1.record()
    torch.cuda.synchronize()
    rows = (int(mb) << 20) // (n * 16) & ~7 if int(mb) > 0 else 0
    print(f"n={n} T={T} budget {mb} MiB (rows per batch {rows or 'plan'}): {e0.elapsed_time(e1) / reps:.4f} ms per call, "
          f"mismatching calls {bad}", flush=True)
