import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ["PBA_NO_TORCH"] = "1"
import numpy as np
from pacbioassembly_amd import engine as eng
from oraclelib import Oracle
O = Oracle()
g = eng.synth_genome(2, 600000)
reads, offs, starts = eng.synth_reads(3, g, 4, 15000, nthreads=4)
for r in range(3):
    a = reads[int(offs[r]):int(offs[r+1])].tobytes(); b = g[int(starts[r]):int(starts[r])+19600].tobytes()
    for rep in range(3):
        t = time.time(); res = O.align(a, b, 0.3); dt = time.time() - t
        print(f"read {r} rep {rep}: {dt:.2f} s rc={res['rc']} cells={res['cells']} -> {res['cells']/dt/1e9:.3f} GCUPS", flush=True)
# raw page-touch rate
n = 1 << 30
t = time.time(); x = np.empty(n, np.uint8); x[::4096] = 1; print("first touch 1 GiB (4K stride): %.2f s" % (time.time() - t), flush=True)
t = time.time(); x[::4096] = 2; print("second touch: %.3f s" % (time.time() - t), flush=True)
t = time.time(); x[:] = 3; print("memset 1 GiB: %.3f s" % (time.time() - t), flush=True)
