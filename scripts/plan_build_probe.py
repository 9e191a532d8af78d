"""Host time of the LDS plan builder (rk_lds_plan_build_host) on the ml1m-shaped graph: the cost that sits on the perturb-retrain
loop.  With the tuning build (RECAD_TUNING_LIB=.../librecad_hip_tuning.so): RK_LDS_PLAN_THREADS=n, RK_LDS_PLAN_PIN=0|1.
    python3 scripts/plan_build_probe.py [workload=ml1m] [dim=64]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, '.')
import _tune  # noqa: E402,F401  (binds RECAD_TUNING_LIB's variant build, if set, before the product library is loaded)
from recad_amd import _lib, synth

name = sys.argv[1] if len(sys.argv) > 1 else "ml1m"
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 64
d = synth.make(name)
rp, ci = d["train"][0], d["train"][1]
U = len(rp) - 1
I = int(ci.max()) + 1
R = sp.csr_matrix((np.ones(len(ci), np.float32), ci, rp), shape=(U, I))
A = sp.bmat([[None, R], [R.T, None]]).tocsr()
A.sort_indices()
rp2, cc = A.indptr.astype(np.int32), A.indices.astype(np.int32)
L = _lib.lib()
ts = []
for rep in range(8):
    plan, nw, info = C.c_void_p(), C.c_int64(), _lib.LdsInfo()
    t0 = time.perf_counter()
    rc = L.rk_lds_plan_build_host(U, I, rp2.ctypes.data_as(C.c_void_p), cc.ctypes.data_as(C.c_void_p), None, dim, 256, C.byref(plan), C.byref(nw), C.byref(info))
    ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0
    L.rk_lds_plan_destroy(plan)
print(f"{name} d={dim}: {nw.value} plan words; build ms: " + " ".join(f"{t:.1f}" for t in ts) + f"; cpus {os.cpu_count()}, affinity {len(os.sched_getaffinity(0))}"
      f"; THREADS={os.environ.get('RK_LDS_PLAN_THREADS', '-')} PIN={os.environ.get('RK_LDS_PLAN_PIN', '-')}", flush=True)
