cd "${GRAFT_REPO_ROOT:-.}"
cat > /tmp/pp_dbg.py <<'PY'
import sys, os, ctypes
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from gemm_bench import bench16, L
ctx = L.context(0)
lib = L.load()
for (M, N, K, nm) in ((4096, 4096, 4096, "square 4k"), (768, 147456, 3072, "w2 fwd 768")):
    bench16(ctx, M, N, K, L.CT_F16, name=nm, tile=24, reps=3)
    d = np.zeros(64, dtype=np.uint64)
    lib.ghn3_debug_read(ctypes.c_void_p(d.ctypes.data))
    d = d.reshape(8, 8)
    print("wave: total barrier vmcnt mfma load  (cycles per k-tile), nkt =", int(d[0, 5]))
    for w in range(8):
        n = max(int(d[w, 5]), 1)
        print(w, [round(float(v) / n, 1) for v in d[w, :5]])
PY
GHN3_PP_DEBUG=4 timeout 300 python /tmp/pp_dbg.py 2>&1 | grep -v amdgpu
