"""
Builds libghn3_hip.so (gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU.

    python -m ghn3_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OUT_DIR = os.path.join(HERE, 'lib')
LIB = os.path.join(OUT_DIR, 'libghn3_hip.so')
SOURCES = ['gemm.hip', 'gemm_p8.hip', 'gemm_small.hip', 'gemm_x3.hip', 'gemm_x3d.hip', 'gemm_wg.hip', 'attention.hip', 'elementwise.hip', 'target_ops.hip', 'runtime.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
         '-I' + CSRC, '-Wno-unused-result'] + os.environ.get('GHN3_HIPCC_EXTRA', '').split()


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_hash():
    """sha256 over the HIP sources, the headers and the host compiler (what determines which kernels a step launches and how):
    recorded with every PMC traffic figure (tools/pmc_traffic.py) so that bench.py never reports a figure measured on other
    code, and printed by __graft_entry__.build()."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(('.hip', '.h'))]
    files += [os.path.join(ROOT, 'include', 'ghn3_hip.h'), os.path.join(HERE, 'program.py')]
    for f in files:
        with open(f, 'rb') as fh:
            h.update(os.path.basename(f).encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(ROOT, 'include', 'ghn3_hip.h'), os.path.join(CSRC, 'ghn3_internal.h')]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT_DIR, src.replace('.hip', '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + r.stdout)
        return r.stdout

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
