"""Builds libcdrl_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libcdrl_hip.so')
OBJ = os.path.join(HERE, 'build')
SOURCES = ['common', 'bn', 'gemm', 'gemm_tn_direct', 'gemm_tn_lds', 'gemm_pw', 'gemm_pw_bf16', 'gemm_pw_x3', 'gemm_pw_bwd', 'gemm_x3', 'conv', 'dwfused', 'rnn', 'heads', 'loss', 'optim', 'gae', 'sample', 'augment', 'engine', 'capi']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wall', '-Wno-unused-function']


def _hipcc():
    for c in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return 'hipcc'


def _digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, '..', 'include')):
        for f in sorted(os.listdir(root)):
            if f.endswith(('.hip', '.h')):
                h.update(f.encode())
                h.update(open(os.path.join(root, f), 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _headers_digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, '..', 'include')):
        for f in sorted(os.listdir(root)):
            if f.endswith('.h'):
                h.update(f.encode())
                h.update(open(os.path.join(root, f), 'rb').read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _compile(name, hdig=None, force=False):
    """One translation unit; skipped when its source, every header and the flags are unchanged (per-object stamp)."""
    src = os.path.join(CSRC, name + '.hip')
    obj = os.path.join(OBJ, name + '.o')
    stamp = obj + '.digest'
    dig = hashlib.sha256((hdig or _headers_digest()).encode() + open(src, 'rb').read()).hexdigest()
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [_hipcc()] + FLAGS + ['-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {name}.hip:\n{r.stdout}\n{r.stderr}')
    open(stamp, 'w').write(dig)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, 'digest.txt')
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        if verbose:
            print('[cdrl] libcdrl_hip.so up to date')
        return LIB
    hdig = _headers_digest()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(lambda n: _compile(n, hdig, force), SOURCES))
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    open(stamp, 'w').write(dig)
    if verbose:
        print('[cdrl] built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
