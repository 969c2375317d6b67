"""Build libeventclip_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m eventclip_amd.build [--force] [--jobs N] [--diag]

Every ``csrc/*.hip`` is compiled to an object next to it (cached by mtime) and
linked into ``eventclip_amd/libeventclip_hip.so``.  The .so is git-ignored but
travels with the tree to the GPU box.  ``--diag`` builds ``libeventclip_hip_diag.so`` with
-DEC_GEMM_DIAG -DEC_ATTN_DIAG -DEC_EVENTS_DIAG instead: the same library plus ec_gemm's timing / stamp /
timeline variants (csrc/gemm_diag.inc) and the attention / events phase stamps, for
tools/ only (the product never loads it).
"""
import argparse
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
LIB = os.path.join(HERE, 'libeventclip_hip.so')

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
         '-fno-gpu-rdc', '-Wall', '-Wno-unused-function', '-I', INCLUDE]

# the diagnostic build: each translation unit keys its own extras on its own macro
DIAG_FLAGS = ['-DEC_GEMM_DIAG', '-DEC_ATTN_DIAG', '-DEC_EVENTS_DIAG']


def _newest_header():
    hs = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(CSRC, '*.inc')) + glob.glob(os.path.join(INCLUDE, '*.h'))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force, diag=False):
    obj = src[:-4] + ('.diag.o' if diag else '.o')
    stamp = max(os.path.getmtime(src), _newest_header())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= stamp:
        return obj, False
    cmd = [HIPCC] + FLAGS + (DIAG_FLAGS if diag else []) + ['-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed on {os.path.basename(src)}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, jobs=None, verbose=False, diag=False):
    lib = LIB.replace('.so', '_diag.so') if diag else LIB
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    if not srcs:
        raise RuntimeError('no HIP sources found')
    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    with ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, force, diag), srcs))
    objs = [o for o, _ in res]
    rebuilt = any(c for _, c in res)
    if rebuilt or force or not os.path.exists(lib) or \
            os.path.getmtime(lib) < max(os.path.getmtime(o) for o in objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print('linked', lib)
    return lib


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--jobs', type=int, default=None)
    ap.add_argument('--diag', action='store_true', help='libeventclip_hip_diag.so (-DEC_GEMM_DIAG)')
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs, verbose=True, diag=a.diag))
