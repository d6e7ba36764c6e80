#!/usr/bin/env python3
"""
Build libdlwpcs.so (HIP, gfx950 only) in-tree: dlwp-cs_amd/lib/libdlwpcs.so.

hipcc cross-compiles without a GPU.  The library is linked against the HIP runtime that PyTorch-ROCm ships
(torch/lib/libamdhip64.so) when torch is importable, so that streams / device pointers created by torch are valid
inside the library (one HIP runtime per process); /opt/rocm/lib is the fallback search path.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
TAG = os.environ.get('DLWPCS_LIB_TAG', '')          # development only: instrumented side build (see DLWP/_native.py)
OBJ = os.path.join(HERE, 'build' + ('_' + TAG if TAG else ''))
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libdlwpcs%s.so' % ('_' + TAG if TAG else ''))
SOURCES = ['halo_table.cpp', 'prof.cpp', 'comm.cpp', 'elementwise.hip', 'conv_mfma.hip', 'conv_inst_f32.hip', 'conv_inst_bf16.hip', 'conv_inst_edge.hip',
           'conv_generic.hip', 'wgrad_batch.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
CFLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=' + ARCH, '-x', 'hip', '-Wall', '-Wno-unused-function'] + \
    os.environ.get('DLWPCS_EXTRA_CFLAGS', '').split()


def _torch_lib_dir():
    try:
        import torch
        d = os.path.join(os.path.dirname(torch.__file__), 'lib')
        if os.path.exists(os.path.join(d, 'libamdhip64.so')):
            return d
    except Exception:
        pass
    return None


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any((not os.path.exists(s)) or os.path.getmtime(s) > t for s in (src,) + tuple(extra))


def _deps(obj, fallback):
    """Headers the object was compiled from (the compiler's own -MD list beside it); every header when there is no list yet."""
    d = os.path.splitext(obj)[0] + '.d'
    if not os.path.exists(d):
        return fallback
    try:
        words = open(d).read().replace('\\\n', ' ').split()
    except OSError:
        return fallback
    own = os.path.dirname(HERE)
    return tuple(w for w in words[1:] if w.endswith('.h') and os.path.abspath(w).startswith(own))


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    headers = tuple(os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith('.h')) + \
        (os.path.join(HERE, '..', 'include', 'dlwpcs.h'),)
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, os.path.splitext(src)[0] + '.o')
        objs.append(o)
        if force or _newer(s, o, _deps(o, headers)):
            jobs.append([HIPCC] + CFLAGS + ['-MD', '-MF', os.path.splitext(o)[0] + '.d', '-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr[-8000:]))
        if verbose and r.stderr.strip():
            print(r.stderr[-4000:])

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(run, jobs))
    if jobs or force or not os.path.exists(LIB):
        # Link with g++ (like torch.utils.cpp_extension) so that WE choose which libamdhip64 is recorded as NEEDED:
        # torch's copy has SONAME 'libamdhip64.so', /opt/rocm's 'libamdhip64.so.7'.  Two HIP runtimes in one process
        # would not share streams or allocations, so bind to torch's when it is there.
        link = ['g++', '-shared', '-fPIC', '-o', LIB] + objs
        tl = _torch_lib_dir()
        if tl:
            link += ['-L' + tl]
        link += ['-L/opt/rocm/lib', '-lamdhip64', '-ldl', '-Wl,-rpath,/opt/rocm/lib', '-Wl,--no-undefined']
        run(link)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
