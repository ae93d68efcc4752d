"""Build libmyolo.so (the gfx950 HIP kernels + C ABI) in-tree with hipcc.  No JIT cache: the .so travels with the repo
snapshot to the GPU box; it is git-ignored."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'lib', 'libmyolo.so')
ARCH = 'gfx950'


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + \
        [os.path.join(os.path.dirname(HERE), 'include', 'myolo.h')]
    return any(os.path.getmtime(d) > t for d in deps)


# augment.hip restates C code compiled for baseline x86-64 (Pillow / OpenCV float arithmetic): a fused multiply-add changes the
# truncated uint8 results.  hipcc's default -ffp-contract=fast ignores source pragmas, so the whole file is built without contraction.
PER_FILE_FLAGS = {'augment.hip': ['-ffp-contract=off']}


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    objs = []
    procs = []
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + \
        [os.path.join(os.path.dirname(HERE), 'include', 'myolo.h')]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    for src in sources():
        obj = os.path.join(HERE, 'lib', os.path.basename(src)[:-4] + '.o')
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), hdr_t):
            continue                                     # object is newer than its source and every header
        cmd = [hipcc, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC'] + PER_FILE_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('hipcc failed: %s\n%s' % (' '.join(cmd), out.decode()))
        if verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n' + r.stdout.decode())
    if verbose:
        print('built', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
