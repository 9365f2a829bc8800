"""In-tree build of libpaac_hip.so (gfx950 only).  Used by __graft_entry__.build() and `python -m paac_amd.build`."""
import fcntl
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpaac_hip.so")
SOURCES = ["api.hip", "net_fwd.hip", "net_bwd.hip", "misc.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "paac_hip.h")]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-but-set-variable"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, extra_flags=(), lib_path=None, obj_suffix=""):
    """extra_flags / lib_path / obj_suffix: diagnostic variants (e.g. -DPAAC_DMM_STAMPS into libpaac_hip_stamps.so).
    Safe to call from several processes at once (every torchrun rank calls it for a user architecture): the build runs
    under an exclusive file lock next to the library, objects and library are written to temporary names and renamed into
    place, so a rank that has already loaded the library never sees it rewritten under it."""
    lib = lib_path or LIB
    with open(lib + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, extra_flags, lib, obj_suffix)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose, extra_flags, lib, obj_suffix):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", obj_suffix + ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + list(extra_flags) + ["-c", s, "-o", o])

    def run(cmd):
        # cmd ends in "-o <target>": produce <target>.tmp.<pid>, then rename over the target (atomic on one filesystem)
        at = cmd.index("-o") + 1
        target = cmd[at]
        tmp = "%s.tmp.%d" % (target, os.getpid())
        cmd = cmd[:at] + [tmp] + cmd[at + 1:]
        try:
            _run(cmd)
            os.replace(tmp, target)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)

    def _run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        if verbose and r.stderr.strip():
            print(r.stderr[-4000:], file=sys.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(lib, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


def parse_user_arch(spec):
    """'32,64,64,1024' (filter counts of the reference trunks' layer shapes, then the fc width) or '32:8:4,64:5:2,512'
    (filters:size:stride per layer, then the fc width) -> (convs, fc)."""
    parts = str(spec).split(",")
    convs = []
    for i, p in enumerate(parts[:-1]):
        v = [int(x) for x in p.split(":")]
        if len(v) == 1:
            if i >= len(FAMILY):
                raise ValueError("more than three layers in %r" % (spec,))
            v = [v[0]] + list(FAMILY[i])
        if len(v) != 3:
            raise ValueError("layer %r: expected filters or filters:size:stride" % (p,))
        convs.append(tuple(v))
    return convs, int(parts[-1])


FAMILY = [(8, 4), (4, 2), (3, 1)]      # (kernel size, stride) of the reference trunks' layers (networks.py:145-149, :161-167)


def user_arch_library(convs, fc):
    """In-tree path of the library compiled for a user architecture (convs: [(filters, size, stride), ...])."""
    convs = [tuple(int(v) for v in c) for c in convs]
    tag = "_".join(str(f) for f, _, _ in convs) + "_%d" % fc
    if any((k, s) != FAMILY[i] for i, (_, k, s) in enumerate(convs)):      # other layer shapes: they are part of the name
        tag += "_k" + "_".join("%dx%d" % (k, s) for _, k, s in convs)
    return os.path.join(HERE, "libpaac_hip_user_%s.so" % tag), "_user_" + tag


def build_user_arch(convs, fc, verbose=False):
    """Compile libpaac_hip for a user architecture (include/paac_hip.h: PAAC_ARCH_USER; reference networks.py:117-120): two or
    three VALID conv layers (filters, size, stride) over the 84 x 84 x 4 input, then an fc layer.  The geometry is a
    compile-time template argument of every kernel, so a new architecture is a new build (about a minute of hipcc), cached
    in-tree by its shape.  The reference trunks' layer shapes -- conv 8x8 / 4, conv 4x4 / 2 [, conv 3x3 / 1] -- run on the MFMA
    data-gradient forms; any other kernel size / stride runs its forward and weight gradient on the same generic MFMA
    contraction (dmm.h) and its data gradient on a direct kernel.  Returns the library path."""
    convs = [tuple(int(v) for v in c) for c in convs]
    if len(convs) not in (2, 3) or any(len(c) != 3 for c in convs):
        raise NotImplementedError("user architectures have two or three conv layers (filters, size, stride); got %r" % (convs,))
    if any(f % 16 or f < 16 for f, _, _ in convs) or fc % 256 or fc < 256:
        raise NotImplementedError("filter counts must be multiples of 16 and the fc width a multiple of 256 (MFMA tiles); "
                                  "got %r, fc %d" % (convs, fc))
    if any(k < 1 or s < 1 for _, k, s in convs):
        raise ValueError("kernel sizes and strides must be positive; got %r" % (convs,))
    if (convs[0][1] * 4) % 16:
        raise NotImplementedError("the first layer's kernel size must be 4, 8, 12 or 16 (kernel width x 4 input channels is "
                                  "read in MFMA K groups of 16); got %d" % convs[0][1])
    size = 84
    for _, k, s in convs:
        if size < k:
            raise ValueError("layer shapes %r leave no output (VALID convolutions over 84 x 84)" % (convs,))
        size = (size - k) // s + 1
    lib, suffix = user_arch_library(convs, fc)
    flags = ["-DPAAC_USER_ARCH", "-DPAAC_USER_NCONV=%d" % len(convs), "-DPAAC_USER_C1=%d" % convs[0][0],
             "-DPAAC_USER_C2=%d" % convs[1][0], "-DPAAC_USER_C3=%d" % (convs[2][0] if len(convs) == 3 else 0),
             "-DPAAC_USER_H=%d" % fc]
    layers = convs + [(0, 3, 1)] * (3 - len(convs))
    flags += ["-DPAAC_USER_K%d=%d" % (i + 1, k) for i, (_, k, _) in enumerate(layers)]
    flags += ["-DPAAC_USER_S%d=%d" % (i + 1, st) for i, (_, _, st) in enumerate(layers)]
    return build(verbose=verbose, extra_flags=flags, lib_path=lib, obj_suffix=suffix)


if __name__ == "__main__":
    if "--user-arch" in sys.argv:        # e.g. --user-arch 32,64,64,1024 (filter counts, then the fc width), or
        spec = sys.argv[sys.argv.index("--user-arch") + 1]          # 32:8:4,64:5:2,64:3:1,512 (filters:size:stride per layer)
        print(build_user_arch(*parse_user_arch(spec), verbose=True))
    elif "--stamps" in sys.argv:
        print(build(extra_flags=["-DPAAC_DMM_STAMPS"], lib_path=os.path.join(HERE, "libpaac_hip_stamps.so"),
                    obj_suffix="_stamps"))
    else:
        print(build(force="--force" in sys.argv))
