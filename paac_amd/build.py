"""In-tree build of libpaac_hip.so (gfx950 only).  Used by __graft_entry__.build() and `python -m paac_amd.build`."""
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
    """extra_flags / lib_path / obj_suffix: diagnostic variants (e.g. -DPAAC_DMM_STAMPS into libpaac_hip_stamps.so)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = lib_path or LIB
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", obj_suffix + ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([hipcc] + FLAGS + list(extra_flags) + ["-c", s, "-o", o])

    def run(cmd):
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


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        print(build(extra_flags=["-DPAAC_DMM_STAMPS"], lib_path=os.path.join(HERE, "libpaac_hip_stamps.so"),
                    obj_suffix="_stamps"))
    else:
        print(build(force="--force" in sys.argv))
