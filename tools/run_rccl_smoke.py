"""Runs tests/test_dp_gpu.py's RCCL world-of-one script directly (full stdout / stderr): a debugging aid."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_dp_gpu as t
env = dict(os.environ, PAAC_DIST_FORCE="1", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
           MASTER_PORT=str(t._free_port()))
res = subprocess.run([sys.executable, "-c", t._RCCL_SMOKE % dict(root=ROOT)], cwd=ROOT, env=env, capture_output=True, text=True,
                     timeout=600)
print("rc", res.returncode)
print(res.stdout[-3000:])
print(res.stderr[-6000:])
