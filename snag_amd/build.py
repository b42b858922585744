"""Build libsnag_hip.so for gfx950 with hipcc (no torch, no cmake).

    python -m snag_amd.build          # incremental
    python -m snag_amd.build --force

The shared library is written next to this file (snag_amd/libsnag_hip.so): it is git-ignored
but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libsnag_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc", "-ffp-contract=off", "-fvisibility=hidden"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "snag_hip.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(path)
            and os.path.getmtime(obj) > _deps_mtime()):
        return obj, False
    cmd = [HIPCC, *FLAGS, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(4, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        # link to a private name and rename: the library appears atomically, so a rank that polls for it (ensure_built) can never
        # dlopen a half-written file
        tmp = f"{LIB}.tmp.{os.getpid()}"
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", tmp, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        os.replace(tmp, LIB)
        if verbose:
            print(f"built {LIB} from {len(objs)} objects")
    elif verbose:
        print(f"{LIB} up to date")
    return LIB


def ensure_built(local_rank=0, timeout_s=900.0):
    """Make sure the library file exists before anything loads it: the first local rank builds it when it is missing (a fresh
    checkout on the GPU box; normally the prebuilt file travels with the tree), the others wait for the file to appear."""
    import time
    if os.path.exists(LIB):
        return LIB
    if local_rank == 0:
        return build(force=False, verbose=False)
    t0 = time.time()
    while not os.path.exists(LIB):
        if time.time() - t0 > timeout_s:
            raise RuntimeError(f"{LIB} did not appear within {timeout_s:.0f} s")
        time.sleep(0.5)
    return LIB               # build() renames the finished file into place, so existing means complete


if __name__ == "__main__":
    build(force="--force" in sys.argv)
