"""Hash of the code a profile belongs to: kernel sources, the C ABI header and the plan compiler.  bench.py reports
numbers taken from profiles/ (HBM traffic, in-step kernel durations) only when this hash matches the running tree."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def code_sha():
    h = hashlib.sha1()
    pkg = os.path.join(ROOT, "double-yolo-kaist_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h"))
                   + glob.glob(os.path.join(pkg, "dyk", "*.py")) + [os.path.join(ROOT, "include", "dyk_hip.h")])
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(code_sha())
