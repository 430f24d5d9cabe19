"""Hash of the native sources a libdyk_hip.so is built from.

The Makefile runs this file as a script and compiles the digest into the library (`dyk_build_sha()`); `dyk/lib.py`
recomputes it from the tree and refuses a library whose digest differs -- a stale object with the right ABI version does
not load.  No imports beyond the standard library: the Makefile must not pay for `import torch`.
"""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(_HERE)
ROOT = os.path.dirname(PKG)


def native_sources():
    csrc = os.path.join(PKG, "csrc")
    files = glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) \
        + [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "dyk_hip.h")]
    return sorted(f for f in files if os.path.basename(f) not in ("probe.hip", "build_sha.h"))


def native_sha():
    """digest of the in-tree native sources, or None when the sources are not there (a deployment that ships only the
    library: nothing to compare the build against)"""
    h = hashlib.sha1()
    try:
        for f in native_sources():
            h.update(os.path.relpath(f, ROOT).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    except OSError:
        return None
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(native_sha())
