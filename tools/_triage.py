"""The triage tools run on tools/variants/triage.so (bash tools/mktriage.sh: the library built with -DORBIT_TRIAGE);
the product library reads no environment variable and carries no stamps."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use_triage_library():
    path = os.path.join(ROOT, "tools", "variants", "triage.so")
    if not os.path.exists(path):
        sys.exit(f"{path} is missing: build it here with `bash tools/mktriage.sh` (it travels with the snapshot)")
    sys.path.insert(0, ROOT)
    from orbit_amd import _lib

    assert _lib._lib is None, "call use_triage_library() before anything loads the product library"
    _lib.LIB_PATH = path
    return path
