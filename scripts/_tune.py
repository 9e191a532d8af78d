"""scripts/ only: A/B work against a variant build of the same ABI.  recad_amd reads no environment variable; a probe that is to
run on `make tuning`'s library says so here -- RECAD_TUNING_LIB=<path or file name under recad_amd/lib/> -- and this module binds
it through the explicit `_lib.load(path)` before anything has loaded the product library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_p = os.environ.get("RECAD_TUNING_LIB")
if _p:
    from recad_amd import _lib

    _lib.load(_p)
    print(f"[scripts/_tune] bound {_lib.LIB_PATH}", file=sys.stderr, flush=True)
