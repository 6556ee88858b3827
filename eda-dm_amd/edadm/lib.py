"""ctypes binding of libedadm.so.  The signatures are parsed from include/edadm.h so the header
stays the single source of truth; `declared_symbols()` lists what the header promises."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
HEADER = os.path.join(_ROOT, "include", "edadm.h")
SO_PATH = os.environ.get("EDADM_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "csrc", "libedadm.so")   # override: instrumented diagnostic builds

_CT = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "float": ctypes.c_float,
    "void": None,
}


def _parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|int64_t|void)\s+(edadm_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    t = a.replace("const", "").split()[0]
                    argtypes.append(_CT[t])
        protos[name] = (_CT[ret], argtypes)
    return protos


PROTOS = _parse_header()


def declared_symbols():
    return sorted(PROTOS)


class EdadmError(RuntimeError):
    pass


_lib = None


def load():
    """Load libedadm.so (built by __graft_entry__.build() / csrc/Makefile).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise EdadmError("libedadm.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(expected at %s)" % SO_PATH)
        lib = ctypes.CDLL(SO_PATH)
        for name, (ret, argtypes) in PROTOS.items():
            fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
            fn.restype = ret
            fn.argtypes = argtypes
        _lib = lib
    return _lib


# tests / tools: a dict here counts the entry points that run (name -> calls), e.g. to assert which contraction kernels a
# reconstruction unit actually took
CALLS = None
# diagnostic (tools/prof_elementwise.sh): EDADM_TRACE_BYTES=<path> installs edadm/trace_bytes.py's hook here -- algorithmic bytes per
# entry point, joined with rocprofv3 counters of the same process by tools/elementwise_hbm.py
TRACE_HOOK = None


def call(name, *args):
    """Invoke an int-returning entry point; non-zero status raises."""
    if CALLS is not None:
        CALLS[name] = CALLS.get(name, 0) + 1
    if TRACE_HOOK is not None:
        TRACE_HOOK(name, args)
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise EdadmError("%s failed with status %d" % (name, rc))


if os.environ.get("EDADM_TRACE_BYTES"):
    from . import trace_bytes as _tb
    _tb.install(os.environ["EDADM_TRACE_BYTES"])
