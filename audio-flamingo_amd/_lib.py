"""ctypes binding of libafk.so (C ABI declared in include/afk.h).

The argtypes are derived from the header itself, so the binding cannot drift from the ABI.  There is NO CPU
fallback: if the library is missing or a call fails, an exception is raised (SURVEY.md §8b / tier rule ③).
"""
from __future__ import annotations

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("AFK_LIB_PATH") or os.path.join(_HERE, "lib", "libafk.so")  # AFK_LIB_PATH: A/B a second build of the same ABI
HEADER_PATH = os.path.join(_REPO, "include", "afk.h")

_CTYPE = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


class AfkError(RuntimeError):
    pass


def parse_header(path: str = HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in afk.h"""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int64_t|int)\s+(afk_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if a.startswith("const char*"):
                    argtypes.append(ctypes.c_char_p)
                    argnames.append(a.split("*")[-1].strip())
                elif "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split("*")[-1].strip())
                else:
                    parts = a.replace("const ", "").split()
                    argtypes.append(_CTYPE[parts[0]])
                    argnames.append(parts[-1])
        protos[name] = (ctypes.c_char_p if ret.startswith("const char") else ctypes.c_int64 if ret == "int64_t" else ctypes.c_int, argtypes, argnames)
    return protos


_lib = None
_protos = None


def source_hash() -> str:
    """sha256 prefix of the kernel sources in this tree, in the Makefile's order (csrc/*.hip sorted, then the headers)"""
    import glob
    import hashlib

    csrc = os.path.join(_HERE, "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")), key=os.path.basename)
    files += [os.path.join(csrc, "common.h"), os.path.join(csrc, "gemm_common.h"), HEADER_PATH]
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AfkError(
            f"libafk.so not found at {LIB_PATH}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C audio-flamingo_amd/csrc`). There is no CPU fallback."
        )
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (ret, argtypes, _) in _protos.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = ret
        fn.argtypes = argtypes
    # a deployment may ship libafk.so + afk.h without csrc/: the overrides are honoured BEFORE the sources are opened, and missing sources
    # mean "cannot verify" (a warning), never an error - only a real mismatch raises
    if "AFK_LIB_PATH" not in os.environ and os.environ.get("AFK_ALLOW_STALE_LIB") != "1":
        built = lib.afk_build_id().decode()
        try:
            tree = source_hash()
        except OSError as e:
            import warnings

            warnings.warn(f"audio_flamingo_amd: kernel sources not found ({e}); cannot verify that {LIB_PATH} (build {built}) matches them")
            tree = built
        if built != tree:
            raise AfkError(f"{LIB_PATH} was built from other sources (library {built}, tree {tree}): rebuild it "
                           f"(`make -C audio-flamingo_amd/csrc`); AFK_ALLOW_STALE_LIB=1 overrides")
    _lib = lib
    return lib


def has_probes() -> bool:
    """was the loaded library built with -DAFK_PROBES (timing probes / rejected GEMM schedules compiled in)?  False for the shipped build"""
    return bool(load().afk_has_probes())


def prototypes():
    load()
    return _protos


def call(name: str, *args):
    """Call an int-returning afk_* entry point; raise AfkError with afk_last_error() on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise AfkError(f"{name} failed ({rc}): {lib.afk_last_error().decode()}")
    return rc
