"""Builds the HIP shared library (libptrs_hip.so) in-tree for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off (and no fast-math) is part of the contract:
the kernels must produce the same binary32 results as the CPU oracle (csrc/pt_vec.h).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libptrs_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: the SLP vectoriser pairs independent binary32 operations into v_pk_mul_f32 / v_pk_add_f32 (20 000 of them in
# this library); on these kernels a packed instruction costs more than the two plain ones it replaces and the pairing costs
# registers (k_shade<Matte> 201 -> 183, k_extend_rf pair form 76 -> 64).  Same-box A/B on MI355X: Cornell +3 %, colonnade +4 %,
# classroom +5 % (its shade kernels -11 %).  Same bits either way.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-Wall", "-Wno-unused-function", "-Wno-unused-parameter"]


def _gen_tables_inc():
    src = os.path.join(ROOT, "data", "sobol_tables.bin")
    dst = os.path.join(CSRC, "sobol_tables_data.inc")
    if os.path.exists(dst) and os.path.getmtime(dst) >= os.path.getmtime(src):
        return
    data = open(src, "rb").read()
    with open(dst, "w") as f:
        for i in range(0, len(data), 32):
            f.write(",".join(str(b) for b in data[i:i + 32]) + ",\n")


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".h", ".hip", ".inc"))] + [
        os.path.join(ROOT, "include", "ptrs.h"), os.path.join(ROOT, "include", "ptrs_detmath.h")]


def source_hash(extra_flags=()):
    """sha256 over what determines the kernels: every file of csrc/, the two headers, the compiler flags INCLUDING a build's extra
    flags (-D tuning macros of A/B builds).  The library carries it (ptrs_build_id(), compiled in as PTRS_BUILD_ID); counter summaries
    under profiles/ record the id of the library they were measured with, and bench.py uses a summary only when it equals the id of
    the library it runs."""
    import hashlib
    _gen_tables_inc()
    h = hashlib.sha256(" ".join(FLAGS + sorted(extra_flags)).encode())
    for s in sources():
        h.update(os.path.basename(s).encode())
        h.update(open(s, "rb").read())
    return h.hexdigest()[:16]


def build(force=False, extra_flags=(), verbose=False, out=LIB):
    """Compiles `out` unless it was built from exactly these sources and flags before (a sidecar file next to the library holds the id
    it was built with: mtimes say nothing about flags)."""
    _gen_tables_inc()
    extra_flags = [f for f in extra_flags]
    bid = source_hash(extra_flags)
    side = out + ".buildid"
    if not force and os.path.exists(out) and os.path.exists(side) and open(side).read().strip() == bid:
        return out
    cmd = [HIPCC] + FLAGS + extra_flags + ['-DPTRS_BUILD_ID="%s"' % bid, "-o", out, os.path.join(CSRC, "ptrs_hip.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(side, "w") as f:
        f.write(bid + "\n")
    return out


HEADLESS = os.path.join(HERE, "ptrs_headless")


def build_host(force=False):
    """C++ host mirror + headless CLI (g++, links libptrs_hip.so with rpath $ORIGIN)."""
    srcs = [os.path.join(HERE, "host", f) for f in ("headless.cpp", "ptrs_host.cpp", "ptrs_gltf.cpp")]
    deps = srcs + [os.path.join(HERE, "host", "ptrs_host.hpp"), LIB]
    if not force and os.path.exists(HEADLESS) and all(os.path.getmtime(HEADLESS) >= os.path.getmtime(d) for d in deps):
        return HEADLESS
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wextra", "-o", HEADLESS] + srcs +
                          ["-L" + HERE, "-lptrs_hip", "-lz", "-Wl,-rpath,$ORIGIN"])
    return HEADLESS


if __name__ == "__main__":
    # python -m pathtracer-rs_amd.build [--force] [--variant NAME] [-D... -R... -save-temps]: NAME builds libptrs_NAME.so (A/B builds, PTRS_LIB)
    extra = [a for a in sys.argv[1:] if a.startswith(("-D", "-R", "-save", "-m"))]
    if "--variant" in sys.argv:
        name = sys.argv[sys.argv.index("--variant") + 1]
        print("built", build(force="--force" in sys.argv, verbose=True, extra_flags=extra, out=os.path.join(HERE, "libptrs_%s.so" % name)))
    else:
        build(force="--force" in sys.argv, verbose=True, extra_flags=extra)
        print("built", LIB)
        print("built", build_host(force="--force" in sys.argv))
