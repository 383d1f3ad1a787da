"""Build the gfx950 shared library (libdta_hip.so) in-tree with hipcc.  No GPU needed: hipcc cross-compiles.

    python -m deeptreeattention_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["conv.hip", "conv_bf16.hip", "stage.hip", "heads.hip", "capi.hip", "capi_modules.hip", "preprocess.hip", "xchg.hip", "meta.hip"]
HEADERS = ["common.h", "kernels.h", os.path.join("..", "..", "include", "dta_hip.h")]
LIB = os.path.join(HERE, "libdta_hip.so")
# the developer library: the same sources with -DDTA_DEV_SWITCHES (common.h: dev_getenv) -- environment switches for
# same-box A/B runs of alternative launch plans.  The product library above contains no getenv at all.
DEV_LIB = os.path.join(HERE, "libdta_hip_dev.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_digest():
    """Hash of every source / header the library is built from: compiled into capi.hip (dta_build_id) so that
    measurement artifacts (profiles/*.json) can name the exact build they were taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES) + sorted(HEADERS):
        h.update(f.encode())
        h.update(open(os.path.normpath(os.path.join(CSRC, f)), "rb").read())
    return h.hexdigest()[:16]


def _uses_dev_switches(src):
    return "dev_getenv(" in open(os.path.join(CSRC, src)).read()


def build(force=False, verbose=True, dev=True):
    """Compile every source (in parallel) and link libdta_hip.so; with dev=True also libdta_hip_dev.so, which shares every
    object except those of the sources that read a developer switch (recompiled with -DDTA_DEV_SWITCHES)."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    digest = source_digest()
    idstamp = os.path.join(objdir, "build_id.txt")
    id_stale = not os.path.exists(idstamp) or open(idstamp).read() != digest
    objs, dev_objs, procs = [], [], []
    extra = os.environ.get("DTA_EXTRA_HIPCC_FLAGS", "").split()
    stamp = os.path.join(objdir, "flags.txt")   # objects built with other flags (developer -D switches) are stale
    if not os.path.exists(stamp) or open(stamp).read() != " ".join(FLAGS + extra):
        force = True
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        variants = [("", [])] + ([("_dev", ["-DDTA_DEV_SWITCHES"])] if dev and _uses_dev_switches(src) else [])
        for suffix, defs in variants:
            o = os.path.join(objdir, src.replace(".hip", suffix + ".o"))
            (dev_objs if suffix else objs).append(o)
            if force or _stale(o, [s] + hdrs) or (src == "capi.hip" and id_stale):
                cmd = [hipcc] + FLAGS + extra + defs + (['-DDTA_BUILD_ID="%s"' % digest] if src == "capi.hip" else []) + ["-c", s, "-o", o]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((src + suffix, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        if len(variants) == 1:
            dev_objs.append(objs[-1])
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} failed ---\n{out}\n")
    if failed:
        raise RuntimeError("hipcc failed")
    with open(stamp, "w") as f:
        f.write(" ".join(FLAGS + extra))
    with open(idstamp, "w") as f:
        f.write(digest)
    for lib, members in ((LIB, objs),) + (((DEV_LIB, dev_objs),) if dev else ()):
        if force or procs or _stale(lib, members):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + members
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, dev="--no-dev" not in sys.argv))
