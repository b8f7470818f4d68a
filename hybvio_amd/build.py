"""In-tree build of libhybvio_hip.so (hipcc cross-compiles gfx950 without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libhybvio_hip.so")

# -ffp-contract=off: the LK float sequence must match the oracle bit for bit (no FMA fusion).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall"]


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(PKG, "..", "include", "hybvio_hip.h")]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build_hip(force: bool = False, verbose: bool = False, lib: str | None = None) -> str:
    """One object per source (compiled in parallel, rebuilt only when the source or a header changed), then one link.
    lib: a second build beside the default one (A/B runs through HV_LIB_OVERRIDE): its objects live in their own directory."""
    lib = lib or LIB
    if not force and lib == LIB and not _stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" if lib == LIB else "obj_" + os.path.splitext(os.path.basename(lib))[0])
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = ["-DHV_EKF_PHASE_STAMPS"] if os.environ.get("HV_EKF_PHASE_STAMPS") == "1" else []
    extra += ["-D" + d for d in os.environ.get("HV_EXTRA_DEFINES", "").split() if d]        # developer experiments
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + extra
    tag = os.path.join(objdir, "flags.txt")                                                  # other flags: every object is stale
    tag_text = " ".join(flags)
    same_flags = os.path.exists(tag) and open(tag).read() == tag_text
    headers = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(PKG, "..", "include", "hybvio_hip.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        fresh = same_flags and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), newest_header)
        if not fresh or (force and os.environ.get("HV_BUILD_REUSE_OBJECTS") != "1"):
            cmd = [hipcc] + flags + ["-c", "-o", obj, src]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, sources()))
    with open(tag, "w") as f:
        f.write(tag_text)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return lib


HOST = os.path.join(PKG, "host")
HOST_LIB = os.path.join(LIBDIR, "libhybvio_host.so")
HOST_TEST = os.path.join(LIBDIR, "test_host_adapters")


def build_host(force: bool = False, verbose: bool = False) -> tuple[str, str]:
    """C++ adapters (pure C-ABI client: plain g++, no HIP headers) and their test executable."""
    build_hip()
    srcs = sorted(glob.glob(os.path.join(HOST, "*.cpp")))
    test_src = os.path.join(PKG, "..", "tests", "cpp", "test_host_adapters.cpp")
    deps = srcs + glob.glob(os.path.join(HOST, "*.hpp")) + [test_src, LIB]
    fresh = all(os.path.exists(f) for f in (HOST_LIB, HOST_TEST)) and \
        all(os.path.getmtime(d) <= min(os.path.getmtime(HOST_LIB), os.path.getmtime(HOST_TEST)) for d in deps)
    if fresh and not force:
        return HOST_LIB, HOST_TEST
    cxx = os.environ.get("CXX", "g++")
    common = ["-std=c++17", "-O2", "-Wall", "-Wextra", "-fPIC"]
    link = ["-L" + LIBDIR, "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    cmds = [[cxx] + common + ["-shared", "-o", HOST_LIB] + srcs + link + ["-lhybvio_hip"],
            [cxx] + common + ["-o", HOST_TEST, test_src] + link + ["-lhybvio_host", "-lhybvio_hip"]]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HOST_LIB, HOST_TEST


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
