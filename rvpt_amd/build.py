"""Build the native library (hipcc, gfx950) in-tree: rvpt_amd/librvpt_hip.so."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "librvpt_hip.so"
SOURCES = [_PKG / "csrc" / n for n in ("rvpt_kernels.hip", "rvpt_packets.hip", "rvpt_bvh4.hip", "rvpt_abi.hip", "bvh_builder.cpp", "bvh_wide.cpp")]
LAB_SOURCES = SOURCES + [_PKG / "csrc" / "rvpt_bvh8.hip"]  # the laboratory build (librvpt_hip_debug.so) also carries the walks that measured slower
HEADERS = [_PKG / "csrc" / "rvpt_kernels.h", _PKG / "csrc" / "rvpt_packets.h", _PKG / "csrc" / "rvpt_early_out.h", _PKG / "csrc" / "rvpt_device.h", _PKG / "csrc" / "rvpt_math.h", _PKG / "csrc" / "rvpt_rect.h",
           _PKG / "csrc" / "rvpt_vis.h", _PKG.parent / "include" / "rvpt_hip.h", _PKG.parent / "include" / "rvpt_hip_lab.h"]

# -ffp-contract=off: the arithmetic specification fixes where FMAs happen; applies to the
# device code and to the few host-side evaluations (tan of the half field of view) alike.
# -fno-slp-vectorize: hipcc otherwise pairs the scalar f32 ops of the intersect loop into v_pk_*_f32, which
# run at half rate on gfx950 and scramble the LDS reads into ds_read2_b32 (measured -15 %, profiles/).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function"]


KERNEL_SOURCES = [_PKG / "csrc" / n for n in ("rvpt_kernels.hip", "rvpt_packets.hip", "rvpt_bvh4.hip", "rvpt_packets.h", "rvpt_early_out.h",
                                              "rvpt_device.h", "rvpt_kernels.h", "rvpt_math.h", "rvpt_rect.h", "rvpt_vis.h")]


def _code_only(text: str) -> bytes:
    """A source file without comments and without layout: what the compiler sees of it, near enough.  (A comment edit must not
    invalidate a profile; a code edit must.)"""
    import re
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return " ".join(text.split()).encode()


def kernel_sha() -> str:
    """Identity of the device code: sha256 over the kernel sources with comments and layout stripped (not the ABI layer).  tools/summarize_prof.py stamps it on every replayable profile
    figure (profiles/pmc_traffic.json); bench.py replays a figure only while it still matches."""
    import hashlib
    h = hashlib.sha256()
    for p in KERNEL_SOURCES:
        if p.exists():
            h.update(p.name.encode() + b"\0" + _code_only(p.read_text()) + b"\0")
    return h.hexdigest()[:16]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (need ROCm's hipcc to build rvpt_amd/librvpt_hip.so)")


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS)


def build_native(force: bool = False, verbose: bool = False) -> Path:
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc(), *FLAGS, *map(str, SOURCES), "-o", str(LIB_PATH)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    if verbose:
        print(" ".join(cmd))
    return LIB_PATH


DEBUG_LIB_PATH = _PKG / "librvpt_hip_debug.so"


def build_native_debug(force: bool = False) -> Path:
    """The LABORATORY build of the library (include/rvpt_hip_lab.h): the same sources with -DRVPT_HIP_LAB=1 — the selftests, the host-side forms of the device
    data, the opt-in walks that measured slower (rvpt_bvh8.hip, trace_bvh4q) and the tuning knobs — and with the kernels' internal checks compiled in
    (-DRV_REPORT_STACK_OVERFLOW=1: a BVH traversal that pushes past the stack the host sized sets an error word, which rvpt_hip_wait reports under RVPT_HIP_DEBUG=1
    — a few percent of the walk's throughput, hence not in the release build).  native.load_lab() / Context(lab=True) / RVPT_HIP_LAB=1 select it."""
    if not force and DEBUG_LIB_PATH.exists() and DEBUG_LIB_PATH.stat().st_mtime >= max(p.stat().st_mtime for p in LAB_SOURCES + HEADERS):
        return DEBUG_LIB_PATH
    cmd = [hipcc(), *FLAGS, "-DRVPT_HIP_LAB=1", "-DRV_REPORT_STACK_OVERFLOW=1", *map(str, LAB_SOURCES), "-o", str(DEBUG_LIB_PATH)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    return DEBUG_LIB_PATH


HOST_DIR = _PKG / "host"
HOST_BIN_DIR = _PKG / "bin"  # git-ignored build outputs (travel to the GPU box with the snapshot)
HOST_TARGETS = {"rvpt_render": ["render_main.cpp", "rvpt_host.cpp"], "host_selftest": ["host_selftest.cpp", "rvpt_host.cpp"]}


def build_host(force: bool = False) -> Path:
    """Compile the C++ host layer (rvpt_amd/host/: the mirror of the reference's class RVPT above the C ABI) with
    g++ and link it against the in-tree librvpt_hip.so: the headless CLI `rvpt_render` and the GPU-free `host_selftest`."""
    build_native()
    HOST_BIN_DIR.mkdir(exist_ok=True)
    srcs = list(HOST_DIR.glob("*.cpp")) + list(HOST_DIR.glob("*.h")) + [_PKG.parent / "include" / "rvpt_hip.h"]
    newest = max(p.stat().st_mtime for p in srcs + [LIB_PATH])
    for name, files in HOST_TARGETS.items():
        out = HOST_BIN_DIR / name
        if not force and out.exists() and out.stat().st_mtime >= newest:
            continue
        cmd = [shutil.which("g++") or "g++", "-O2", "-std=c++17", "-Wall", "-Wextra", *[str(HOST_DIR / f) for f in files], "-o", str(out),
               f"-L{_PKG}", "-lrvpt_hip", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("g++ failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    return HOST_BIN_DIR


if __name__ == "__main__":
    print(build_native(force=True, verbose=True))
    print(build_native_debug(force=True))
    print(build_host(force=True))
