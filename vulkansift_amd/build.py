"""Build libvulkansift.so in-tree: gcc for the C host, hipcc (gfx950) for the kernels.

    python -m vulkansift_amd.build [--force] [--sanitize]

--sanitize builds a second library, lib/libvulkansift_asan.so, whose C host is compiled with
-fsanitize=address,undefined (the reference's VKSIFT_SANITIZE option, CMakeLists.txt:30-31,91-97); the HIP
objects are shared with the normal build. Load it with VKSIFT_LIB=<path> and
LD_PRELOAD=$(gcc -print-file-name=libasan.so) (the sanitizer runtime has to come first in the process).

The shared library lands in vulkansift_amd/lib/ (git-ignored, but shipped to the GPU box by gpurun).
hipcc cross-compiles gfx950 without a GPU, so this also serves as the "does it build" check.
"""
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OUT_DIR = os.path.join(PKG, "lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB_PATH = os.path.join(OUT_DIR, "libvulkansift.so")

HOST_SRCS = ["host/vksift_api.c", "host/vksift_instance.c", "host/vksift_detect.c", "host/vksift_buffers.c", "host/vksift_match.c", "host/vksift_ext.c", "host/vksift_sharded.c",
             "host/vksift_hostmath.c", "host/vksift_log.c", "host/vksift_synth.c"]
HIP_SRCS = ["hip/runtime.hip", "hip/pyramid.hip", "hip/extrema.hip", "hip/features.hip", "hip/match.hip", "hip/records.hip"]

ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
INCLUDES = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(CSRC, "host"), "-I" + CSRC]

# -ffp-contract=off: fused multiply-adds are spelled fmaf() in the sources; nothing else may be
# contracted, so the kernels stay bit-identical to the CPU oracle (see csrc/detmath.h).
CFLAGS = ["-O2", "-std=gnu11", "-fPIC", "-fexceptions", "-ffp-contract=off", "-fno-fast-math", "-mavx2", "-mfma", "-Wall", "-Wextra",
          "-Wno-unused-parameter"]
HIPFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
            "-fno-gpu-rdc"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


_TOOL_VERSION = {}


def _tool_version(tool):
    if tool not in _TOOL_VERSION:
        r = subprocess.run([tool, "--version"], capture_output=True, text=True)
        _TOOL_VERSION[tool] = (r.stdout + r.stderr).strip()
    return _TOOL_VERSION[tool]


def _signature(cmd, src, headers):
    """What an object file was built FROM: the full command line (every flag), the compiler's version banner, the bytes of the source
    and of every header it may include. Kept in a sidecar next to the object (<obj>.sig); an object is rebuilt when the sidecar
    differs — a changed flag reaches the object file even when no time stamp moved (round 3: -fno-slp-vectorize did not reach
    pyramid.hip.o until a header happened to change)."""
    h = hashlib.sha256()
    # paths relative to the checkout: the same tree under another root (the GPU box's copy) has the same signature
    h.update(("\0".join(cmd).replace(ROOT, "<root>") + "\0" + _tool_version(cmd[0])).encode())
    for path in [src] + sorted(headers):
        h.update(os.path.relpath(path, ROOT).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(obj, sig):
    try:
        return not os.path.exists(obj) or open(obj + ".sig").read().strip() != sig
    except OSError:
        return True


def _compile(cmd, src, obj, headers, force, verbose, tag):
    sig = _signature(cmd, src, headers)
    if force or _stale(obj, sig):
        if verbose:
            print(tag, os.path.relpath(src, CSRC))
        if os.path.exists(obj + ".sig"):
            os.remove(obj + ".sig")
        _run(cmd)
        with open(obj + ".sig", "w") as f:
            f.write(sig + "\n")
        return True
    return False


def _all_headers():
    hs = []
    for d in (os.path.join(ROOT, "include"), CSRC):
        for base, _, files in os.walk(d):
            hs += [os.path.join(base, f) for f in files if f.endswith(".h")]
    return hs


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError("build step failed: " + os.path.basename(cmd[-1]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)


# Per-file code generation options. max-ilp: the default scheduler serialises the independent accumulator chains of the
# unrolled filters to save registers, which leaves an s_nop after almost every dependent packed-fp32 pair.
HIP_EXTRA = {
    # MFMA results straight into VGPRs: the matcher's epilogue reads every accumulator element once, and the default AGPR
    # form costs one v_accvgpr_read per element (2 of ~18 VALU instructions per MFMA) for nothing — it has registers to spare
    "hip/match.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
    # no SLP vectoriser: it packs the atan2 / exp polynomials of the two samples a descriptor iteration handles into v_pk_fma_f32
    # chains — each packed op issues like two scalar ones AND waits for its predecessor (s_nop after every one), where the two
    # scalar chains interleave without wait states. Same operations, same results; descriptor kernel -5.6 % (1.98 -> 1.87 ms)
    "hip/features.hip": ["-fno-slp-vectorize"],
    # the same for the blur kernels, whose unrolled filters are written with explicit two-row interleaving already: 128 x 640x480
    # detect + match +1.8 % (the whole detection 5.33 -> 5.01 ms per call, most of it in the coarse octaves' launches)
    "hip/pyramid.hip": ["-fno-slp-vectorize"],
}


def _extra_flags(src):
    """VKSIFT_SCHED="pyramid=max-ilp,features=max-ilp" overrides HIP_EXTRA for experiments."""
    spec = os.environ.get("VKSIFT_SCHED")
    if spec is None:
        return HIP_EXTRA.get(src, [])
    for item in spec.split(","):
        if "=" in item:
            name, strat = item.split("=", 1)
            if os.path.basename(src).split(".")[0] == name and strat:
                return ["-mllvm", "-amdgpu-sched-strategy=" + strat]
    return []


SANITIZE_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
ASAN_LIB_PATH = os.path.join(OUT_DIR, "libvulkansift_asan.so")


def build(force=False, verbose=False, sanitize=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = _all_headers()
    objs = []
    rebuilt = False
    for src in HOST_SRCS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, os.path.basename(src) + (".asan.o" if sanitize else ".o"))
        objs.append(o)
        rebuilt |= _compile(["gcc"] + CFLAGS + (SANITIZE_FLAGS if sanitize else []) + INCLUDES + ["-DVKSIFT_BUILD", "-c", s, "-o", o], s, o, headers, force, verbose, "[cc ]")
    for src in HIP_SRCS:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(o)
        rebuilt |= _compile([HIPCC] + HIPFLAGS + _extra_flags(src) + INCLUDES + ["-c", s, "-o", o], s, o, headers, force, verbose, "[hip]")
    out = ASAN_LIB_PATH if sanitize else LIB_PATH
    if force or rebuilt or _newer(out, objs):
        if verbose:
            print("[ld ]", os.path.relpath(out, ROOT))
        san = []
        if sanitize:
            # gcc compiled the instrumented objects, so gcc's runtimes serve them: libubsan linked here, libasan preloaded by the user
            san = [subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()]
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + san +
             ["-L" + os.path.join(ROCM, "lib"), "-lroctx64", "-lm", "-ldl", "-lpthread", "-Wl,-rpath," + os.path.join(ROCM, "lib")])
    return out


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, verbose=True, sanitize="--sanitize" in sys.argv)
    print(p)
