"""Native clients of the public headers: the drop-in claim at the C / C++ level (not only through the ctypes mirror)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NATIVE = os.path.join(ROOT, "tests", "native")
LIBDIR = os.path.join(ROOT, "vulkansift_amd", "lib")


def _build(src, out, cxx=False):
    cmd = ["g++" if cxx else "gcc", "-O1", "-std=c++17" if cxx else "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(NATIVE, src), "-o", out,
           "-L" + LIBDIR, "-lvulkansift", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def test_native_clients_compile_and_link(vk, tmp_path):
    """gcc/g++ accept the public headers and every symbol the clients use resolves in libvulkansift.so (no GPU needed)"""
    _build("client_match.c", str(tmp_path / "client_match"))
    _build("client_errors.cpp", str(tmp_path / "client_errors"), cxx=True)


def _fnv(b):
    h = 1469598103934665603
    for x in b:
        h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.gpu
def test_c_client_equals_python_mirror(vk, tmp_path):
    exe = _build("client_match.c", str(tmp_path / "client_match"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    head, dig = lines[-2].split(), lines[-1].split()
    img1, img2 = vk.gen_synthetic_image(101, 320, 240), vk.gen_synthetic_image(102, 320, 240)
    with vk.Instance(vk.default_config(input_image_max_size=320 * 240)) as inst:
        inst.detectFeatures(img1, 0)
        inst.detectFeatures(img2, 1)
        f1, f2 = inst.downloadFeatures(0), inst.downloadFeatures(1)
        inst.matchFeatures(0, 1)
        m12 = inst.downloadMatches()
        inst.matchFeatures(1, 0)
        m21 = inst.downloadMatches()
    assert int(head[1]) == len(f1) and int(head[2]) == len(f2) and int(head[4]) == len(m12) and int(head[5]) == len(m21)
    assert int(head[7]) > 0                                   # the reference example's CPU filter keeps some matches
    got = [int(x, 16) for x in dig[1:5]]
    want = [_fnv(f1.tobytes()), _fnv(f2.tobytes()), _fnv(m12.tobytes()), _fnv(m21.tobytes())]
    assert got == want


@pytest.mark.gpu
def test_cxx_client_exceptions_cross_the_c_frames(vk, tmp_path):
    exe = _build("client_errors.cpp", str(tmp_path / "client_errors"), cxx=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok 3 caught 4" in r.stdout
