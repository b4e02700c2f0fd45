"""Repository rules that keep the parity claims honest (no GPU needed)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _files(d, exts):
    out = []
    for base, dirs, files in os.walk(os.path.join(ROOT, d)):
        dirs[:] = [x for x in dirs if x not in ("__pycache__", "lib")]
        out += [os.path.join(base, f) for f in files if f.endswith(exts)]
    return out


def test_product_never_touches_the_oracle():
    """the product path must not import, link, call or execute anything under oracle/"""
    for f in _files("vulkansift_amd", (".py", ".c", ".h", ".hip")):
        txt = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f
        assert not re.search(r"#\s*include[^\n]*oracle", txt), f
        assert "liboracle" not in txt, f
        if f.endswith("vksift_sharded.c"):
            # the one dlopen of the product: RCCL, loaded lazily for the sharded matcher — every library name it tries is librccl
            names = re.findall(r'"([^"\n]*\.so[^"\n]*)"', txt)
            assert names and all("librccl" in n for n in names), (f, names)
        else:
            assert "dlopen" not in txt, f
        assert "/root/reference" not in txt, f


def test_no_cpu_fallback_in_binding():
    txt = open(os.path.join(ROOT, "vulkansift_amd", "api.py")).read()
    assert "raise RuntimeError" in txt and "There is no CPU fallback" in txt


def test_gpu_side_code_does_not_read_the_reference_tree():
    for f in ["bench.py", "__graft_entry__.py"] + [os.path.relpath(p, ROOT) for p in _files("tests", (".py",))]:
        if f.endswith("test_layout.py"):
            continue
        assert "/root/reference" not in open(os.path.join(ROOT, f)).read(), f


def test_no_compat_layers_in_kernels():
    for f in _files("vulkansift_amd/csrc", (".hip", ".h", ".c")):
        txt = open(f).read()
        for bad in ("__HIP_PLATFORM_AMD__", "__CUDACC__", "cuda_runtime", "hipify", "triton"):
            assert bad not in txt, (f, bad)


def test_required_layout_exists():
    for p in ("bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "oracle/sift_oracle.c", "oracle/Makefile",
              "include/vulkansift/vulkansift.h", "include/vulkansift/vulkansift_types.h", "include/vksift_hip.h", "include/vksift_ext.h",
              "tests/golden/make_golden.py", "profiles"):
        assert os.path.exists(os.path.join(ROOT, p)), p
    gi = open(os.path.join(ROOT, ".gitignore")).read()
    assert "oracle/_ref/" in gi and "*.so" in gi
