"""Install the built library the way the reference's `make install` does (CMakeLists.txt:250-278):

    python -m vulkansift_amd.install --prefix /opt/vulkansift

    <prefix>/include/vulkansift/*.h, vksift_ext.h, vksift_hip.h
    <prefix>/lib/libvulkansift.so
    <prefix>/lib/cmake/VulkanSift/VulkanSiftConfig.cmake, VulkanSiftConfigVersion.cmake

so that an application of the reference relinks unchanged: find_package(VulkanSift) with CMAKE_PREFIX_PATH=<prefix>, then
target_link_libraries(app ${VulkanSift_LIB}) and #include <vulkansift/vulkansift.h>.
"""
import argparse
import os
import shutil

from . import build as vbuild

VERSION = "0.1"   # the reference's package version


def install(prefix):
    lib = vbuild.build()
    prefix = os.path.abspath(prefix)
    inc_src = os.path.join(vbuild.ROOT, "include")
    inc_dst = os.path.join(prefix, "include")
    shutil.copytree(inc_src, inc_dst, dirs_exist_ok=True)
    os.makedirs(os.path.join(prefix, "lib"), exist_ok=True)
    shutil.copy2(lib, os.path.join(prefix, "lib", "libvulkansift.so"))
    cm_dst = os.path.join(prefix, "lib", "cmake", "VulkanSift")
    os.makedirs(cm_dst, exist_ok=True)
    for name in ("VulkanSiftConfig.cmake", "VulkanSiftConfigVersion.cmake"):
        text = open(os.path.join(vbuild.ROOT, "cmake", name + ".in")).read().replace("@VERSION@", VERSION)
        with open(os.path.join(cm_dst, name), "w") as f:
            f.write(text)
    return prefix


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--prefix", required=True)
    print(install(ap.parse_args().prefix))
