"""Register budget of the per-keypoint kernels (CPU: hipcc cross-compiles). k_descriptor's two-wave instantiations are pinned to eight
waves per SIMD (amdgpu_waves_per_eu(8, 8): 64 VGPRs) because the kernel is bound by VALU issue and loses 3 % at six waves (DESIGN.md
section 8); a source change that no longer fits shows up as scratch — 8 bytes of it cost 4 % — long before any result changes."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_descriptor_and_orientation_kernels_keep_their_registers():
    import descriptor_floor as df
    df.kernel_isa()  # compiles features.hip with the build's own flags to /tmp/_features_floor.s
    txt = open("/tmp/_features_floor.s").read()
    meta = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", txt)
    seen = 0
    for name, scratch, vgpr in meta:
        if "k_descriptorILi2" in name and "ELb0EEEv5MultiINS_8FeatArgsEE" in name:  # fp32 planes
            assert int(scratch) == 0 and int(vgpr) <= 64, (name, scratch, vgpr)
            seen += 1
        elif "k_descriptorILi" in name or "13k_orientationI" in name:
            # (the binary16 instantiations are NOT pinned: their eight 2-byte tap loads per iteration need the registers to stay in flight —
            # 64 VGPRs cost that mode 17 % of the kernel, 7.12 vs 5.94 ms per 512 frames)
            assert int(scratch) == 0 and int(vgpr) <= 80, (name, scratch, vgpr)
            seen += 1
    assert seen == 12
