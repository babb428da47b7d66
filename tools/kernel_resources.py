#!/usr/bin/env python3
"""Register / scratch / static-LDS use of every kernel, read from the compiler's own metadata (hipcc -S of the kernel
file with the library's flags) -> profiles/<round>_kernel_resources.json.

Why: rocprofv3's counter-collection CSV reports `VGPR_Count` in its own units (half the architectural count for these
wave64 kernels: 40 for an 80-register kernel) and `LDS_Block_Size` = the STATIC group segment only (0 here: the kernels use
dynamic LDS, sized per launch by rvpt_abi.hip::choose_launch and reported by rvpt_hip_get_launch_info).  This file is the
authoritative register count the occupancy statements in DESIGN.md / profiles/README.md refer to.

usage: tools/kernel_resources.py <round>
"""
import json
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from rvpt_amd import build  # noqa: E402


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
    flags = [f for f in build.FLAGS if f not in ("-fPIC", "-shared")]
    text = ""
    with tempfile.TemporaryDirectory() as d:
        for src in ("rvpt_kernels.hip", "rvpt_packets.hip", "rvpt_bvh4.hip", "rvpt_bvh8.hip"):
            out = Path(d) / "k.s"
            subprocess.run([build.hipcc(), *flags, "-S", "--cuda-device-only", str(ROOT / "rvpt_amd" / "csrc" / src), "-o", str(out)],
                           check=True, capture_output=True)
            text += out.read_text()
    demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    res = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        body = m.group(2)
        g = lambda key: int(re.search(rf"\.amdhsa_{key} (\d+)", body).group(1))
        vg = g("next_free_vgpr")
        res[demangle(m.group(1)).replace("(rv::FrameParams)", "").replace("rv::", "").replace("void ", "")] = {
            "vgpr": vg, "sgpr": g("next_free_sgpr"), "scratch_bytes": g("private_segment_fixed_size"), "static_lds_bytes": g("group_segment_fixed_size"),
            "waves_per_simd_by_vgpr": min(8, 512 // (((vg + 7) // 8) * 8))}
    dst = ROOT / "profiles" / f"{rnd}_kernel_resources.json"
    dst.write_text(json.dumps({"flags": flags, "kernels": res}, indent=1) + "\n")
    for k, v in res.items():
        print(f"{k[:72]:72s} {v}")


if __name__ == "__main__":
    main()
