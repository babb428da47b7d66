"""The C ABI's structs, the oracle's constants and the oracle's function coverage against interface facts extracted from
the reference's own compiled compute shader (tests/golden/reference_spv_facts.json, made by tools/spv_facts.py from
assets/shaders/compute_pass.comp.spv in the authoring container: buffer-block member offsets and array strides, the
function set of the module = the live call graph, its float constant pool, work-group size, image format).

This pins what can be pinned against an upstream artefact without a Vulkan stack: byte layouts (drop-in compatibility of
include/rvpt_hip.h), constants, and that no live function is left unrestated.  It does not execute the shader."""
import json
import re
import struct
import subprocess
from pathlib import Path

import pytest

from _util import GOLDEN

ROOT = Path(__file__).resolve().parent.parent
FACTS = json.loads((GOLDEN / "reference_spv_facts.json").read_text())

C_STRUCT_OF = {"RenderSettings": ("rvpt_render_settings", None), "Camera": ("rvpt_camera_data", None),
               "BvhNodes": ("rvpt_bvh_node", "BvhNode"), "Triangles": ("rvpt_triangle", "Triangle"), "Materials": ("rvpt_material", "Material")}


def _members(block):
    m = FACTS["blocks"][block]["members"]
    if "element_members" in m[0]:
        return m[0]["element_members"], m[0]["array_stride"]
    return m, None


def test_abi_structs_have_the_shader_s_byte_layout(tmp_path):
    lines = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{ROOT / "include" / "rvpt_hip.h"}"', "int main(void) {"]
    expect = {}
    for block, (cstruct, _) in C_STRUCT_OF.items():
        members, stride = _members(block)
        for m in members:
            lines.append(f'  printf("{cstruct}.{m["name"]} %zu\\n", offsetof({cstruct}, {m["name"]}));')
            expect[f"{cstruct}.{m['name']}"] = m["offset"]
        lines.append(f'  printf("sizeof.{cstruct} %zu\\n", sizeof({cstruct}));')
        if stride is not None:
            expect[f"sizeof.{cstruct}"] = stride
    expect["sizeof.rvpt_render_settings"] = 40  # last member (vec2 at 32) + 8
    expect["sizeof.rvpt_camera_data"] = 80      # mat4 at 0, vec4 at 64
    lines.append("  return 0; }")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for key, value in expect.items():
        assert int(got[key]) == value, (key, got[key], value)
    # the dead `random_source` block is deliberately not part of the ABI
    assert set(FACTS["blocks"]) - set(C_STRUCT_OF) == {"Random"}


def test_python_settings_block_matches_the_shader_s_uniform_block():
    from rvpt_amd import RenderSettings
    rs = RenderSettings(max_bounces=11, aa=12, current_frame=13, camera_mode=14, top_left_render_mode=15, top_right_render_mode=16,
                        bottom_left_render_mode=17, bottom_right_render_mode=18, split_ratio=(0.25, 0.75))
    raw = rs.pack().tobytes()
    assert len(raw) == 40
    for m in FACTS["blocks"]["RenderSettings"]["members"]:
        if m["name"] == "split_ratio":
            assert struct.unpack_from("<2f", raw, m["offset"]) == (0.25, 0.75)
        else:
            assert struct.unpack_from("<i", raw, m["offset"])[0] == getattr(rs, m["name"])


def test_tile_and_image_format():
    from rvpt_amd import native
    assert FACTS["local_size"] == [native.TILE, native.TILE, 1]              # rvpt.cpp:1035-1036 dispatches W/16 x H/16 groups
    assert {i["format"] for i in FACTS["images"].values()} == {"Rgba8"}     # the format RVPT_HIP_ACCUM_UNORM8 / read(RGBA8) reproduce


def test_oracle_constants_are_the_shader_s():
    pool = {struct.pack("<f", c) for c in FACTS["float_constants"]}
    src = (ROOT / "oracle" / "rvpt_oracle.c").read_text()
    defines = dict(re.findall(r"#define (O_[A-Z_]+) ([0-9.eE+-]+)f", src))
    for name in ("O_PI", "O_TWO_PI", "O_INV_PI", "O_EPSILON"):  # compute_pass.comp:5-12 (the other O_* are sincos's own)
        assert struct.pack("<f", float(defines[name])) in pool, (name, defines[name])
    for literal in (0.5, 0.2, 0.3, 0.7, 0.1, 31.0, 4294967296.0):  # sky gradient, ambient term, Hart's steps, rand()'s 2^32
        assert struct.pack("<f", literal) in pool


# live function of the compiled shader -> where the oracle restates it ("main" = oracle_render's pixel loop)
RESTATED_AS = {
    "camera_ortho_ray": "o_ortho_ray", "camera_pinhole_ray": "o_pinhole_ray", "camera_spherical_ray": "o_spherical_ray",
    "get_camera_ray": "o_camera_ray", "unit_spherical_to_cartesian": "o_unit_spherical", "convert_old_material": "o_scene_hit",
    "distance_triangle": "o_distance_triangle", "dot2": "o_edge_dist2", "eval_integrator": "o_integrator", "frensel_reflectance": "o_fresnel",
    "integrator_Appel": "o_integrator", "integrator_Cook": "o_integrator", "integrator_Hart": "o_hart", "integrator_Kajiya": "o_kajiya",
    "integrator_Utah": "o_integrator", "integrator_Whitted": "o_integrator", "integrator_ao": "o_integrator", "integrator_binary": "o_integrator",
    "integrator_color": "o_integrator", "integrator_depth": "o_integrator", "integrator_normal": "o_integrator",
    "intersect_aabb": "o_aabb_test", "intersect_bvh": "o_closest_hit", "intersect_bvh_any": "o_scene_any", "intersect_scene": "o_scene_hit",
    "intersect_scene_any": "o_scene_any", "intersect_scene_st": "o_hart", "intersect_triangle_fast": "o_tri_test", "main": "oracle_render",
    "map_cosine_hemisphere_simple": "o_kajiya", "map_uniform_sphere": "o_map_uniform_sphere", "mat_eval_Lambert_cos": "o_kajiya",
    "mat_eval_dielectric": "o_kajiya", "mat_eval_mirror": "o_kajiya", "mat_scatter_Lambert_cos": "o_kajiya", "min_idx": "o_hart",
    "rand": "o_rand", "rand_xorshift": "o_rand", "wang_hash": "o_wang_hash",
}


def test_every_live_shader_function_is_restated_by_the_oracle():
    live = {f.split("(")[0] for f in FACTS["functions"]}
    assert live == set(RESTATED_AS)
    src = (ROOT / "oracle" / "rvpt_oracle.c").read_text()
    for glsl, c_name in RESTATED_AS.items():
        assert re.search(rf"\b{c_name}\(", src), (glsl, c_name)
