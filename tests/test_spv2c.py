"""tools/spv2c.py on a hand-assembled SPIR-V module whose results are known in closed form: the constructs the reference's
shader is built from — OpPhi on the edges of a short-circuit `&&`, a structured loop with a back edge, OpSwitch, pointer
parameters (Function storage), access chains into a runtime array of a buffer block, vector shuffle / construct / extract,
OpDot through the shim, OpSelect, unsigned wrap-around, OpConvert*, GLSL.std.450 FMax — translated, compiled with gcc and
executed.  Independent of the oracle: what is checked here is the translator, the tool that pins the oracle.
Needs only gcc (no reference tree, no GPU)."""
import ctypes as C
import struct
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import spv2c  # noqa: E402


class Asm:
    """Minimal SPIR-V assembler: ids are handed out by new(); emit(opcode, *operands) appends one instruction."""

    def __init__(self):
        self.words, self.bound = [], 1

    def new(self):
        self.bound += 1
        return self.bound - 1

    def emit(self, op, *operands):
        ws = []
        for o in operands:
            if isinstance(o, str):
                b = o.encode() + b"\0"
                b += b"\0" * (-len(b) % 4)
                ws += list(struct.unpack("<%dI" % (len(b) // 4), b))
            elif isinstance(o, float):
                ws.append(struct.unpack("<I", struct.pack("<f", o))[0])
            else:
                ws.append(int(o) & 0xFFFFFFFF)
        self.words += [((len(ws) + 1) << 16) | op] + ws

    def binary(self):
        return struct.pack("<%dI" % (5 + len(self.words)), 0x07230203, 0x00010000, 0, self.bound, 0, *self.words)


def build_module():
    a = Asm()
    n = a.new
    glsl, main, f_and, f_loop, f_switch = n(), n(), n(), n(), n()
    void, bool_, u32, i32, f32, v2, v3, uv3 = n(), n(), n(), n(), n(), n(), n(), n()
    fn_void, fn_and, fn_loop, fn_switch = n(), n(), n(), n()
    p_f_func, p_i_func, p_v3_func = n(), n(), n()
    rt_arr, block, p_block, p_f_buf, buf = n(), n(), n(), n(), n()
    p_uv3_in, gid, p_u_in = n(), n(), n()
    c0f, c1f, c2f, c05, c0i, c1i, c2i, c3i, c4u, c0u, c1u, c2u, c3u, cbig = (n() for _ in range(14))
    a.emit(17, 1)                                  # OpCapability Shader
    a.emit(11, glsl, "GLSL.std.450")
    a.emit(14, 0, 1)                               # OpMemoryModel Logical GLSL450
    a.emit(15, 5, main, "main", gid)               # OpEntryPoint GLCompute
    a.emit(16, main, 17, 8, 1, 1)                  # LocalSize 8 1 1
    for i, s in ((main, "main"), (f_and, "f_and(f1;f1;"), (f_loop, "f_loop(i1;"), (f_switch, "f_switch(i1;vf3;"), (buf, "buf"), (gid, "gl_GlobalInvocationID")):
        a.emit(5, i, s)
    a.emit(71, rt_arr, 6, 4)                       # ArrayStride 4
    a.emit(72, block, 0, 35, 0)                    # member 0 Offset 0
    a.emit(71, block, 3)                           # BufferBlock
    a.emit(71, buf, 34, 0)                         # DescriptorSet 0
    a.emit(71, buf, 33, 0)                         # Binding 0
    a.emit(71, gid, 11, 28)                        # BuiltIn GlobalInvocationId
    a.emit(19, void); a.emit(20, bool_); a.emit(21, u32, 32, 0); a.emit(21, i32, 32, 1); a.emit(22, f32, 32)
    a.emit(23, v2, f32, 2); a.emit(23, v3, f32, 3); a.emit(23, uv3, u32, 3)
    a.emit(32, p_f_func, 7, f32); a.emit(32, p_i_func, 7, i32); a.emit(32, p_v3_func, 7, v3)
    a.emit(33, fn_void, void); a.emit(33, fn_and, f32, p_f_func, p_f_func); a.emit(33, fn_loop, f32, p_i_func); a.emit(33, fn_switch, f32, p_i_func, p_v3_func)
    a.emit(29, rt_arr, f32); a.emit(30, block, rt_arr); a.emit(32, p_block, 2, block); a.emit(32, p_f_buf, 2, f32)
    a.emit(59, p_block, buf, 2)
    a.emit(32, p_uv3_in, 1, uv3); a.emit(59, p_uv3_in, gid, 1); a.emit(32, p_u_in, 1, u32)
    for c, v in ((c0f, 0.0), (c1f, 1.0), (c2f, 2.0), (c05, 0.5)):
        a.emit(43, f32, c, v)
    for c, v in ((c0i, 0), (c1i, 1), (c2i, 2), (c3i, 3)):
        a.emit(43, i32, c, v)
    for c, v in ((c4u, 4), (c0u, 0), (c1u, 1), (c2u, 2), (c3u, 3), (cbig, 0x9E3779B9)):
        a.emit(43, u32, c, v)

    # float f_and(float* x, float* y): (x > 0 && y > 0) ? x + y : -1   — the && is an OpPhi over two edges
    px, py = n(), n()
    l0, l1, l2, l3, l4 = n(), n(), n(), n(), n()
    x, y, cx, cy, phi, s, neg = n(), n(), n(), n(), n(), n(), n()
    a.emit(54, f32, f_and, 0, fn_and); a.emit(55, p_f_func, px); a.emit(55, p_f_func, py)
    a.emit(248, l0); a.emit(61, f32, x, px); a.emit(61, f32, y, py); a.emit(186, bool_, cx, x, c0f)
    a.emit(247, l2, 0); a.emit(250, cx, l1, l2)
    a.emit(248, l1); a.emit(186, bool_, cy, y, c0f); a.emit(249, l2)
    a.emit(248, l2); a.emit(245, bool_, phi, cx, l0, cy, l1)
    a.emit(247, l4, 0); a.emit(250, phi, l3, l4)
    a.emit(248, l3); a.emit(129, f32, s, x, y); a.emit(254, s)
    a.emit(248, l4); a.emit(127, f32, neg, c1f); a.emit(254, neg)
    a.emit(56)

    # float f_loop(int* n): sum_{i<n} i*i*0.5 with a Function variable as the counter (header / body / continue / merge)
    pn = n()
    vi, vs = n(), n()
    h, b, cont, m, e, chk = n(), n(), n(), n(), n(), n()
    iv, nv, c, fi, sq, hv, s0, s1, i1, r = (n() for _ in range(10))
    a.emit(54, f32, f_loop, 0, fn_loop); a.emit(55, p_i_func, pn)
    a.emit(248, e); a.emit(59, p_i_func, vi, 7); a.emit(59, p_f_func, vs, 7)
    a.emit(62, vi, c0i); a.emit(62, vs, c0f); a.emit(249, h)
    a.emit(248, h); a.emit(246, m, cont, 0); a.emit(249, chk)
    a.emit(248, chk); a.emit(61, i32, iv, vi); a.emit(61, i32, nv, pn); a.emit(177, bool_, c, iv, nv); a.emit(250, c, b, m)
    a.emit(248, b); a.emit(111, f32, fi, iv); a.emit(133, f32, sq, fi, fi); a.emit(133, f32, hv, sq, c05)
    a.emit(61, f32, s0, vs); a.emit(129, f32, s1, s0, hv); a.emit(62, vs, s1); a.emit(249, cont)
    a.emit(248, cont); a.emit(128, i32, i1, iv, c1i); a.emit(62, vi, i1); a.emit(249, h)
    a.emit(248, m); a.emit(61, f32, r, vs); a.emit(254, r)
    a.emit(56)

    # float f_switch(int* k, vec3* v): 0 -> v.x, 1 -> dot(v, v.zyx), 2 -> max(v.y, 2) * 2, default -> float(uint(k) * 0x9E3779B9 >> 16)
    pk, pv = n(), n()
    e2, k0, k1, k2, kd, mg = n(), n(), n(), n(), n(), n()
    kv, vv, r0, sh, r1, vy, mx, r2, ku, mu, shf, rd, out = (n() for _ in range(13))
    vres = n()
    a.emit(54, f32, f_switch, 0, fn_switch); a.emit(55, p_i_func, pk); a.emit(55, p_v3_func, pv)
    a.emit(248, e2); a.emit(59, p_f_func, vres, 7); a.emit(61, i32, kv, pk); a.emit(61, v3, vv, pv)
    a.emit(247, mg, 0); a.emit(251, kv, kd, 0, k0, 1, k1, 2, k2)
    a.emit(248, k0); a.emit(81, f32, r0, vv, 0); a.emit(62, vres, r0); a.emit(249, mg)
    a.emit(248, k1); a.emit(79, v3, sh, vv, vv, 2, 1, 0); a.emit(148, f32, r1, vv, sh); a.emit(62, vres, r1); a.emit(249, mg)
    a.emit(248, k2); a.emit(81, f32, vy, vv, 1); a.emit(12, f32, mx, glsl, 40, vy, c2f); a.emit(133, f32, r2, mx, c2f); a.emit(62, vres, r2); a.emit(249, mg)
    a.emit(248, kd); a.emit(124, u32, ku, kv); a.emit(132, u32, mu, ku, cbig); a.emit(194, u32, shf, mu, n16 := n()); a.emit(112, f32, rd, shf); a.emit(62, vres, rd); a.emit(249, mg)
    a.emit(248, mg); a.emit(61, f32, out, vres); a.emit(254, out)
    a.emit(56)
    # the constant 16 used above has to be declared before the functions in a real module; the parser does not mind the
    # order of declarations, but keep it tidy: declare it now in the same section the tool scans (types/constants/globals)
    a.emit(43, u32, n16, 16)

    # void main(): g = gid.x; base = 4*g; a = buf[base], b = buf[base+1]; buf[base+2] = f_and(a,b) + f_loop(int(b));
    #             buf[base+3] = f_switch(int(a), vec3(a, b, a*b)) ; selects: if a == 0.5 exactly the and-result is replaced by 7
    em = n()
    pg, g, base, i1_, i2_, i3_, pa, pb, pc, pd, av, bv = (n() for _ in range(12))
    va, vb, vk, vvec, r_and, r_loop, r_sw, bi, ai, ab, vec, summ, is_half, c7, chosen = (n() for _ in range(15))
    a.emit(43, f32, c7, 7.0)
    a.emit(54, void, main, 0, fn_void)
    a.emit(248, em)
    a.emit(59, p_f_func, va, 7); a.emit(59, p_f_func, vb, 7); a.emit(59, p_i_func, vk, 7); a.emit(59, p_v3_func, vvec, 7)
    a.emit(65, p_u_in, pg, gid, c0u); a.emit(61, u32, g, pg); a.emit(132, u32, base, g, c4u)
    a.emit(128, u32, i1_, base, c1u); a.emit(128, u32, i2_, base, c2u); a.emit(128, u32, i3_, base, c3u)
    a.emit(65, p_f_buf, pa, buf, c0i, base); a.emit(65, p_f_buf, pb, buf, c0i, i1_); a.emit(65, p_f_buf, pc, buf, c0i, i2_); a.emit(65, p_f_buf, pd, buf, c0i, i3_)
    a.emit(61, f32, av, pa); a.emit(61, f32, bv, pb)
    a.emit(62, va, av); a.emit(62, vb, bv)
    a.emit(57, f32, r_and, f_and, va, vb)
    a.emit(110, i32, bi, bv); a.emit(62, vk, bi)
    a.emit(57, f32, r_loop, f_loop, vk)
    a.emit(184, bool_, is_half, av, c05)   # OpFOrdLessThan a < 0.5 (feeds the OpSelect)
    a.emit(169, f32, chosen, is_half, c7, r_and)
    a.emit(129, f32, summ, chosen, r_loop); a.emit(62, pc, summ)
    a.emit(110, i32, ai, av); a.emit(62, vk, ai)
    a.emit(133, f32, ab, av, bv); a.emit(80, v3, vec, av, bv, ab); a.emit(62, vvec, vec)
    a.emit(57, f32, r_sw, f_switch, vk, vvec); a.emit(62, pd, r_sw)
    a.emit(253)
    a.emit(56)
    return a.binary()


def expected(av, bv):
    f = np.float32
    av, bv = f(av), f(bv)
    r_and = f(av + bv) if (av > 0 and bv > 0) else f(-1.0)
    n = int(bv)  # ConvertFToS: toward zero
    s = f(0)
    for i in range(max(n, 0)):
        s = f(s + f(f(f(i) * f(i)) * f(0.5)))
    chosen = f(7.0) if av < f(0.5) else r_and
    out2 = f(chosen + s)
    k = int(av)
    v = np.array([av, bv, f(av * bv)], f)
    if k == 0:
        out3 = v[0]
    elif k == 1:
        out3 = f(f(f(v[0] * v[2]) + f(v[1] * v[1])) + f(v[2] * v[0]))   # OpDot order of the shim, no contraction
    elif k == 2:
        out3 = f(max(v[1], f(2.0)) * f(2.0))
    else:
        out3 = f(((k & 0xFFFFFFFF) * 0x9E3779B9 & 0xFFFFFFFF) >> 16)
    return out2, out3


def test_translated_module_computes_what_the_spirv_says(tmp_path):
    spv = tmp_path / "t.spv"
    spv.write_bytes(build_module())
    m = spv2c.Module(str(spv))
    assert [m.name(f["id"]).split("(")[0] for f in m.functions] == ["f_and", "f_loop", "f_switch", "main"]
    e = spv2c.Emitter(m)
    e.build_type_table()
    src = tmp_path / "t.c"
    src.write_text(e.emit())
    lib = tmp_path / "t.so"
    shim = ROOT / "oracle" / "ref_spv"
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-I", str(shim), str(src), "-o", str(lib), "-lm"],
                   check=True, capture_output=True)
    L = C.CDLL(str(lib))

    class Bindings(C.Structure):
        _fields_ = [("binding", C.c_void_p * 8), ("length", C.c_uint32 * 8)]

    cases = [(1.5, 3.0), (0.25, 2.0), (-1.0, 4.0), (2.0, -2.0), (2.75, 0.0), (0.0, 5.0), (3.5, 6.9), (1.25, 1.0), (7.0, 2.0), (-3.0, 2.5)]
    buf = np.zeros(4 * len(cases), np.float32)
    for i, (x, y) in enumerate(cases):
        buf[4 * i], buf[4 * i + 1] = x, y
    b = Bindings()
    b.binding[0] = buf.ctypes.data
    b.length[0] = buf.size
    L.ref_spv_invoke.argtypes = [C.POINTER(Bindings), C.c_uint32, C.c_uint32]
    for i in range(len(cases)):
        L.ref_spv_invoke(C.byref(b), i, 0)
    for i, (x, y) in enumerate(cases):
        want2, want3 = expected(x, y)
        assert buf[4 * i + 2].view(np.uint32) == np.float32(want2).view(np.uint32), (x, y, buf[4 * i + 2], want2)
        assert buf[4 * i + 3].view(np.uint32) == np.float32(want3).view(np.uint32), (x, y, buf[4 * i + 3], want3)


def test_contraction_rule_on_the_module(tmp_path):
    """--contract fuses `x + a*b` where the product has that single use: f_loop's `s0 + sq*0.5` becomes one fma; `i*i` stays
    a multiply (its only use is a multiply)."""
    spv = tmp_path / "t.spv"
    spv.write_bytes(build_module())
    e = spv2c.Emitter(spv2c.Module(str(spv)), contract=True)
    e.build_type_table()
    text = e.emit()
    assert text.count("fmaf(") == 1 and "contracted into its only use" in text


def test_unknown_opcode_is_refused(tmp_path):
    a = Asm()
    a.emit(17, 1)
    a.emit(400, 1, 2)  # not in the tool's table
    bad = tmp_path / "bad.spv"
    bad.write_bytes(a.binary())
    with pytest.raises(SystemExit):
        spv2c.Module(str(bad))
