"""tools/spv2c.py on a SECOND hand-assembled SPIR-V module, covering what tests/test_spv2c.py's module does not: OpPhi at loop
headers (SSA loop-carried values instead of Function variables), nested structured loops with a `break` out of the inner one (a phi
at the inner merge block fed by the loop-exit edge and by the break edge), a matrix built with OpCompositeConstruct, read with
OpCompositeExtract and multiplied with OpMatrixTimesVector, and the contraction rule applied to an OpVectorTimesScalar that is the
LEFT operand of an OpFSub (`v*s - w` -> fma(v, s, -w) per component).  Closed-form expectations; needs only gcc."""
import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))
import spv2c  # noqa: E402
from test_spv2c import Asm  # noqa: E402


def build_module():
    a = Asm()
    n = a.new
    glsl, main, f_nested = n(), n(), n()
    void, bool_, u32, i32, f32, v2, uv3, m2 = (n() for _ in range(8))
    fn_void, fn_nested = n(), n()
    p_f_func, p_i_func = n(), n()
    rt_arr, block, p_block, p_f_buf, buf = (n() for _ in range(5))
    p_uv3_in, gid, p_u_in = n(), n(), n()
    c0f, c100f, c2f, c01f, c0i, c1i, c6i, c8u, c0u = (n() for _ in range(9))
    cu = [n() for _ in range(7)]  # uint 1..7
    a.emit(17, 1)
    a.emit(11, glsl, "GLSL.std.450")
    a.emit(14, 0, 1)
    a.emit(15, 5, main, "main", gid)
    a.emit(16, main, 17, 8, 1, 1)
    for i, s in ((main, "main"), (f_nested, "f_nested(i1;i1;"), (buf, "buf"), (gid, "gl_GlobalInvocationID")):
        a.emit(5, i, s)
    a.emit(71, rt_arr, 6, 4)
    a.emit(72, block, 0, 35, 0)
    a.emit(71, block, 3)
    a.emit(71, buf, 34, 0)
    a.emit(71, buf, 33, 0)
    a.emit(71, gid, 11, 28)
    a.emit(19, void); a.emit(20, bool_); a.emit(21, u32, 32, 0); a.emit(21, i32, 32, 1); a.emit(22, f32, 32)
    a.emit(23, v2, f32, 2); a.emit(23, uv3, u32, 3); a.emit(24, m2, v2, 2)
    a.emit(32, p_f_func, 7, f32); a.emit(32, p_i_func, 7, i32)
    a.emit(33, fn_void, void); a.emit(33, fn_nested, f32, p_i_func, p_i_func)
    a.emit(29, rt_arr, f32); a.emit(30, block, rt_arr); a.emit(32, p_block, 2, block); a.emit(32, p_f_buf, 2, f32)
    a.emit(59, p_block, buf, 2)
    a.emit(32, p_uv3_in, 1, uv3); a.emit(59, p_uv3_in, gid, 1); a.emit(32, p_u_in, 1, u32)
    for c, v in ((c0f, 0.0), (c100f, 100.0), (c2f, 2.0), (c01f, 0.1)):
        a.emit(43, f32, c, v)
    for c, v in ((c0i, 0), (c1i, 1), (c6i, 6)):
        a.emit(43, i32, c, v)
    a.emit(43, u32, c8u, 8); a.emit(43, u32, c0u, 0)
    for k, c in enumerate(cu):
        a.emit(43, u32, c, k + 1)

    # float f_nested(int* pn, int* pm):
    #   acc = 0; for (i = 0; i < n; ++i) { for (j = 0; j < m; ++j) { if (6 < i*j) { acc += 100; break; } acc += float(i + j); } }
    # all loop-carried values are OpPhi results at the two loop headers; the inner merge block has a phi over {loop exit, break}
    pn, pm = n(), n()
    entry, oh, oc, ipre, ih, ic, ib, brk, after, icont, imerge, ocont, omerge = (n() for _ in range(13))
    nv, mv = n(), n()
    i, acc, i1, ci = n(), n(), n(), n()
    j, acc2, j1, cj, t, isbrk, accb, s, fs, acc3, accm = (n() for _ in range(11))
    a.emit(54, f32, f_nested, 0, fn_nested); a.emit(55, p_i_func, pn); a.emit(55, p_i_func, pm)
    a.emit(248, entry); a.emit(61, i32, nv, pn); a.emit(61, i32, mv, pm); a.emit(249, oh)
    a.emit(248, oh); a.emit(245, i32, i, c0i, entry, i1, ocont); a.emit(245, f32, acc, c0f, entry, accm, ocont)
    a.emit(246, omerge, ocont, 0); a.emit(249, oc)
    a.emit(248, oc); a.emit(177, bool_, ci, i, nv); a.emit(250, ci, ipre, omerge)
    a.emit(248, ipre); a.emit(249, ih)
    a.emit(248, ih); a.emit(245, i32, j, c0i, ipre, j1, icont); a.emit(245, f32, acc2, acc, ipre, acc3, icont)
    a.emit(246, imerge, icont, 0); a.emit(249, ic)
    a.emit(248, ic); a.emit(177, bool_, cj, j, mv); a.emit(250, cj, ib, imerge)
    a.emit(248, ib); a.emit(132, i32, t, i, j); a.emit(177, bool_, isbrk, c6i, t); a.emit(247, after, 0); a.emit(250, isbrk, brk, after)
    a.emit(248, brk); a.emit(129, f32, accb, acc2, c100f); a.emit(249, imerge)
    a.emit(248, after); a.emit(128, i32, s, i, j); a.emit(111, f32, fs, s); a.emit(129, f32, acc3, acc2, fs); a.emit(249, icont)
    a.emit(248, icont); a.emit(128, i32, j1, j, c1i); a.emit(249, ih)
    a.emit(248, imerge); a.emit(245, f32, accm, acc2, ic, accb, brk); a.emit(249, ocont)
    a.emit(248, ocont); a.emit(128, i32, i1, i, c1i); a.emit(249, oh)
    a.emit(248, omerge); a.emit(254, acc)
    a.emit(56)

    # void main(): base = 8*gid.x; a = buf[base], b = buf[base+1]
    #   buf[base+2] = f_nested(int(a), int(b))
    #   M = mat2(vec2(a, b), vec2(b, a*b)); mv = M * vec2(a, 2); buf[base+3] = mv.x; buf[base+4] = mv.y + M[1][0]
    #   r = vec2(a, b) * 0.1 - vec2(b, a)   (OpVectorTimesScalar, single use, LEFT operand of the OpFSub); buf[base+5] = r.x; buf[base+6] = r.y
    em = n()
    pg, g, base = n(), n(), n()
    idx = [n() for _ in range(7)]
    ptr = [n() for _ in range(7)]
    av, bv, ai, bi, vn, vm, r_nest = (n() for _ in range(7))
    ab, col0, col1, mat, e10, xv, mvv, mvx, mvy, o4 = (n() for _ in range(10))
    vv, ww, prod, rr, rx, ry = (n() for _ in range(6))
    a.emit(54, void, main, 0, fn_void)
    a.emit(248, em)
    a.emit(59, p_i_func, vn, 7); a.emit(59, p_i_func, vm, 7)
    a.emit(65, p_u_in, pg, gid, c0u); a.emit(61, u32, g, pg); a.emit(132, u32, base, g, c8u)
    a.emit(65, p_f_buf, ptr[0], buf, c0i, base)
    for k in range(1, 7):
        a.emit(128, u32, idx[k], base, cu[k - 1]); a.emit(65, p_f_buf, ptr[k], buf, c0i, idx[k])
    a.emit(61, f32, av, ptr[0]); a.emit(61, f32, bv, ptr[1])
    a.emit(110, i32, ai, av); a.emit(110, i32, bi, bv); a.emit(62, vn, ai); a.emit(62, vm, bi)
    a.emit(57, f32, r_nest, f_nested, vn, vm); a.emit(62, ptr[2], r_nest)
    a.emit(133, f32, ab, av, bv)
    a.emit(80, v2, col0, av, bv); a.emit(80, v2, col1, bv, ab); a.emit(80, m2, mat, col0, col1)
    a.emit(81, f32, e10, mat, 1, 0)
    a.emit(80, v2, xv, av, c2f); a.emit(145, v2, mvv, mat, xv)
    a.emit(81, f32, mvx, mvv, 0); a.emit(81, f32, mvy, mvv, 1); a.emit(129, f32, o4, mvy, e10)
    a.emit(62, ptr[3], mvx); a.emit(62, ptr[4], o4)
    a.emit(80, v2, vv, av, bv); a.emit(80, v2, ww, bv, av)
    a.emit(142, v2, prod, vv, c01f); a.emit(131, v2, rr, prod, ww)
    a.emit(81, f32, rx, rr, 0); a.emit(81, f32, ry, rr, 1); a.emit(62, ptr[5], rx); a.emit(62, ptr[6], ry)
    a.emit(253)
    a.emit(56)
    return a.binary()


def _fmaf(x, y, z):
    libm = C.CDLL("libm.so.6")
    libm.fmaf.restype = C.c_float
    libm.fmaf.argtypes = [C.c_float] * 3
    return np.float32(libm.fmaf(float(x), float(y), float(z)))


def expected(av, bv, contract):
    f = np.float32
    av, bv = f(av), f(bv)
    n, m = int(av), int(bv)
    acc = f(0)
    for i in range(max(n, 0)):
        for j in range(max(m, 0)):
            if 6 < i * j:
                acc = f(acc + f(100.0))
                break
            acc = f(acc + f(i + j))
    ab = f(av * bv)
    two = f(2.0)
    if contract:  # shim: s = c0[k]*x0; s = fma(c1[k], x1, s)
        mvx = _fmaf(bv, two, f(av * av))
        mvy = _fmaf(ab, two, f(bv * av))
        r = [_fmaf(av, f(0.1), -bv), _fmaf(bv, f(0.1), -av)]
    else:
        mvx = f(f(av * av) + f(bv * two))
        mvy = f(f(bv * av) + f(ab * two))
        r = [f(f(av * f(0.1)) - bv), f(f(bv * f(0.1)) - av)]
    return acc, mvx, f(mvy + bv), r[0], r[1]


def _run(tmp_path, contract):
    spv = tmp_path / "m2.spv"
    spv.write_bytes(build_module())
    m = spv2c.Module(str(spv))
    assert [m.name(f["id"]).split("(")[0] for f in m.functions] == ["f_nested", "main"]
    e = spv2c.Emitter(m, contract=contract)
    e.build_type_table()
    text = e.emit()
    src = tmp_path / ("m2c.c" if contract else "m2.c")
    src.write_text(text)
    lib = tmp_path / ("m2c.so" if contract else "m2.so")
    shim = ROOT / "oracle" / "ref_spv"
    subprocess.run(["gcc", "-O1", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", *(["-DREF_SPV_FUSED"] if contract else []), "-I", str(shim),
                    str(src), "-o", str(lib), "-lm"], check=True, capture_output=True)
    L = C.CDLL(str(lib))

    class Bindings(C.Structure):
        _fields_ = [("binding", C.c_void_p * 8), ("length", C.c_uint32 * 8)]

    cases = [(3.0, 4.0), (5.0, 5.0), (0.0, 3.0), (4.0, 0.0), (2.7, 9.3), (1.0, 1.0), (-2.0, 3.0), (7.3, 2.1), (6.0, 7.0), (0.3, 0.7), (3.89, 5.21), (1.16, 4.37), (8.83, 3.43)]
    buf = np.zeros(8 * len(cases), np.float32)
    for i, (x, y) in enumerate(cases):
        buf[8 * i], buf[8 * i + 1] = x, y
    b = Bindings()
    b.binding[0] = buf.ctypes.data
    b.length[0] = buf.size
    L.ref_spv_invoke.argtypes = [C.POINTER(Bindings), C.c_uint32, C.c_uint32]
    for i in range(len(cases)):
        L.ref_spv_invoke(C.byref(b), i, 0)
    for i, (x, y) in enumerate(cases):
        want = expected(x, y, contract)
        got = buf[8 * i + 2: 8 * i + 7]
        for k, (g, w) in enumerate(zip(got, want)):
            assert np.float32(g).view(np.uint32) == np.float32(w).view(np.uint32), (contract, (x, y), k, g, w)
    return text


def test_phi_loops_break_and_matrix(tmp_path):
    text = _run(tmp_path, contract=False)
    assert "fmaf(" not in text


def test_left_operand_vector_times_scalar_is_contracted(tmp_path):
    """--contract: `v*s - w` with the OpVectorTimesScalar as the LEFT operand and no other use becomes fma(v, s, -w) per component;
    the additions of f_nested (no product among their operands) and `mv.y + M[1][0]` stay plain."""
    text = _run(tmp_path, contract=True)
    assert text.count("fmaf(") == 2 and "contracted into its only use" in text
    # 0.1f is inexact: the fused and the unfused result differ for at least one of the cases (the test above would not notice a missing fma otherwise)
    assert any(expected(x, y, True)[3:] != expected(x, y, False)[3:] for x, y in [(3.89, 5.21), (1.16, 4.37), (8.83, 3.43)])
