#!/usr/bin/env python3
"""spv2c — translate the reference's compiled compute shader into C, one statement per SPIR-V instruction.

    python tools/spv2c.py [--dis] [SPV] [-o OUT.c]

Runs in the authoring container only (the GPU box has no /root/reference).  Input is the binary the reference ships and
loads at run time, `assets/shaders/compute_pass.comp.spv` (rvpt.cpp:676-681 builds its only compute pipeline from it).
Output is a C translation unit that `oracle/ref_spv/Makefile` compiles into `oracle/_ref/libref_spv.so` (git-ignored,
never shipped to the GPU box) together with `oracle/ref_spv/spv_shim.h` — the ONLY hand-written arithmetic in that
library: the GLSL.std.450 extended instructions, OpDot, OpMatrixTimesVector, the one OpFDiv whose operands are both
OpDot results (the ray/plane quotient; every other OpFDiv is C's `/`) and the image load/store conversions, i.e.
operations whose evaluation SPIR-V leaves to the Vulkan driver.  Everything else — every add, multiply,
divide, compare, select, conversion, load, store, branch, call and phi of the module's 39 functions — is emitted
mechanically from the instruction stream:

  * every result id becomes one C variable `_<id>` of the id's type, assigned exactly once per execution of its
    instruction; vector instructions are emitted component-wise in component order;
  * OpVariable(Function) becomes a local object and its id a pointer to it; OpAccessChain is pointer arithmetic on C
    structs whose member offsets are statically asserted against the module's Offset/ArrayStride decorations;
  * the structured control flow is kept as it is in the binary: one C label per OpLabel, `goto` per OpBranch,
    `if … goto` per OpBranchConditional, `switch` of gotos per OpSwitch, OpPhi resolved by assignments on the incoming
    edges; merge instructions carry no semantics and are dropped;
  * Uniform / UniformConstant / Input globals are thread-local pointers bound by the harness below; Private globals are
    thread-local objects re-initialised per invocation.

No floating-point expression is ever combined or re-associated by this tool, and the result is compiled with
`-ffp-contract=off`: the library computes what a Vulkan implementation without contraction, with the shim's builtins,
computes for this module.  `--dis` prints a readable listing instead (used while writing the oracle's citations).
"""
from __future__ import annotations

import argparse
import struct
import sys
from collections import OrderedDict, defaultdict
from pathlib import Path

DEFAULT_SPV = "/root/reference/assets/shaders/compute_pass.comp.spv"

OPNAMES = {
    3: "Source", 4: "SourceExtension", 5: "Name", 6: "MemberName", 11: "ExtInstImport", 12: "ExtInst", 14: "MemoryModel",
    15: "EntryPoint", 16: "ExecutionMode", 17: "Capability", 19: "TypeVoid", 20: "TypeBool", 21: "TypeInt", 22: "TypeFloat",
    23: "TypeVector", 24: "TypeMatrix", 25: "TypeImage", 28: "TypeArray", 29: "TypeRuntimeArray", 30: "TypeStruct",
    32: "TypePointer", 33: "TypeFunction", 41: "ConstantTrue", 42: "ConstantFalse", 43: "Constant", 44: "ConstantComposite",
    54: "Function", 55: "FunctionParameter", 56: "FunctionEnd", 57: "FunctionCall", 59: "Variable", 61: "Load", 62: "Store",
    65: "AccessChain", 68: "ArrayLength", 71: "Decorate", 72: "MemberDecorate", 79: "VectorShuffle", 80: "CompositeConstruct",
    81: "CompositeExtract", 98: "ImageRead", 99: "ImageWrite", 104: "ImageQuerySize", 110: "ConvertFToS", 111: "ConvertSToF",
    112: "ConvertUToF", 115: "FConvert", 124: "Bitcast", 127: "FNegate", 128: "IAdd", 129: "FAdd", 130: "ISub", 131: "FSub",
    132: "IMul", 133: "FMul", 136: "FDiv", 142: "VectorTimesScalar", 145: "MatrixTimesVector", 148: "Dot", 166: "LogicalOr",
    167: "LogicalAnd", 168: "LogicalNot", 169: "Select", 171: "INotEqual", 172: "UGreaterThan", 176: "ULessThan",
    177: "SLessThan", 184: "FOrdLessThan", 186: "FOrdGreaterThan", 188: "FOrdLessThanEqual", 190: "FOrdGreaterThanEqual",
    194: "ShiftRightLogical", 196: "ShiftLeftLogical", 198: "BitwiseXor", 245: "Phi", 246: "LoopMerge", 247: "SelectionMerge",
    248: "Label", 249: "Branch", 250: "BranchConditional", 251: "Switch", 253: "Return", 254: "ReturnValue", 255: "Unreachable",
}
# result-type / result-id positions: (has_type, has_result)
NO_RESULT = {3, 4, 5, 6, 14, 15, 16, 17, 56, 62, 71, 72, 99, 246, 247, 249, 250, 251, 253, 254, 255}
RESULT_ONLY = {11, 19, 20, 21, 22, 23, 24, 25, 28, 29, 30, 32, 33, 248}
GLSL450 = {6: "FSign", 13: "Sin", 14: "Cos", 15: "Tan", 31: "Sqrt", 37: "FMin", 38: "UMin", 40: "FMax", 43: "FClamp",
           46: "FMix", 66: "Length", 68: "Cross", 69: "Normalize"}
STORAGE = {0: "UniformConstant", 1: "Input", 2: "Uniform", 6: "Private", 7: "Function"}


def _string(words):
    raw = b"".join(struct.pack("<I", x) for x in words)
    return raw.split(b"\0", 1)[0].decode()


class Inst:
    __slots__ = ("op", "name", "type", "result", "args")

    def __init__(self, op, type_, result, args):
        self.op, self.name, self.type, self.result, self.args = op, OPNAMES[op], type_, result, list(args)

    def __repr__(self):
        lhs = f"%{self.result} = " if self.result is not None else ""
        ty = f" <%{self.type}>" if self.type is not None else ""
        return f"{lhs}Op{self.name}{ty} " + " ".join(str(a) for a in self.args)


class Module:
    def __init__(self, path):
        data = Path(path).read_bytes()
        w = struct.unpack("<%dI" % (len(data) // 4), data)
        if w[0] != 0x07230203:
            raise SystemExit("not a SPIR-V module")
        self.version, self.generator, self.bound = w[1], w[2], w[3]
        self.insts = []
        i = 5
        while i < len(w):
            n, op = w[i] >> 16, w[i] & 0xFFFF
            if op not in OPNAMES:
                raise SystemExit(f"opcode {op} is not in this tool's table; the module changed")
            a = w[i + 1:i + n]
            if op in NO_RESULT:
                inst = Inst(op, None, None, a)
            elif op in RESULT_ONLY:
                inst = Inst(op, None, a[0], a[1:])
            else:
                inst = Inst(op, a[0], a[1], a[2:])
            self.insts.append(inst)
            i += n
        self.names, self.member_names = {}, {}
        self.decor, self.member_decor = defaultdict(dict), defaultdict(dict)
        self.types, self.consts, self.globals = OrderedDict(), OrderedDict(), OrderedDict()
        self.functions = []
        self.ext_import = None
        self.local_size = None
        self.entry = None
        cur = None
        for x in self.insts:
            if x.op == 5:
                self.names[x.args[0]] = _string(x.args[1:])
            elif x.op == 6:
                self.member_names[(x.args[0], x.args[1])] = _string(x.args[2:])
            elif x.op == 11:
                self.ext_import = x.result
                assert _string(x.args) == "GLSL.std.450"
            elif x.op == 15:
                self.entry = x.args[1]
            elif x.op == 16 and x.args[1] == 17:
                self.local_size = tuple(x.args[2:5])
            elif x.op == 71:
                self.decor[x.args[0]][x.args[1]] = x.args[2:]
            elif x.op == 72:
                self.member_decor[(x.args[0], x.args[1])][x.args[2]] = x.args[3:]
            elif 19 <= x.op <= 33:
                self.types[x.result] = x
            elif x.op in (41, 42, 43, 44):
                self.consts[x.result] = x
            elif x.op == 59 and cur is None:
                self.globals[x.result] = x
            elif x.op == 54:
                cur = {"inst": x, "id": x.result, "ret": x.type, "ftype": x.args[1], "params": [], "blocks": OrderedDict(), "vars": []}
                self.functions.append(cur)
                blk = None
            elif x.op == 55:
                cur["params"].append(x)
            elif x.op == 56:
                cur = None
            elif cur is not None:
                if x.op == 248:
                    blk = []
                    cur["blocks"][x.result] = blk
                elif x.op == 59:
                    cur["vars"].append(x)
                else:
                    blk.append(x)

    def name(self, i):
        return self.names.get(i, "")

    # --- disassembly ---------------------------------------------------------------------------------------------
    def type_str(self, t):
        x = self.types[t]
        k = x.name
        if k == "TypeVoid":
            return "void"
        if k == "TypeBool":
            return "bool"
        if k == "TypeInt":
            return ("i" if x.args[1] else "u") + str(x.args[0])
        if k == "TypeFloat":
            return "f" + str(x.args[0])
        if k == "TypeVector":
            return f"{self.type_str(x.args[0])}x{x.args[1]}"
        if k == "TypeMatrix":
            return f"mat[{self.type_str(x.args[0])}x{x.args[1]}]"
        if k == "TypeImage":
            return "image"
        if k == "TypeArray":
            return f"{self.type_str(x.args[0])}[{self.const_value(x.args[1])}]"
        if k == "TypeRuntimeArray":
            return f"{self.type_str(x.args[0])}[]"
        if k == "TypeStruct":
            return "struct " + (self.name(t) or str(t))
        if k == "TypePointer":
            return f"{self.type_str(x.args[1])}*{STORAGE.get(x.args[0], x.args[0])}"
        if k == "TypeFunction":
            return "fn"
        return k

    def const_value(self, c):
        x = self.consts[c]
        t = self.types[x.type]
        if x.op == 41:
            return True
        if x.op == 42:
            return False
        if x.op == 44:
            return [self.const_value(a) for a in x.args]
        if t.name == "TypeInt":
            v = x.args[0]
            if t.args[1] and v >= 1 << 31:
                v -= 1 << 32
            return v
        if t.name == "TypeFloat":
            if t.args[0] == 32:
                return struct.unpack("<f", struct.pack("<I", x.args[0]))[0]
            return struct.unpack("<d", struct.pack("<II", x.args[0], x.args[1]))[0]
        raise AssertionError(x)

    def disassemble(self, out):
        def ref(a):
            if a in self.consts:
                return f"{self.const_value(a)!r}"
            n = self.name(a)
            return f"%{a}" + (f"({n})" if n else "")

        for g, x in self.globals.items():
            out.write(f"global %{g} {self.name(g)}: {self.type_str(x.type)} decor={dict(self.decor.get(g, {}))}\n")
        for f in self.functions:
            ps = ", ".join(f"%{p.result}({self.name(p.result)}): {self.type_str(p.type)}" for p in f["params"])
            out.write(f"\nfunction %{f['id']} {self.name(f['id'])} ({ps}) -> {self.type_str(f['ret'])}\n")
            for v in f["vars"]:
                out.write(f"    var %{v.result}({self.name(v.result)}): {self.type_str(v.type)}\n")
            for lbl, blk in f["blocks"].items():
                out.write(f"  L{lbl}:\n")
                for x in blk:
                    if x.op == 12:
                        body = f"{GLSL450[x.args[1]]}(" + ", ".join(ref(a) for a in x.args[2:]) + ")"
                    elif x.op in (79,):
                        body = f"VectorShuffle {ref(x.args[0])} {ref(x.args[1])} {list(x.args[2:])}"
                    elif x.op == 81:
                        body = f"CompositeExtract {ref(x.args[0])} {list(x.args[1:])}"
                    elif x.op in (246, 247, 249, 250, 251):
                        body = x.name + " " + " ".join(ref(a) if k == 0 and x.op in (250, 251) else f"L{a}" if x.op != 251 or k % 2 == 0 or k == 1 else str(a)
                                                       for k, a in enumerate(x.args))
                    else:
                        body = x.name + " " + " ".join(ref(a) for a in x.args)
                    lhs = f"%{x.result}" + (f"({self.name(x.result)})" if self.name(x.result) else "") + f": {self.type_str(x.type)} = " if x.result is not None else ""
                    out.write(f"    {lhs}{body}\n")


# =====================================================================================================================
class Emitter:
    """C back end.  One instance per module."""

    EXPORTED = ("wang_hash", "rand", "map_uniform_sphere", "camera_pinhole_ray", "camera_ortho_ray", "camera_spherical_ray",
                "intersect_triangle_fast", "intersect_aabb", "frensel_reflectance", "distance_triangle")

    def __init__(self, m: Module, contract: bool = False):
        self.m = m
        self.export = set(self.EXPORTED)
        self.rng_assign = ""
        self.contract = contract  # see plan_contraction()
        self.ctype_cache = {}
        self.typedefs = []  # emitted in dependency order as types are requested
        self.out = []

    # --- types ---------------------------------------------------------------------------------------------------
    def scalar_kind(self, t):
        x = self.m.types[t]
        if x.name == "TypeVector":
            return self.scalar_kind(x.args[0])
        if x.name == "TypeBool":
            return "b"
        if x.name == "TypeInt":
            return "i" if x.args[1] else "u"
        if x.name == "TypeFloat":
            return "f" if x.args[0] == 32 else "d"
        raise AssertionError(x)

    def vec_n(self, t):
        x = self.m.types[t]
        return x.args[1] if x.name == "TypeVector" else 0

    def ctype(self, t):
        if t in self.ctype_cache:
            return self.ctype_cache[t]
        x = self.m.types[t]
        k = x.name
        if k == "TypeVoid":
            c = "void"
        elif k == "TypeBool":
            c = "bool"
        elif k == "TypeInt":
            assert x.args[0] == 32
            c = "int32_t" if x.args[1] else "uint32_t"
        elif k == "TypeFloat":
            c = "float" if x.args[0] == 32 else "double"
        elif k == "TypeVector":
            e = self.ctype(x.args[0])
            c = f"vec{x.args[1]}{self.scalar_kind(t)}"
            self.typedefs.append(f"typedef struct {{ {e} v[{x.args[1]}]; }} {c};\n#define SHIM_HAS_{c}")
        elif k == "TypeMatrix":
            col = self.ctype(x.args[0])
            c = f"mat{x.args[1]}_{col}"
            self.typedefs.append(f"typedef struct {{ {col} c[{x.args[1]}]; }} {c};\n#define SHIM_HAS_{c}")
        elif k == "TypeImage":
            c = "shim_image*"
            self.typedefs.append("#define SHIM_HAS_image")
        elif k == "TypeArray":
            e = self.ctype(x.args[0])
            n = self.m.const_value(x.args[1])
            c = f"arr{t}"
            self.typedefs.append(f"typedef struct {{ {e} a[{n}]; }} {c};")
        elif k == "TypeRuntimeArray":
            c = self.ctype(x.args[0])  # only ever reached through its containing block
        elif k == "TypeStruct":
            c = f"S{t}_" + "".join(ch if ch.isalnum() else "_" for ch in (self.m.name(t) or "anon"))
            members = []
            for j, mt in enumerate(x.args):
                mx = self.m.types[mt]
                if mx.name == "TypeRuntimeArray":
                    members.append(f"{self.ctype(mx.args[0])} m{j}[0];")
                else:
                    members.append(f"{self.ctype(mt)} m{j};")
            self.typedefs.append(f"typedef struct {{ {' '.join(members)} }} {c};  /* {self.m.name(t)}: "
                                 + ", ".join(self.m.member_names.get((t, j), "?") for j in range(len(x.args))) + " */")
            # layout checks against the module's decorations
            for j, mt in enumerate(x.args):
                off = self.m.member_decor.get((t, j), {}).get(35)
                if off is not None:
                    self.typedefs.append(f"_Static_assert(offsetof({c}, m{j}) == {off[0]}, \"Offset decoration of {self.m.name(t)}.{self.m.member_names.get((t, j))}\");")
                mx = self.m.types[mt]
                if mx.name in ("TypeArray", "TypeRuntimeArray"):
                    stride = self.m.decor.get(mt, {}).get(6)
                    if stride is not None:
                        self.typedefs.append(f"_Static_assert(sizeof({self.ctype(mx.args[0])}) == {stride[0]}, \"ArrayStride decoration\");")
                if mx.name == "TypeMatrix":
                    ms = self.m.member_decor.get((t, j), {}).get(7)
                    if ms is not None:
                        self.typedefs.append(f"_Static_assert(sizeof({self.ctype(mx.args[0])}) == {ms[0]}, \"MatrixStride decoration\");")
                        assert 5 in self.m.member_decor[(t, j)], "RowMajor matrices are not handled"
        elif k == "TypePointer":
            c = self.ctype(x.args[1]) + "*"
        else:
            raise AssertionError(x)
        self.ctype_cache[t] = c
        return c

    # --- constants -----------------------------------------------------------------------------------------------
    def const_expr(self, c):
        x = self.m.consts[c]
        t = self.m.types[x.type]
        if x.op == 41:
            return "true"
        if x.op == 42:
            return "false"
        if x.op == 44:
            inner = ", ".join(self.const_expr(a) for a in x.args)
            return "{{" + inner + "}}"
        if t.name == "TypeInt":
            return f"INT32_C({self.m.const_value(c)})" if t.args[1] else f"UINT32_C({x.args[0]})"
        if t.name == "TypeFloat":
            if t.args[0] == 32:  # bit pattern, not a decimal literal: no parsing/rounding in between
                return f"shim_f32_bits(UINT32_C(0x{x.args[0]:08x}))"
            return f"shim_f64_bits(UINT64_C(0x{(x.args[1] << 32) | x.args[0]:016x}))"
        raise AssertionError(x)

    # --- functions -----------------------------------------------------------------------------------------------
    def fname(self, f):
        n = self.m.name(f["id"]) or f"fn{f['id']}"
        base = n.split("(")[0]
        return f"f{f['id']}_{base}"

    def v(self, i):
        return f"_{i}"

    def emit(self):
        m = self.m
        o = self.out
        body = []
        # constants, globals: force every type to exist first
        for t in m.types:
            if m.types[t].name not in ("TypeFunction",):
                self.ctype(t)
        const_decls = []
        for c, x in m.consts.items():
            ct = self.ctype(x.type)
            if x.op == 44:
                const_decls.append(f"#define {self.v(c)} (({ct}){self.const_expr(c)})")
            else:
                const_decls.append(f"#define {self.v(c)} (({ct}){self.const_expr(c)})")
        glob_decls, private_init, bind_fields = [], [], []
        for g, x in m.globals.items():
            pt = m.types[x.type]
            storage, pointee = pt.args[0], pt.args[1]
            ct = self.ctype(pointee)
            nm = m.name(g) or f"g{g}"
            if STORAGE[storage] == "Private":
                if nm == "rng_state":
                    self.rng_assign = f"s{g} = state; _{g} = &s{g};"
                glob_decls.append(f"static __thread {ct} s{g}; static __thread {ct}* _{g};  /* Private {nm} */")
                init = f"s{g} = {self.v(x.args[1])}; " if len(x.args) > 1 else f"memset(&s{g}, 0, sizeof s{g}); "
                private_init.append(f"{init}_{g} = &s{g};")
            elif STORAGE[storage] == "Input":
                assert m.decor[g].get(11) == [28], "only gl_GlobalInvocationID is expected as an input"
                glob_decls.append(f"static __thread {ct} s{g}; static __thread {ct}* _{g};  /* Input {nm} */")
                private_init.append(f"_{g} = &s{g}; s{g}.v[0] = gid_x; s{g}.v[1] = gid_y; s{g}.v[2] = 0;")
            else:
                binding = m.decor[g].get(33, [None])[0]
                label = nm or m.name(pointee)
                if m.types[pointee].name == "TypeImage":
                    glob_decls.append(f"static __thread {ct} s{g}; static __thread {ct}* _{g};  /* image binding {binding}: {label} */")
                    private_init.append(f"s{g} = (shim_image*)b->binding[{binding}]; _{g} = &s{g};")
                else:
                    glob_decls.append(f"static __thread {ct}* _{g};  /* {STORAGE[storage]} binding {binding}: {label or m.name(pointee)} */")
                    private_init.append(f"_{g} = ({ct}*)b->binding[{binding}];")
                    if any(m.types[mt].name == "TypeRuntimeArray" for mt in m.types[pointee].args):
                        private_init.append(f"len{g} = b->length[{binding}];")
                        glob_decls.append(f"static __thread uint32_t len{g};  /* OpArrayLength of binding {binding} */")
        # function prototypes
        protos = []
        for f in m.functions:
            ps = ", ".join(f"{self.ctype(p.type)} {self.v(p.result)}" for p in f["params"]) or "void"
            protos.append(f"static {self.ctype(f['ret'])} {self.fname(f)}({ps});")
        for f in m.functions:
            body.extend(self.emit_function(f))
        # exported wrappers of selected functions, so that tests can compare them one at a time (values such as the
        # barycentrics of intersect_triangle_fast never reach a pixel; only a call of the function itself shows them)
        exports = []
        self.rand_fname = next((self.fname(f) for f in m.functions if (m.name(f["id"]) or "").split("(")[0] == "rand"), None)
        for f in m.functions:
            base = (m.name(f["id"]) or "").split("(")[0]
            if base in self.export:
                ps = ", ".join(f"{self.ctype(p.type)} {self.v(p.result)}" for p in f["params"]) or "void"
                call = f"{self.fname(f)}(" + ", ".join(self.v(p.result) for p in f["params"]) + ")"
                ret = self.ctype(f["ret"])
                exports.append(f"__attribute__((visibility(\"default\"))) {ret} ref_spv_fn_{base}(const shim_bindings* b{', ' + ps if f['params'] else ''})\n"
                               f"{{\n    ref_spv_bind(b, 0, 0);\n    {'return ' if ret != 'void' else ''}{call};\n}}")
        entry = next(f for f in m.functions if f["id"] == m.entry)
        o.append("/* GENERATED by tools/spv2c.py from the reference's compute_pass.comp.spv — do not edit, do not commit. */")
        o.append("#include <math.h>\n#include <stdbool.h>\n#include <stddef.h>\n#include <stdint.h>\n#include <string.h>")
        o.append('#include "spv_shim.h"')
        o.extend(self.typedefs)
        o.append("#include \"spv_shim_vec.h\"")
        o.extend(const_decls)
        o.extend(glob_decls)
        o.extend(protos)
        o.extend(body)
        o.append(f"""
/* binds the module's global variables for this thread: descriptors, gl_GlobalInvocationID = (gid_x, gid_y, 0), Private
 * variables back to their initial values */
static void ref_spv_bind(const shim_bindings* b, uint32_t gid_x, uint32_t gid_y)
{{
    {' '.join(private_init)}
}}
/* harness entry: one shader invocation against the bound resources */
__attribute__((visibility("default"))) void ref_spv_invoke(const shim_bindings* b, uint32_t gid_x, uint32_t gid_y)
{{
    ref_spv_bind(b, gid_x, gid_y);
    {self.fname(entry)}();
}}
/* direct calls of single functions of the module (pointer parameters are the module's Function-storage pointers) */
{chr(10).join(exports)}
/* rng_state is a Private global: seeded here for the functions that draw random numbers */
__attribute__((visibility("default"))) void ref_spv_set_rng(uint32_t state) {{ {self.rng_assign} }}
{self.rand_stream_export()}
__attribute__((visibility("default"))) void ref_spv_local_size(uint32_t* xyz) {{ xyz[0] = {m.local_size[0]}; xyz[1] = {m.local_size[1]}; xyz[2] = {m.local_size[2]}; }}
__attribute__((visibility("default"))) uint32_t ref_spv_function_count(void) {{ return {len(m.functions)}; }}
__attribute__((visibility("default"))) uint32_t ref_spv_instruction_count(void) {{ return {sum(len(b) for f in m.functions for b in f['blocks'].values())}; }}
""")
        return "\n".join(o) + "\n"

    def plan_contraction(self, f):
        """--contract: the one floating-point contraction rule of the build's arithmetic specification.

        An OpFAdd / OpFSub `x ± y` of 32-bit floats becomes one fused multiply-add when an operand is the result of an
        OpFMul / OpVectorTimesScalar of the same block that has no other use: `x ± a*b -> fma(±a, b, x)` if the RIGHT
        operand is such a product, else `a*b ± y -> fma(a, b, ±y)` if the left one is.  Nothing else is contracted and
        nothing is re-associated.  (This is the licence GLSL gives a Vulkan compiler for expressions not marked
        `precise`; the rule only fixes which of the allowed results is taken.)
        Returns {add/sub result id: (product inst, side)} and the set of product ids that are folded away.
        """
        uses = defaultdict(int)
        defs = {}
        for lbl, blk in f["blocks"].items():
            for x in blk:
                if x.result is not None:
                    defs[x.result] = (lbl, x)
                for a in x.args:
                    if isinstance(a, int):
                        uses[a] += 1  # literals may alias ids; only ever makes the rule more conservative
        plan, folded = {}, set()
        for lbl, blk in f["blocks"].items():
            for x in blk:
                if x.name not in ("FAdd", "FSub") or self.scalar_kind(x.type) != "f":
                    continue
                for side in (1, 0):
                    d = defs.get(x.args[side])
                    if d and d[0] == lbl and d[1].name in ("FMul", "VectorTimesScalar") and uses[x.args[side]] == 1 \
                            and x.args[side] not in folded and x.args[0] != x.args[1]:
                        plan[x.result] = (d[1], side)
                        folded.add(x.args[side])
                        break
        return plan, folded

    def emit_fused(self, x, prod, side):
        v = self.v
        n = self.vec_n(x.type)
        other = x.args[1 - side]
        out = []
        for k in (range(n) if n else [None]):
            ix = "" if k is None else f".v[{k}]"
            a = f"{v(prod.args[0])}{ix}"
            b = f"{v(prod.args[1])}{ix}" if (k is None or self.vec_n(self.type_of(prod.args[1]))) else v(prod.args[1])
            o = f"{v(other)}{ix}"
            if x.name == "FAdd":
                e = f"fmaf({a}, {b}, {o})"
            elif side == 1:  # x - a*b
                e = f"fmaf(-{a}, {b}, {o})"
            else:            # a*b - y
                e = f"fmaf({a}, {b}, -{o})"
            out.append(f"{v(x.result)}{ix} = {e};")
        return out

    def rand_stream_export(self):
        if not self.rand_fname or not self.rng_assign:
            return "/* the module has no rand() / rng_state: no ref_spv_rand_stream */"
        return ("__attribute__((visibility(\"default\"))) void ref_spv_rand_stream(const shim_bindings* b, uint32_t state, uint32_t n, float* out)\n"
                "{\n    ref_spv_bind(b, 0, 0);\n    " + self.rng_assign + "\n    for (uint32_t i = 0; i < n; ++i) out[i] = " + self.rand_fname + "();\n}")

    def emit_function(self, f):
        m = self.m
        L = []
        # result ids produced by OpDot in this function: an OpFDiv of two of them goes to the shim (spv_shim.h, item 3)
        self.dot_results = {x.result for blk in f["blocks"].values() for x in blk if x.name == "Dot"}
        plan, folded = self.plan_contraction(f) if self.contract else ({}, set())
        ps = ", ".join(f"{self.ctype(p.type)} {self.v(p.result)}" for p in f["params"]) or "void"
        L.append(f"\n/* {m.name(f['id'])} */")
        L.append(f"static {self.ctype(f['ret'])} {self.fname(f)}({ps})\n{{")
        # declarations: function variables, then one C variable per result id
        for vx in f["vars"]:
            ct = self.ctype(m.types[vx.type].args[1])
            # an uninitialised Function variable holds an undefined value in SPIR-V; zero is the value chosen here
            init = f" = {self.v(vx.args[1])}" if len(vx.args) > 1 else " = {0}"
            L.append(f"    {ct} s{vx.result}{init}; {ct}* const {self.v(vx.result)} = &s{vx.result};  /* {m.name(vx.result)} */")
        for blk in f["blocks"].values():
            for x in blk:
                if x.result is not None and m.types[x.type].name != "TypeVoid":
                    L.append(f"    {self.ctype(x.type)} {self.v(x.result)};")
        # phi edge assignments
        phi_moves = defaultdict(list)  # predecessor label -> [(phi id, value id)]
        for lbl, blk in f["blocks"].items():
            phis = [x for x in blk if x.op == 245]
            ids = {x.result for x in phis}
            for x in phis:
                for val, pred in zip(x.args[0::2], x.args[1::2]):
                    assert val not in ids, "phi reading another phi of its own block needs parallel copies"
                    phi_moves[pred].append((x.result, val))
        for lbl, blk in f["blocks"].items():
            L.append(f"L{lbl}: ;")
            for x in blk:
                if x.op in (249, 250, 251):  # terminators that leave through an edge: phi copies first
                    for dst, val in phi_moves.get(lbl, []):
                        L.append(f"    {self.v(dst)} = {self.v(val)};")
                if x.result in folded:
                    L.append(f"    /* {self.v(x.result)} = Op{x.name}: contracted into its only use */")
                    continue
                lines = self.emit_fused(x, *plan[x.result]) if x.result in plan else self.emit_inst(f, x)
                for line in lines:
                    L.append("    " + line)
        L.append("}")
        return L

    # --- instructions --------------------------------------------------------------------------------------------
    def access_chain(self, base_ptr_type, base, idx):
        m = self.m
        t = m.types[base_ptr_type].args[1]
        e = f"(*{self.v(base)})"
        for i in idx:
            x = m.types[t]
            if x.name == "TypeStruct":
                j = m.const_value(i)
                mt = x.args[j]
                e = f"{e}.m{j}"
                t = mt
            elif x.name == "TypeVector":
                e = f"{e}.v[{self.v(i)}]"
                t = x.args[0]
            elif x.name == "TypeMatrix":
                e = f"{e}.c[{self.v(i)}]"
                t = x.args[0]
            elif x.name == "TypeArray":
                e = f"{e}.a[{self.v(i)}]"
                t = x.args[0]
            elif x.name == "TypeRuntimeArray":
                e = f"{e}[{self.v(i)}]"
                t = x.args[0]
            else:
                raise AssertionError(x)
        return "&" + e

    def componentwise(self, x, fmt):
        """fmt is a format string over {r} {a} {b} {c} already indexed per component."""
        n = self.vec_n(x.type)
        r = self.v(x.result)
        ops = [self.v(a) for a in x.args]
        out = []
        if n == 0:
            out.append(fmt.format(r=r, a=ops[0] if ops else "", b=ops[1] if len(ops) > 1 else "", c=ops[2] if len(ops) > 2 else "") + ";")
        else:
            for k in range(n):
                def comp(i):
                    if i >= len(ops):
                        return ""
                    # operand may be scalar (e.g. Select with scalar condition)
                    return f"{ops[i]}.v[{k}]" if self.vec_n(self.type_of(x.args[i])) else ops[i]
                out.append(fmt.format(r=f"{r}.v[{k}]", a=comp(0), b=comp(1), c=comp(2)) + ";")
        return out

    def type_of(self, i):
        return self._types[i]

    def build_type_table(self):
        m = self.m
        tt = {}
        for c, x in m.consts.items():
            tt[c] = x.type
        for g, x in m.globals.items():
            tt[g] = x.type
        for f in m.functions:
            for p in f["params"]:
                tt[p.result] = p.type
            for vx in f["vars"]:
                tt[vx.result] = vx.type
            for blk in f["blocks"].values():
                for x in blk:
                    if x.result is not None:
                        tt[x.result] = x.type
        self._types = tt

    def emit_inst(self, f, x):
        m, v = self.m, self.v
        n = x.name
        r = v(x.result) if x.result is not None else None
        a = x.args
        bin_ops = {"IAdd": "+", "ISub": "-", "IMul": "*", "FAdd": "+", "FSub": "-", "FMul": "*", "FDiv": "/",
                   "BitwiseXor": "^", "ShiftRightLogical": ">>", "ShiftLeftLogical": "<<",
                   "LogicalOr": "||", "LogicalAnd": "&&",
                   "INotEqual": "!=", "UGreaterThan": ">", "ULessThan": "<", "SLessThan": "<",
                   "FOrdLessThan": "<", "FOrdGreaterThan": ">", "FOrdLessThanEqual": "<=", "FOrdGreaterThanEqual": ">="}
        if n == "FDiv" and a[0] in self.dot_results and a[1] in self.dot_results and self.vec_n(x.type) == 0 and self.scalar_kind(x.type) == "f":
            return [f"{r} = shim_fdiv_dots({v(a[0])}, {v(a[1])});"]
        if n in bin_ops:
            op = bin_ops[n]
            if n in ("IAdd", "ISub", "IMul", "ShiftRightLogical", "ShiftLeftLogical", "BitwiseXor"):
                # two's-complement wrap-around regardless of signedness: compute in uint32_t
                ct = self.ctype(m.types[x.type].args[0]) if self.vec_n(x.type) else self.ctype(x.type)
                return self.componentwise(x, "{r} = (" + ct + ")((uint32_t){a} " + op + " (uint32_t){b})")
            if n in ("UGreaterThan", "ULessThan"):
                return self.componentwise(x, "{r} = (uint32_t){a} " + op + " (uint32_t){b}")
            if n == "SLessThan":
                return self.componentwise(x, "{r} = (int32_t){a} " + op + " (int32_t){b}")
            return self.componentwise(x, "{r} = {a} " + op + " {b}")
        if n == "FNegate":
            return self.componentwise(x, "{r} = -{a}")
        if n == "LogicalNot":
            return self.componentwise(x, "{r} = !{a}")
        if n == "Select":
            return self.componentwise(x, "{r} = {a} ? {b} : {c}")
        if n == "ConvertFToS":
            return self.componentwise(x, "{r} = shim_f2i({a})")
        if n in ("ConvertSToF", "ConvertUToF", "FConvert"):
            et = self.ctype(m.types[x.type].args[0]) if self.vec_n(x.type) else self.ctype(x.type)
            src = {"ConvertSToF": "(int32_t)", "ConvertUToF": "(uint32_t)", "FConvert": ""}[n]
            return self.componentwise(x, "{r} = (" + et + ")" + src + "{a}")
        if n == "Bitcast":
            assert self.vec_n(x.type) == self.vec_n(self.type_of(a[0]))
            return [f"memcpy(&{r}, &{v(a[0])}, sizeof {r});"]
        if n == "VectorTimesScalar":
            k = self.vec_n(x.type)
            return [f"{r}.v[{i}] = {v(a[0])}.v[{i}] * {v(a[1])};" for i in range(k)]
        if n == "MatrixTimesVector":
            mt = m.types[self.type_of(a[0])]
            cols, rows = mt.args[1], self.vec_n(mt.args[0])
            return [f"{r} = shim_mat{cols}x{rows}_times_vec({v(a[0])}, {v(a[1])});"]
        if n == "Dot":
            k = self.vec_n(self.type_of(a[0]))
            return [f"{r} = shim_dot{k}({v(a[0])}, {v(a[1])});"]
        if n == "ExtInst":
            assert a[0] == m.ext_import
            name = GLSL450[a[1]]
            ops = a[2:]
            k = self.vec_n(x.type)
            kind = self.scalar_kind(x.type) if name not in ("Length",) else self.scalar_kind(self.type_of(ops[0]))
            if name in ("Length", "Cross", "Normalize"):
                kk = self.vec_n(self.type_of(ops[0]))
                return [f"{r} = shim_{name.lower()}{kk}{kind}(" + ", ".join(v(p) for p in ops) + ");"]
            # component-wise builtins: scalar shim function applied per component
            fn = f"shim_{name.lower()}_{kind}"
            if k == 0:
                return [f"{r} = {fn}(" + ", ".join(v(p) for p in ops) + ");"]
            out = []
            for i in range(k):
                argl = ", ".join(f"{v(p)}.v[{i}]" if self.vec_n(self.type_of(p)) else v(p) for p in ops)
                out.append(f"{r}.v[{i}] = {fn}({argl});")
            return out
        if n == "Load":
            return [f"{r} = *{v(a[0])};"]
        if n == "Store":
            return [f"*{v(a[0])} = {v(a[1])};"]
        if n == "AccessChain":
            return [f"{r} = {self.access_chain(self.type_of(a[0]), a[0], a[1:])};"]
        if n == "ArrayLength":
            return [f"{r} = len{a[0]};"]
        if n == "VectorShuffle":
            n1 = self.vec_n(self.type_of(a[0]))
            out = []
            for i, c in enumerate(a[2:]):
                src = f"{v(a[0])}.v[{c}]" if c < n1 else f"{v(a[1])}.v[{c - n1}]"
                out.append(f"{r}.v[{i}] = {src};")
            return out
        if n == "CompositeConstruct":
            tx = m.types[x.type]
            if tx.name == "TypeVector":
                out, i = [], 0
                for p in a:
                    pn = self.vec_n(self.type_of(p))
                    if pn == 0:
                        out.append(f"{r}.v[{i}] = {v(p)};")
                        i += 1
                    else:
                        for c in range(pn):
                            out.append(f"{r}.v[{i}] = {v(p)}.v[{c}];")
                            i += 1
                assert i == tx.args[1]
                return out
            if tx.name == "TypeMatrix":
                return [f"{r}.c[{i}] = {v(p)};" for i, p in enumerate(a)]
            if tx.name == "TypeStruct":
                return [f"{r}.m{i} = {v(p)};" for i, p in enumerate(a)]
            if tx.name == "TypeArray":
                return [f"{r}.a[{i}] = {v(p)};" for i, p in enumerate(a)]
            raise AssertionError(x)
        if n == "CompositeExtract":
            t = self.type_of(a[0])
            e = v(a[0])
            for i in a[1:]:
                tx = m.types[t]
                if tx.name == "TypeStruct":
                    e, t = f"{e}.m{i}", tx.args[i]
                elif tx.name == "TypeVector":
                    e, t = f"{e}.v[{i}]", tx.args[0]
                elif tx.name == "TypeMatrix":
                    e, t = f"{e}.c[{i}]", tx.args[0]
                elif tx.name == "TypeArray":
                    e, t = f"{e}.a[{i}]", tx.args[0]
                else:
                    raise AssertionError(tx)
            return [f"{r} = {e};"]
        if n == "ImageRead":
            return [f"{r} = shim_image_read({v(a[0])}, {v(a[1])});"]
        if n == "ImageWrite":
            return [f"shim_image_write({v(a[0])}, {v(a[1])}, {v(a[2])});"]
        if n == "ImageQuerySize":
            return [f"{r} = shim_image_size({v(a[0])});"]
        if n == "FunctionCall":
            callee = next(g for g in m.functions if g["id"] == a[0])
            call = f"{self.fname(callee)}(" + ", ".join(v(p) for p in a[1:]) + ")"
            if m.types[x.type].name == "TypeVoid":
                return [call + ";"]
            return [f"{r} = {call};"]
        if n == "Phi":
            return [f"/* {r} = phi: assigned on the incoming edges */"]
        if n in ("LoopMerge", "SelectionMerge"):
            return [f"/* {n} */"]
        if n == "Branch":
            return [f"goto L{a[0]};"]
        if n == "BranchConditional":
            return [f"if ({v(a[0])}) goto L{a[1]}; else goto L{a[2]};"]
        if n == "Switch":
            cases = " ".join(f"case {lit}: goto L{lbl};" for lit, lbl in zip(a[2::2], a[3::2]))
            return [f"switch ((int32_t){v(a[0])}) {{ {cases} default: goto L{a[1]}; }}"]
        if n == "Return":
            return ["return;"]
        if n == "ReturnValue":
            return [f"return {v(a[0])};"]
        if n == "Unreachable":
            return ["shim_unreachable();"]
        raise SystemExit(f"no C emission for Op{n}")


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("spv", nargs="?", default=DEFAULT_SPV)
    ap.add_argument("--dis", action="store_true", help="print a listing instead of C")
    ap.add_argument("--contract", action="store_true", help="apply the build's FMA contraction rule (see plan_contraction)")
    ap.add_argument("-o", "--output", default="-")
    args = ap.parse_args()
    m = Module(args.spv)
    out = sys.stdout if args.output == "-" else open(args.output, "w")
    if args.dis:
        m.disassemble(out)
        return
    e = Emitter(m, contract=args.contract)
    e.build_type_table()
    out.write(e.emit())


if __name__ == "__main__":
    main()
