#!/usr/bin/env python3
"""Extract interface facts from the reference's compiled compute shader (assets/shaders/compute_pass.comp.spv) ->
tests/golden/reference_spv_facts.json.  Run in the authoring container (the GPU box has no /root/reference).

What is kept is metadata of the binary, not its code: buffer block names with their bindings, member byte offsets and
array strides (the layouts the C ABI's structs must match), the names of the functions that made it into the module
(the live call graph), the storage image format, the work-group size and the pool of 32-bit float
constants (so that the constants restated in the oracle can be checked against the compiled shader).
"""
import json
import struct
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SPV = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/assets/shaders/compute_pass.comp.spv")


def string(words):
    raw = b"".join(struct.pack("<I", w) for w in words)
    return raw.split(b"\0", 1)[0].decode()


def main():
    data = SPV.read_bytes()
    w = struct.unpack("<%dI" % (len(data) // 4), data)
    assert w[0] == 0x07230203
    names, member_names, offsets, strides, bindings, sets, blocks = {}, {}, {}, {}, {}, {}, set()
    structs, floats32, consts, functions, images, arrays, pointers, variables = {}, set(), [], [], {}, {}, {}, {}
    local_size = None
    i = 5
    while i < len(w):
        n, op = w[i] >> 16, w[i] & 0xFFFF
        a = w[i + 1:i + n]
        if op == 5:      # OpName
            names[a[0]] = string(a[1:])
        elif op == 6:    # OpMemberName
            member_names[(a[0], a[1])] = string(a[2:])
        elif op == 16 and a[1] == 17:  # OpExecutionMode LocalSize
            local_size = list(a[2:5])
        elif op == 71:   # OpDecorate
            if a[1] == 6:
                strides[a[0]] = a[2]
            elif a[1] == 33:
                bindings[a[0]] = a[2]
            elif a[1] == 34:
                sets[a[0]] = a[2]
            elif a[1] in (2, 3):  # Block / BufferBlock
                blocks.add(a[0])
        elif op == 72 and a[2] == 35:  # OpMemberDecorate Offset
            offsets[(a[0], a[1])] = a[3]
        elif op == 22 and a[1] == 32:  # OpTypeFloat 32
            floats32.add(a[0])
        elif op == 25:   # OpTypeImage: sampled type, dim, depth, arrayed, ms, sampled, format
            images[a[0]] = a[7]
        elif op in (28, 29):  # OpTypeArray / OpTypeRuntimeArray
            arrays[a[0]] = a[1]
        elif op == 30:   # OpTypeStruct
            structs[a[0]] = list(a[1:])
        elif op == 32:   # OpTypePointer
            pointers[a[0]] = a[2]
        elif op == 43 and a[0] in floats32:  # OpConstant float32
            consts.append(struct.unpack("<f", struct.pack("<I", a[2]))[0])
        elif op == 54:   # OpFunction
            functions.append(a[1])
        elif op == 59:   # OpVariable
            variables[a[1]] = a[0]
        i += n
    out = {"source": "assets/shaders/compute_pass.comp.spv", "local_size": local_size, "functions": sorted(names.get(f, "?") for f in functions),
           "float_constants": sorted(set(consts)), "blocks": {}, "images": {}}
    image_formats = {4: "Rgba8", 1: "Rgba32f", 0: "Unknown"}
    for var, ptr in variables.items():
        t = pointers.get(ptr)
        if t in images:
            out["images"][names.get(var, str(var))] = {"binding": bindings.get(var), "format": image_formats.get(images[t], images[t])}
        if t in blocks:
            members = []
            for k, mt in enumerate(structs[t]):
                m = {"name": member_names.get((t, k), "?"), "offset": offsets.get((t, k))}
                if mt in arrays:  # runtime array of records: element stride + the record's own member offsets
                    m["array_stride"] = strides.get(mt)
                    el = arrays[mt]
                    if el in structs:
                        m["element"] = names.get(el, "?")
                        m["element_members"] = [{"name": member_names.get((el, j), "?"), "offset": offsets.get((el, j))} for j in range(len(structs[el]))]
                members.append(m)
            out["blocks"][names.get(t, str(t))] = {"binding": bindings.get(var), "set": sets.get(var), "members": members}
    dst = ROOT / "tests" / "golden" / "reference_spv_facts.json"
    dst.write_text(json.dumps(out, indent=1) + "\n")
    print(dst, len(out["functions"]), "functions,", len(out["float_constants"]), "float constants,", len(out["blocks"]), "blocks")


if __name__ == "__main__":
    main()
