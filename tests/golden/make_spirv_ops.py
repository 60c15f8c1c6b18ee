"""Regenerates tests/golden/test_spirv_ops.json: the instruction list of the reference's compiled ray shader,
RayTracing/Assets/Shaders/Compiled/Ray/Test.spirv (1,524 bytes, dxc output of Test.hlsl:1-40), DECODED AS DATA.

The module is read as 32-bit little-endian words and split into (opcode, operand words) records exactly as the SPIR-V
binary format lays them out (word 0 of an instruction = word count << 16 | opcode). Nothing in it is executed, loaded by a
driver or translated: the output is a table of numbers — opcode names, result ids, operand ids, literal constants — which
tests/test_spirv_pin.py holds the oracle's claimed operation order against. It is the one artefact in the reference that
fixes the arithmetic of SURVEY.md §8a rows a1-a3 (reciprocal multiply for /1080, FMA-contracted discriminant and
numerator, one divide, two normalisations, strict `> 0`).

Run in the build container (the reference tree does not travel to the GPU box):  python tests/golden/make_spirv_ops.py
"""
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SPIRV = "/root/reference/RayTracing/Assets/Shaders/Compiled/Ray/Test.spirv"

# the opcodes that occur in this module (SPIR-V 1.0 numbering) and the GLSL.std.450 extended instructions it uses
OPCODES = {3: "OpSource", 5: "OpName", 11: "OpExtInstImport", 12: "OpExtInst", 14: "OpMemoryModel", 15: "OpEntryPoint", 16: "OpExecutionMode",
           17: "OpCapability", 19: "OpTypeVoid", 20: "OpTypeBool", 21: "OpTypeInt", 22: "OpTypeFloat", 23: "OpTypeVector", 25: "OpTypeImage",
           32: "OpTypePointer", 33: "OpTypeFunction", 43: "OpConstant", 44: "OpConstantComposite", 54: "OpFunction", 56: "OpFunctionEnd",
           59: "OpVariable", 61: "OpLoad", 71: "OpDecorate", 79: "OpVectorShuffle", 80: "OpCompositeConstruct", 81: "OpCompositeExtract",
           99: "OpImageWrite", 112: "OpConvertUToF", 127: "OpFNegate", 129: "OpFAdd", 131: "OpFSub", 133: "OpFMul", 136: "OpFDiv",
           142: "OpVectorTimesScalar", 148: "OpDot", 186: "OpFOrdGreaterThan", 245: "OpPhi", 247: "OpSelectionMerge", 248: "OpLabel",
           249: "OpBranch", 250: "OpBranchConditional", 253: "OpReturn"}
GLSL450 = {31: "Sqrt", 50: "Fma", 69: "Normalize"}
# instructions with (result type, result id) in words 1-2; the rest of this module's instructions have neither
HAS_RESULT = {12, 43, 44, 54, 59, 61, 79, 80, 81, 112, 127, 129, 131, 133, 136, 142, 148, 186, 245}
# of those, the operand words that are ids of other values (everything else is a literal)
LITERAL_TAIL = {81: 1, 79: None}  # OpCompositeExtract: composite id, then literal indices; OpVectorShuffle: two ids, then literal components


def decode(path=SPIRV):
    raw = open(path, "rb").read()
    assert len(raw) % 4 == 0
    w = struct.unpack("<%dI" % (len(raw) // 4), raw)
    assert w[0] == 0x07230203, "not a SPIR-V module"
    out = {"source": "RayTracing/Assets/Shaders/Compiled/Ray/Test.spirv (reference @ 2024-10-16), decoded by tests/golden/make_spirv_ops.py; data, not code",
           "bytes": len(raw), "sha256": hashlib.sha256(raw).hexdigest(), "version": "%d.%d" % ((w[1] >> 16) & 0xff, (w[1] >> 8) & 0xff),
           "generator": "0x%08x" % w[2], "bound": w[3], "instructions": []}
    i = 5
    while i < len(w):
        wc, op = w[i] >> 16, w[i] & 0xffff
        a = list(w[i + 1:i + wc])
        rec = {"op": OPCODES.get(op, "Op%d" % op)}
        if op in HAS_RESULT:
            rec["type"], rec["id"] = a[0], a[1]
            a = a[2:]
        elif op in (19, 20, 21, 22, 23, 25, 32, 33, 248, 11):  # types, labels, the ext-inst import: result id only
            rec["id"] = a[0]
            a = a[1:]
        if op == 12:  # OpExtInst: set id, instruction number, operands
            rec["set"], rec["inst"] = a[0], GLSL450.get(a[1], a[1])
            a = a[2:]
        if op == 43:  # OpConstant: one literal word (all constants here are 32-bit)
            rec["bits"] = "0x%08x" % a[0]
            rec["f32"] = struct.unpack("<f", struct.pack("<I", a[0]))[0]
            a = []
        if op in (5, 15, 11, 3):  # names / strings: keep the words as hex, they carry no arithmetic
            a = ["0x%08x" % x for x in a]
        rec["operands"] = a
        out["instructions"].append(rec)
        i += wc
    return out


if __name__ == "__main__":
    if not os.path.exists(SPIRV):
        sys.exit(f"{SPIRV} not found: this script runs in the build container only")
    with open(os.path.join(HERE, "test_spirv_ops.json"), "w") as f:
        json.dump(decode(), f, indent=0)
    print("wrote tests/golden/test_spirv_ops.json")
