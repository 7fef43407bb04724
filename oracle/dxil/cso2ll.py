"""TEST INFRASTRUCTURE (like everything under oracle/): turns the reference's SHIPPED compiled shaders (/root/reference/Bin/*.cso,
DXIL = LLVM bitcode in a DXBC container) into LLVM IR that the host compiler can build for x86-64, so that the reference's own
compiled code can be run on the CPU next to the oracle (oracle/_ref/libmv_dxref.so, built by `make -C oracle ref`).

Nothing of the reference is copied into the repository: the .cso files are read where they lie, the generated IR and the library go
to oracle/_ref/ (git-ignored).  What is NOT the reference's here, and is said so wherever results are compared: the resource
intrinsics (dx.op.sampleLevel, typed loads / stores, format conversion) are answered by oracle/dxil/dxil_rt.c with the oracle's
sampler / format contract -- that contract is hardware behaviour the bytecode does not contain.  Control flow, arithmetic, constants,
the order of operations and every early-out are the compiled shader's.

Modes of the translation (`mode`):
  "fp32": HLSL min16float is a minimum-precision hint (the shaders were compiled without -enable-16bit-types): the DXIL `half`
          values are carried in 32 bits, as WARP and most GPUs do.  The compiler has already folded min16float CONSTANTS to half
          precision (g_maxDist = 0xH42EE = 3.46484375, not 2*sqrt(3)); those are kept as compiled, widened exactly.
  "fp16": `half` stays IEEE half: what a GPU that honours the hint at 16 bits computes.
Fast-math flags are dropped (no reassociation / contraction by the host compiler); dx.op.tertiary FMad keeps ONE rounding or two
according to the runtime's switch (dxref_set_fused_mad).
"""
import re
import struct
import subprocess
import sys

LLVM_DIS = "/opt/rocm/lib/llvm/bin/llvm-dis"


def extract_bitcode(path):
    """DXBC container -> the DXIL part's LLVM bitcode, with the data layout's `i8:32` (which current LLVM rejects) rewritten to the
    equally long `i8:08` in the bit stream (the string is stored as 6-bit VBR chunks, two per character)."""
    d = open(path, "rb").read()
    if d[:4] != b"DXBC":
        raise ValueError(f"{path}: not a DXBC container")
    n = struct.unpack_from("<I", d, 28)[0]
    for o in struct.unpack_from(f"<{n}I", d, 32):
        if d[o:o + 4] != b"DXIL":
            continue
        sz = struct.unpack_from("<I", d, o + 4)[0]
        body = d[o + 8:o + 8 + sz]
        boff, bsz = struct.unpack_from("<II", body, 16)
        bc = bytearray(body[8 + boff:8 + boff + bsz])
        if bc[:4] != b"BC\xc0\xde":
            raise ValueError(f"{path}: no LLVM bitcode in the DXIL part")

        def vbr6(s):
            v = k = 0
            for c in s:
                o_ = ord(c)
                v |= ((o_ & 31) | 32) << k
                k += 6
                v |= ((o_ >> 5) & 31) << k
                k += 6
            return v, k

        pat, k = vbr6("i8:32")
        rep, _ = vbr6("i8:08")
        big = int.from_bytes(bc, "little")
        mask = (1 << k) - 1
        hits = [p for p in range(len(bc) * 8 - k) if (big >> p) & mask == pat]
        for p in hits:
            big = (big & ~(mask << p)) | (rep << p)
        return big.to_bytes(len(bc), "little")
    raise ValueError(f"{path}: no DXIL part")


def disassemble(bitcode):
    r = subprocess.run([LLVM_DIS, "-", "-o", "-"], input=bitcode, capture_output=True)
    if r.returncode != 0:
        raise RuntimeError("llvm-dis: " + r.stderr.decode())
    return r.stdout.decode()


def _half_const(m):
    h = int(m.group(1), 16)
    f = struct.unpack("<e", struct.pack("<H", h))[0]
    return "0x%016X" % struct.unpack("<Q", struct.pack("<d", float(f)))[0]


def _stub(ret, name, args):
    """IR definition of one dx.op function: forwards to the C runtime function rt_<name with dots as underscores>.
    Aggregates come back through an out pointer; %dx.types.Handle travels as its pointer; i1 / i8 / i16 arguments as i32."""
    cname = "rt_" + name.replace("dx.op.", "").replace(".", "_")
    params, call_args, pre = [], [], []
    for i, a in enumerate(args):
        a = a.strip()
        v = f"%a{i}"
        params.append(f"{a} {v}")
        if a == "%dx.types.Handle":
            pre.append(f"  %h{i} = extractvalue %dx.types.Handle {v}, 0")
            call_args.append(f"ptr %h{i}")
        elif a in ("i1", "i8", "i16"):
            pre.append(f"  %z{i} = zext {a} {v} to i32")
            call_args.append(f"i32 %z{i}")
        else:
            call_args.append(f"{a} {v}")
    body = [f"define internal {ret} @{name}({', '.join(params)}) alwaysinline {{"] + pre
    decl_args = ", ".join(x.split(" ")[0] for x in call_args)
    if ret == "void":
        body += [f"  call void @{cname}({', '.join(call_args)})", "  ret void"]
        decl = f"declare void @{cname}({decl_args})"
    elif ret == "%dx.types.Handle":
        body += [f"  %p = call ptr @{cname}({', '.join(call_args)})", "  %r = insertvalue %dx.types.Handle undef, ptr %p, 0", "  ret %dx.types.Handle %r"]
        decl = f"declare ptr @{cname}({decl_args})"
    elif ret.startswith("%dx.types."):
        body += [f"  %out = alloca {ret}", f"  call void @{cname}({', '.join(['ptr %out'] + call_args)})", f"  %r = load {ret}, ptr %out", f"  ret {ret} %r"]
        decl = f"declare void @{cname}({', '.join(['ptr'] + ([decl_args] if decl_args else []))})"
    else:
        body += [f"  %r = call {ret} @{cname}({', '.join(call_args)})", f"  ret {ret} %r"]
        decl = f"declare {ret} @{cname}({decl_args})"
    return "\n".join(body + ["}"]), decl


def translate(ll, entry, mode="fp32"):
    """llvm-dis output of a DXIL module -> IR for x86-64: entry point @main renamed to @<entry>, dx.op declarations replaced by
    definitions that call the C runtime, fast-math flags removed, `half` widened in mode "fp32"."""
    out = []
    for line in ll.splitlines():
        if line.startswith("target datalayout") or line.startswith("target triple") or line.startswith("source_filename") or line.startswith("; ModuleID"):
            continue
        if line.startswith("!") or line.startswith("attributes #"):
            continue  # metadata and attribute groups are not needed to run the code
        out.append(line)
    s = "\n".join(out)
    s = re.sub(r"\s*#\d+\b", "", s)  # attribute group references
    s = re.sub(r",\s*!\w+(?:\.\w+)* !\d+", "", s)  # metadata attachments (tbaa, noalias ...)
    s = re.sub(r"\b(fadd|fsub|fmul|fdiv|frem|fcmp|fneg) fast\b", r"\1", s)
    s = re.sub(r"\b(fadd|fsub|fmul|fdiv|frem|fcmp|fneg) (?:(?:nnan|ninf|nsz|arcp|contract|afn|reassoc) )+", r"\1 ", s)
    s = s.replace("define void @main()", f"define void @{entry}()")
    if mode == "fp32":
        s = re.sub(r"fptrunc float (\S+) to half", r"bitcast float \1 to float", s)
        s = re.sub(r"fpext half (\S+) to float", r"bitcast float \1 to float", s)
        s = re.sub(r"0xH([0-9A-Fa-f]{4})", _half_const, s)
        s = re.sub(r"\bhalf\b", "float", s)
        s = re.sub(r"@dx\.op\.(\w+)\.f16\b", r"@dx.op.\1.f32", s)
    elif mode != "fp16":
        raise ValueError(mode)
    # WaveReadLaneAt reads a REGISTER of another lane -- also of a lane that is inactive at the reading site (CSVolumeCull reads the
    # vertices of lanes 6 and 7 inside `if (wTid.x < 6)`; on hardware an inactive lane's register still holds the value).  The lanes
    # run as threads here, so every value that some ReadLaneAt passes is published by each lane where it is DEFINED, and the read
    # fetches the wanted lane's published value (waiting for it if that lane is behind).
    pub, lines = {}, s.splitlines()
    rd = re.compile(r"^(\s*)(%\S+) = call (float|i32) @dx\.op\.waveReadLaneAt\.(f32|i32)\(i32 117, (float|i32) (%[\w.]+), i32 ([^)]+)\)\s*$")
    fn = 0  # values are numbered per function (a library holds several)
    for i, line in enumerate(lines):
        if line.startswith("define "):
            fn += 1
        m = rd.match(line)
        if not m:
            continue
        ind, res, ty, suf, _, val, lane = m.groups()
        if (fn, val) not in pub:
            pub[(fn, val)] = (len(pub), ty, suf)
        lines[i] = f"{ind}{res} = call {ty} @rt_readlane_{suf}(i32 {pub[(fn, val)][0]}, {ty} {val}, i32 {lane})"
    if pub:
        out2, pending = [], []
        fn = 0
        for i, line in enumerate(lines):
            out2.append(line)
            if line.startswith("define "):
                fn += 1
            m = re.match(r"^\s*(%[\w.]+) = (\w+)", line)
            if m and (fn, m.group(1)) in pub:
                pending.append((fn, m.group(1)))
            # publish right behind the definition; behind the block's last phi when the definition is a phi
            nxt = lines[i + 1] if i + 1 < len(lines) else ""
            if pending and not re.match(r"^\s*%[\w.]+ = phi ", nxt):
                for v in pending:
                    k, ty, suf = pub[v]
                    out2.append(f"  call void @rt_publish_{suf}(i32 {k}, {ty} {v[1]})")
                pending = []
        lines = out2
        lines += ["", "declare void @rt_publish_f32(i32, float)", "declare void @rt_publish_i32(i32, i32)",
                  "declare float @rt_readlane_f32(i32, float, i32)", "declare i32 @rt_readlane_i32(i32, i32, i32)"]
    s = "\n".join(lines)
    # dx.op declarations -> stubs
    decls, seen, body_lines = [], set(), []
    for line in s.splitlines():
        m = re.match(r"declare (\S+) @(dx\.op\.[\w.]+)\((.*)\)\s*$", line)
        if m:
            ret, name, args = m.group(1), m.group(2), [a for a in m.group(3).split(",") if a.strip()]
            if name in seen:
                continue
            seen.add(name)
            stub, decl = _stub(ret, name, args)
            body_lines.append(stub)
            decls.append(decl)
            continue
        if line.startswith("; Function Attrs"):
            continue
        body_lines.append(line)
    uniq = []
    for d in decls:
        if d not in uniq:
            uniq.append(d)
    return ('target triple = "x86_64-unknown-linux-gnu"\n' + "\n".join(body_lines) + "\n\n" + "\n".join(uniq) + "\n")


def translate_lib(ll, prefix, mode="fp32"):
    """A lib_6_x module (the work-graph library Bin/LibRayMarch.cso): every defined function is a node; it becomes
    @<prefix>_<node>_<mode>.  What a library has that a single shader has not, and how it is answered:
      * resources are GLOBALS holding handles (`load %dx.types.Handle, ptr @"\01?g_txGrids@@..."[i]`): the mangled names become
        @lib_<name>, defined by the runtime (dxil_rt.c) as descriptor slots the test binds by name (dxref_lib_bind);
      * a cbuffer / structured buffer of a user type is loaded BY VALUE from a `_legacy` global and turned into a handle by
        dx.op.createHandleForLib.<type>: the call becomes a load of the slot @lib_h_<name>, the by-value load stays as dead code;
      * createHandleForLib / annotateHandle / annotateNodeHandle / annotateNodeRecordHandle pass their handle through;
      * node records live in addrspace(6): plain pointers here; getNodeRecordPtr is the address of record i behind the handle;
      * createNodeInputRecordHandle / createNodeOutputHandle / allocateNodeOutputRecords / outputComplete go to the runtime, which
        plays the work-graph scheduler (dxref_node_begin / dxref_node_records)."""
    s = ll
    s = re.sub(r'@"\\01\?(g_\w+)@@[^"]*_legacy"', r"@lib_\1_legacy", s)
    s = re.sub(r'@"\\01\?(g_\w+)@@[^"]*"', r"@lib_\1", s)
    s = s.replace("@cbSampleRes", "@lib_cbSampleRes")
    s = s.replace("ptr addrspace(6)", "ptr")
    s = "\n".join(l for l in s.splitlines() if not l.startswith("@llvm.global_ctors"))
    # user-typed resources: the by-value global becomes a dead zero, the handle comes from the runtime's slot
    slots = {}
    out = []
    for line in s.splitlines():
        m = re.match(r'^(@[\w.]+) = external global (%\S+|%"[^"]+")\s*$', line)
        if m:
            slots[m.group(2)] = "@lib_h_" + m.group(1)[1:].replace("_legacy", "").replace("lib_", "")
            out.append(f"{m.group(1)} = internal global {m.group(2)} zeroinitializer")
            continue
        out.append(line)
    s = "\n".join(out)
    for ty, slot in slots.items():
        s = re.sub(r'call %dx\.types\.Handle @"?dx\.op\.createHandleForLib\.[^"(]+"?\(i32 160, ' + re.escape(ty) + r' %\d+\)',
                   f"load %dx.types.Handle, ptr {slot}", s)
    s += "\n" + "\n".join(f"{slot} = external constant %dx.types.Handle" for slot in slots.values()) + "\n"
    # pass-through and record-address intrinsics: definitions instead of declarations
    defs = []

    def passthrough(m):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = ", ".join(f"{a.strip()} %a{i}" for i, a in enumerate(args.split(",")))
        defs.append(f"define internal {ret} @{name}({params}) alwaysinline {{\n  ret {ret} %a1\n}}")
        return ""
    s = re.sub(r"^declare (%dx\.types\.\w+) @(dx\.op\.(?:annotateHandle|annotateNodeHandle|annotateNodeRecordHandle|createHandleForLib\.dx\.types\.Handle))\((.*)\).*$",
               passthrough, s, flags=re.M)

    def record_ptr(m):
        name, ty = m.group(1), m.group(2)
        defs.append(f"define internal ptr @{name}(i32 %op, %dx.types.NodeRecordHandle %h, i32 %i) alwaysinline {{\n"
                    f"  %p = extractvalue %dx.types.NodeRecordHandle %h, 0\n  %q = getelementptr %{ty}, ptr %p, i32 %i\n  ret ptr %q\n}}")
        return ""
    s = re.sub(r"^declare ptr @(dx\.op\.getNodeRecordPtr\.((?:struct\.)?[\w.]+))\(.*\).*$", record_ptr, s, flags=re.M)
    s = re.sub(r'^declare %dx\.types\.Handle @"?dx\.op\.createHandleForLib\.hostlayout[^\n]*$', "", s, flags=re.M)
    nodes = re.findall(r"^define void @(\w+)\(\)", s, flags=re.M)
    for n in nodes:
        s = re.sub(r"@" + n + r"\b", f"@{prefix}_{n}_{mode}", s)
    return translate(s + "\n" + "\n".join(defs) + "\n", "main", mode), nodes


def resources(ll):
    """The dx.resources metadata as text lines (for a human: which rangeId is which register range)."""
    return [l for l in ll.splitlines() if l.startswith("!") and ("ptr undef" in l or "dx.resources" in l)]


if __name__ == "__main__":
    cso, entry, mode, out = sys.argv[1:5]
    text = disassemble(extract_bitcode(cso))
    if "@main()" in text:
        open(out, "w").write(translate(text, entry, mode))
    else:
        ir, nodes = translate_lib(text, entry, mode)
        open(out, "w").write(ir)
        print("library nodes:", " ".join(nodes))
    if len(sys.argv) > 5:
        open(sys.argv[5], "w").write(text)
